cd $GRAFT_REPO_ROOT
O=gpurun_out/r5e
mkdir -p $O; rm -f $O/*.txt
python tools/shape_bench.py --only 2 --steps 200 > /dev/null 2>&1     # (warm the box)
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
tail -6 $O/pytest_gpu.txt
for i in 9 12 11 13; do python tools/shape_bench.py --only $i --steps 60 2>/dev/null | tail -1 >> $O/shapes.txt; done
for i in 12 13; do python tools/class_profile.py $i > $O/class_profile_$i.txt 2>&1; done
cat $O/shapes.txt
