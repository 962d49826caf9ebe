cd $GRAFT_REPO_ROOT
O=gpurun_out/r5f
mkdir -p $O; rm -f $O/*.txt
python tools/shape_bench.py --only 2 --steps 200 > /dev/null 2>&1     # (warm the box)
for rep in 1 2; do for i in 9 12 11 13; do python tools/shape_bench.py --only $i --steps 60 2>/dev/null | tail -1 >> $O/shapes.txt; done; done
for i in 12 13; do python tools/class_profile.py $i > $O/class_profile_$i.txt 2>&1; done
timeout 600 python -m pytest tests -m gpu -x -q -k "precision2 or bf16" > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
cat $O/shapes.txt; head -8 $O/class_profile_12.txt; grep -E "ln_|attn" $O/class_profile_13.txt
