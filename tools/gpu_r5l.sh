cd $GRAFT_REPO_ROOT
O=gpurun_out/r5l
mkdir -p $O; rm -f $O/*.txt
python tools/shape_bench.py --only 2 --steps 200 > /dev/null 2>&1
for i in 4 6 9 12 5; do python tools/shape_bench.py --only $i --steps 100 2>/dev/null | tail -1 >> $O/shapes.txt; done
python tools/graph_sync_stress.py 12 > $O/stress.txt 2>&1; tail -1 $O/stress.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
cat $O/shapes.txt
