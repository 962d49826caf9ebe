cd $GRAFT_REPO_ROOT
O=gpurun_out/r5k
mkdir -p $O; rm -f $O/*.txt
python tools/shape_bench.py --only 2 --steps 200 > /dev/null 2>&1
for rep in 1 2; do for i in 6 9 12; do
  python tools/shape_bench.py --only $i --steps 100 2>/dev/null | tail -1 | sed 's/^/separate /' >> $O/shapes.txt
  GT_LN_XCHG=1 python tools/shape_bench.py --only $i --steps 100 2>/dev/null | tail -1 | sed 's/^/ln-xchg  /' >> $O/shapes.txt
done; done
GT_LN_XCHG=1 python tools/class_profile.py 6 > $O/class_profile_6.txt 2>&1
timeout 600 python -m pytest tests -m gpu -x -q -k "layernorm_row_exchange" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
cat $O/shapes.txt; head -6 $O/class_profile_6.txt
