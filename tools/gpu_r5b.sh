# round 5: LayerNorm fused into the 64x64-tile Linears / dgrads through the in-launch row exchange -- A/B and parity
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b
mkdir -p $O; rm -f $O/shapes.txt
python tools/shape_bench.py --only 2 --steps 200 > /dev/null 2>&1     # (warm the box)
python tools/dbg_poll.py > $O/dbg_poll.txt 2>&1
for i in 6 9; do
  python tools/shape_bench.py --only $i --steps 100 2>/dev/null | tail -1 | sed 's/^/ln-xchg  /' >> $O/shapes.txt
  GT_LN_XCHG=0 python tools/shape_bench.py --only $i --steps 100 2>/dev/null | tail -1 | sed 's/^/separate /' >> $O/shapes.txt
done
for i in 6 9; do python tools/class_profile.py $i > $O/class_profile_$i.txt 2>&1; done
timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_hip_api.py::test_quad_pair_exchange_fails_safe_under_contention_and_on_timeout > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
cat $O/shapes.txt; cat $O/dbg_poll.txt | tail -20
