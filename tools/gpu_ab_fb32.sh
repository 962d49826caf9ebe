# round 6 A/B: d_model 32 SPLIT schedule with the last forward phase and backward phase 0 in one launch (seq_fb32_kernel; GT_SEQ_FUSE_B0_D32=0: two launches)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6f
mkdir -p $O
python tools/shape_bench.py --only 2 --steps 200 > /dev/null 2>&1
for rep in 1 2 3; do
  python tools/shape_bench.py --only 1 --steps 300 2>/dev/null | tail -1 | sed 's/^/fused   : /' >> $O/ab.txt
  GT_SEQ_FUSE_B0_D32=0 python tools/shape_bench.py --only 1 --steps 300 2>/dev/null | tail -1 | sed 's/^/separate: /' >> $O/ab.txt
done
cat $O/ab.txt
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_api.py -m gpu -q -x > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
