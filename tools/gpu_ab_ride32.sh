# round 6 A/B: the d_model-32 SPLIT schedule with RIDER workgroups for the weight gradients (GT_SEQ_RIDE_D32=1) against the grouped launches at the end
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6r
mkdir -p $O
python tools/shape_bench.py --only 2 --steps 200 > /dev/null 2>&1
for rep in 1 2 3; do
  GT_SEQ_RIDE_D32=1 python tools/shape_bench.py --only 1 --steps 300 2>/dev/null | tail -1 | sed 's/^/riders : /' >> $O/ab.txt
  python tools/shape_bench.py --only 1 --steps 300 2>/dev/null | tail -1 | sed 's/^/grouped: /' >> $O/ab.txt
done
for b in 8 32 64; do
  GT_SEQ_RIDE_D32=1 python tools/shape_bench.py --only 1 --batch $b --steps 300 2>/dev/null | tail -1 | sed 's/^/riders : /' >> $O/ab.txt
  python tools/shape_bench.py --only 1 --batch $b --steps 300 2>/dev/null | tail -1 | sed 's/^/grouped: /' >> $O/ab.txt
done
cat $O/ab.txt
GT_SEQ_RIDE_D32=1 python tools/class_profile.py 1 > $O/class_profile_1_riders.txt 2>&1
cat $O/class_profile_1_riders.txt
GT_SEQ_RIDE_D32=1 timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_api.py -m gpu -q -x > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
