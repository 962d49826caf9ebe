# round 6 A/B: the reference CLI's default shape (d_model 64, 16 heads of 4, F 256, L 7, bs 16) on the SPLIT schedule with the vector-ALU attention (default now) against the whole-sequence kernels (GT_SEQ_SPLIT=0)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6s
mkdir -p $O
python tools/shape_bench.py --only 2 --steps 200 > /dev/null 2>&1
for rep in 1 2 3; do
  python tools/shape_bench.py --only 16 --steps 300 2>/dev/null | tail -1 | sed 's/^/split : /' >> $O/ab.txt
  GT_SEQ_SPLIT=0 python tools/shape_bench.py --only 16 --steps 300 2>/dev/null | tail -1 | sed 's/^/whole : /' >> $O/ab.txt
done
for b in 8 32 64 128; do
  python tools/shape_bench.py --only 16 --batch $b --steps 300 2>/dev/null | tail -1 | sed 's/^/split : /' >> $O/ab.txt
  GT_SEQ_SPLIT=0 python tools/shape_bench.py --only 16 --batch $b --steps 300 2>/dev/null | tail -1 | sed 's/^/whole : /' >> $O/ab.txt
done
cat $O/ab.txt
python tools/class_profile.py 16 2>&1 | grep -v amdgpu > $O/class_profile_16.txt; cat $O/class_profile_16.txt
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_api.py -m gpu -q -x > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
