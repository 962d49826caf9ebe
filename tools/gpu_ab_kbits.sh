# round 6 A/B: keep bits of the FFN activation (FFN1's epilogue writes one bit per element, the FFN2 dgrad reads them instead of hact); GT_FFN_KBITS=0: off
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6k2
mkdir -p $O
python tools/shape_bench.py --only 2 --steps 200 > /dev/null 2>&1
for rep in 1 2; do
for i in 7 11 13 6 9 5; do
  python tools/shape_bench.py --only $i --steps 40 --warmup 5 2>/dev/null | tail -1 | sed 's/^/kbits on : /' >> $O/ab.txt
  GT_FFN_KBITS=0 python tools/shape_bench.py --only $i --steps 40 --warmup 5 2>/dev/null | tail -1 | sed 's/^/kbits off: /' >> $O/ab.txt
done
done
cat $O/ab.txt
python tools/class_profile.py 7 > $O/class_profile_7.txt 2>&1
python tools/class_profile.py 11 > $O/class_profile_11.txt 2>&1
grep -h "ffn\|kernel time" $O/class_profile_*.txt
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
