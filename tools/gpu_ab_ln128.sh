# round 6 A/B: LayerNorm inside the 128x128-tile Linears (GT_LN_XCHG128=0: the row pass of its own)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6a
mkdir -p $O
python tools/shape_bench.py --only 2 --steps 200 > /dev/null 2>&1
for rep in 1 2; do
for i in 7 11 13; do
  python tools/shape_bench.py --only $i --steps 40 --warmup 5 2>/dev/null | tail -1 | sed 's/^/ln128 on : /' >> $O/ab.txt
  GT_LN_XCHG128=0 python tools/shape_bench.py --only $i --steps 40 --warmup 5 2>/dev/null | tail -1 | sed 's/^/ln128 off: /' >> $O/ab.txt
done
done
for b in 256; do
  python tools/shape_bench.py --only 7 --batch $b --steps 60 2>/dev/null | tail -1 | sed 's/^/ln128 on : /' >> $O/ab.txt
  GT_LN_XCHG128=0 python tools/shape_bench.py --only 7 --batch $b --steps 60 2>/dev/null | tail -1 | sed 's/^/ln128 off: /' >> $O/ab.txt
  python tools/shape_bench.py --only 11 --batch $b --steps 60 2>/dev/null | tail -1 | sed 's/^/ln128 on : /' >> $O/ab.txt
  GT_LN_XCHG128=0 python tools/shape_bench.py --only 11 --batch $b --steps 60 2>/dev/null | tail -1 | sed 's/^/ln128 off: /' >> $O/ab.txt
done
cat $O/ab.txt
python tools/class_profile.py 7 > $O/class_profile_7.txt 2>&1
python tools/class_profile.py 11 > $O/class_profile_11.txt 2>&1
head -30 $O/class_profile_7.txt $O/class_profile_11.txt
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
