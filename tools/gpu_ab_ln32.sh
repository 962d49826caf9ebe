# round 6 A/B: LayerNorm inside the generic kernel's 32x32-tile Linears (GT_LN_XCHG32=0: the row pass of its own) -- the d_model-256 YAML shapes
cd $GRAFT_REPO_ROOT
O=gpurun_out/ln32; mkdir -p $O
python tools/shape_bench.py --only 2 --steps 200 > /dev/null 2>&1
for rep in 1 2; do
for i in 4 14 15; do
  python tools/shape_bench.py --only $i --steps 100 2>/dev/null | tail -1 | sed 's/^/ln32 on : /' >> $O/ab.txt
  GT_LN_XCHG32=0 python tools/shape_bench.py --only $i --steps 100 2>/dev/null | tail -1 | sed 's/^/ln32 off: /' >> $O/ab.txt
done
done
for b in 16 64; do
  python tools/shape_bench.py --only 4 --batch $b --steps 100 2>/dev/null | tail -1 | sed 's/^/ln32 on : /' >> $O/ab.txt
  GT_LN_XCHG32=0 python tools/shape_bench.py --only 4 --batch $b --steps 100 2>/dev/null | tail -1 | sed 's/^/ln32 off: /' >> $O/ab.txt
done
cat $O/ab.txt
python tools/class_profile.py 4 > $O/class_profile_4.txt 2>&1; head -20 $O/class_profile_4.txt
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1
tail -4 $O/pytest_gpu.txt
