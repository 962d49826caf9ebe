# round 6 A/B: the QUAD pair exchange on sequence-number tags (shipped) vs constant tag + consumer re-zeroing (libgroove_oldx.so: the seq TUs of the revision before)
cd $GRAFT_REPO_ROOT
O=gpurun_out/xtag; mkdir -p $O
python tools/shape_bench.py --only 2 --steps 300 > /dev/null 2>&1
for rep in 1 2 3; do
  python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seq tags  ', d['ms_per_step'], d['value'])" >> $O/ab.txt
  GT_LIB_PATH=$PWD/transformergrooveinfilling_amd/lib/libgroove_oldx.so python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('re-zeroing', d['ms_per_step'], d['value'])" >> $O/ab.txt
done
for b in 16 32; do
  python tools/shape_bench.py --only 2 --batch $b --steps 300 2>/dev/null | tail -1 | sed 's/^/seq tags  : /' >> $O/ab.txt
  GT_LIB_PATH=$PWD/transformergrooveinfilling_amd/lib/libgroove_oldx.so python tools/shape_bench.py --only 2 --batch $b --steps 300 2>/dev/null | tail -1 | sed 's/^/re-zeroing: /' >> $O/ab.txt
done
cat $O/ab.txt
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1
tail -4 $O/pytest_gpu.txt
