# round 6 A/B: d_model 128 SPLIT / QUAD kernels with the LayerNorm passes' small operands requested ahead (GT_SEQ_PFLN128; variant pfln128off = 0)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6q
mkdir -p $O
L=$PWD/transformergrooveinfilling_amd/lib
python tools/shape_bench.py --only 2 --steps 200 > /dev/null 2>&1
for rep in 1 2 3; do
  python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pfln128 on : bench %.4f ms  %.0f seq/s' % (d['ms_per_step'], d['value']))" >> $O/ab.txt
  GT_LIB_PATH=$L/libgroove_pfln128off.so python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pfln128 off: bench %.4f ms  %.0f seq/s' % (d['ms_per_step'], d['value']))" >> $O/ab.txt
  for i in 2 3; do
    python tools/shape_bench.py --only $i --steps 300 2>/dev/null | tail -1 | sed 's/^/pfln128 on : /' >> $O/ab.txt
    GT_LIB_PATH=$L/libgroove_pfln128off.so python tools/shape_bench.py --only $i --steps 300 2>/dev/null | tail -1 | sed 's/^/pfln128 off: /' >> $O/ab.txt
  done
done
for b in 16 32 128; do
  python tools/shape_bench.py --only 2 --batch $b --steps 300 2>/dev/null | tail -1 | sed 's/^/pfln128 on : /' >> $O/ab.txt
  GT_LIB_PATH=$L/libgroove_pfln128off.so python tools/shape_bench.py --only 2 --batch $b --steps 300 2>/dev/null | tail -1 | sed 's/^/pfln128 off: /' >> $O/ab.txt
done
cat $O/ab.txt
