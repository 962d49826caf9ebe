# round 6 A/B: the WHOLE-sequence kernels of d_model 32 (one workgroup per sequence: the testing YAML, C1) with the stage-ahead operand requests (variant whole = -DGT_SEQ_PF32_WHOLE=1)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6w3
mkdir -p $O
L=$PWD/transformergrooveinfilling_amd/lib
python tools/shape_bench.py --only 2 --steps 200 > /dev/null 2>&1
for rep in 1 2 3; do
  GT_LIB_PATH=$L/libgroove_whole.so python tools/shape_bench.py --only 0 --steps 300 2>/dev/null | tail -1 | sed 's/^/whole pf32 on : /' >> $O/ab.txt
  python tools/shape_bench.py --only 0 --steps 300 2>/dev/null | tail -1 | sed 's/^/whole pf32 off: /' >> $O/ab.txt
done
for b in 8 64 128; do
  GT_LIB_PATH=$L/libgroove_whole.so python tools/shape_bench.py --only 0 --batch $b --steps 300 2>/dev/null | tail -1 | sed 's/^/whole pf32 on : /' >> $O/ab.txt
  python tools/shape_bench.py --only 0 --batch $b --steps 300 2>/dev/null | tail -1 | sed 's/^/whole pf32 off: /' >> $O/ab.txt
done
GT_SEQ_SPLIT=1 python tools/shape_bench.py --only 0 --steps 300 2>/dev/null | tail -1 | sed 's/^/GT_SEQ_SPLIT=1 (shipped lib): /' >> $O/ab.txt
cat $O/ab.txt
