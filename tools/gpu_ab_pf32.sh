# round 6 A/B: d_model 32 SPLIT kernels with every stage's global operands requested a stage ahead (GT_SEQ_PF32; variant pf0 = off)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6p9
mkdir -p $O
L=$PWD/transformergrooveinfilling_amd/lib
python tools/shape_bench.py --only 2 --steps 200 > /dev/null 2>&1
for rep in 1 2 3; do
for i in 1 0 2; do
  python tools/shape_bench.py --only $i --steps 300 2>/dev/null | tail -1 | sed 's/^/pf32 on : /' >> $O/ab.txt
  GT_LIB_PATH=$L/libgroove_pf0.so python tools/shape_bench.py --only $i --steps 300 2>/dev/null | tail -1 | sed 's/^/pf32 off: /' >> $O/ab.txt
done
done
cat $O/ab.txt
GT_SEQ_SPLIT=1 GT_LIB_PATH=$L/libgroove_stamps.so python tools/seq_stamps.py 1 > $O/seq_stamps_closedhh.txt 2>&1
GT_SEQ_SPLIT=1 GT_LIB_PATH=$L/libgroove_stamps0.so python tools/seq_stamps.py 1 > $O/seq_stamps_closedhh_off.txt 2>&1
head -12 $O/seq_stamps_closedhh_off.txt | cut -c1-220
head -12 $O/seq_stamps_closedhh.txt | cut -c1-220
true

