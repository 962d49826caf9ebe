cd $GRAFT_REPO_ROOT
O=gpurun_out/r6p10
mkdir -p $O
L=$PWD/transformergrooveinfilling_amd/lib
python tools/shape_bench.py --only 2 --steps 200 > /dev/null 2>&1
for rep in 1 2 3; do
  python tools/shape_bench.py --only 1 --steps 300 2>/dev/null | tail -1 | sed 's/^/pf32 on      : /' >> $O/ab.txt
  GT_LIB_PATH=$L/libgroove_pfln0.so python tools/shape_bench.py --only 1 --steps 300 2>/dev/null | tail -1 | sed 's/^/no ln at start: /' >> $O/ab.txt
done
cat $O/ab.txt
