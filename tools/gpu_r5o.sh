#!/bin/bash
# round 5: four slabs in flight in registers on the small-tile generic GEMM (GT_RING=4) against the one-deep loop (variant build -DGT_RING=1)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5o
o=gpurun_out/r5o/ab.txt; : > $o
V=$PWD/transformergrooveinfilling_amd/lib/libgroove_ring1.so
for round in 1 2; do
  for i in 4 14 15 6 9 5 2; do
    echo "ring 4 | $(python tools/shape_bench.py --only $i --steps 200 2>/dev/null | tail -1)" >> $o
    echo "one-deep | $(GT_LIB_PATH=$V python tools/shape_bench.py --only $i --steps 200 2>/dev/null | tail -1)" >> $o
  done
  for i in 0 1; do
    echo "GT_SEQ=0 ring 4 | $(GT_SEQ=0 python tools/shape_bench.py --only $i --steps 200 2>/dev/null | tail -1)" >> $o
    echo "GT_SEQ=0 one-deep | $(GT_SEQ=0 GT_LIB_PATH=$V python tools/shape_bench.py --only $i --steps 200 2>/dev/null | tail -1)" >> $o
  done
done
cat $o
python tools/class_profile.py 4 > gpurun_out/r5o/class_profile_4.txt 2>&1; cat gpurun_out/r5o/class_profile_4.txt
GT_LIB_PATH=$V python tools/class_profile.py 4 > gpurun_out/r5o/class_profile_4_onedeep.txt 2>&1; cat gpurun_out/r5o/class_profile_4_onedeep.txt
timeout 1200 python -m pytest tests/test_hip_parity.py -q -m gpu --timeout 600 2>&1 | tail -5
