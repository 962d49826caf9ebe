#!/bin/bash
# round 5: LayerNorm as the prologue of the Linear that consumes it (gt_lnpro.h; GT_LN_PRO=0: the row pass of its own)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5r
o=gpurun_out/r5r/ab.txt; : > $o
for round in 1 2; do
  for i in 4 14 15; do
    echo "prologue | $(python tools/shape_bench.py --only $i --steps 300 2>/dev/null | tail -1)" >> $o
    echo "row pass | $(GT_LN_PRO=0 python tools/shape_bench.py --only $i --steps 300 2>/dev/null | tail -1)" >> $o
  done
done
cat $o
python tools/class_profile.py 4 2>&1 | tee gpurun_out/r5r/class_profile_4.txt | head -22
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_api.py -q -m gpu --timeout 600 2>&1 | tail -4
