#!/bin/bash
# round 5: OutputLayer on the skinny kernel (GT_HEADS_KERNEL=0: the generic GEMM)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5q
o=gpurun_out/r5q/ab.txt; : > $o
for round in 1 2; do
  for i in 4 6 9 12 11 5 14; do
    echo "skinny | $(python tools/shape_bench.py --only $i --steps 200 2>/dev/null | tail -1)" >> $o
    echo "generic | $(GT_HEADS_KERNEL=0 python tools/shape_bench.py --only $i --steps 200 2>/dev/null | tail -1)" >> $o
  done
done
cat $o
python tools/class_profile.py 9 2>&1 | grep -E "heads|kernel time"
python tools/class_profile.py 6 2>&1 | grep -E "heads|kernel time"
python tools/class_profile.py 4 2>&1 | grep -E "heads|kernel time"
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_api.py -q -m gpu --timeout 600 2>&1 | tail -4
