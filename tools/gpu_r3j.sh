#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
for round in 1 2; do
  for mn in 100000000 1024 256; do
    for sh in 6 7 11; do
      echo "lds_min=$mn $(GT_ATTN_BWD_LDS_MIN=$mn python tools/shape_bench.py --only $sh --steps 40 2>/dev/null | tail -1)"
    done
  done
done | tee gpurun_out/r3j_ab.log
GT_ATTN_BWD_LDS_MIN=100000000 python tools/class_profile.py 7 20 2>/dev/null | grep -i "attn\|kernel time" | tee -a gpurun_out/r3j_ab.log
python tools/class_profile.py 7 20 2>/dev/null | grep -i "attn\|kernel time" | tee -a gpurun_out/r3j_ab.log
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r3j_tests.log
