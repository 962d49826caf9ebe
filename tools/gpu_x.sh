#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
L=$R/transformergrooveinfilling_amd/lib
for round in 1 2 3; do
  for so in nopre5 hip; do
    echo "$so $(GT_LIB_PATH=$L/libgroove_$so.so python tools/shape_bench.py --only 2 --steps 300 2>/dev/null | tail -1)"
  done
done | tee gpurun_out/x_ab.log
for so in nopre5 hip; do for b in 32 128; do echo "$so $(GT_LIB_PATH=$L/libgroove_$so.so python tools/shape_bench.py --only 2 --batch $b --steps 200 2>/dev/null | tail -1)"; done; done | tee -a gpurun_out/x_ab.log
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "seq or rider or train or determin" 2>&1 | tail -2 | tee -a gpurun_out/x_ab.log
