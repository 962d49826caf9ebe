#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
L=$R/transformergrooveinfilling_amd/lib
for round in 1 2; do
  for so in hip wc1024 wc2048 wc4096 nofuse nofuse_wc2048; do
    for i in 7 5 6; do echo "$so | $(GT_LIB_PATH=$L/libgroove_$so.so python tools/shape_bench.py --only $i --steps 30 2>/dev/null | tail -1)"; done
  done
done | tee gpurun_out/r3f_ab.log
