#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
L=$R/transformergrooveinfilling_amd/lib
for round in 1 2; do
  for so in hip stag8 stag16 stag24; do
    echo "$so $(GT_LIB_PATH=$L/libgroove_$so.so python tools/shape_bench.py --only 2 --steps 300 2>/dev/null | tail -1)"
  done
done | tee gpurun_out/r3h_ab.log
