#!/bin/bash
# round 5, late A/B: (1) LayerNorm row exchange forced at the d_model-256 YAMLs (1024 tokens: 64 tiles of 64x64), (2) weight gradients per layer on
# the side stream at d_model 512 / 2048 tokens (variant build: -DGT_WGRAD_DEFER_MAX_M=0), (3) the two remaining d_model-256 YAMLs
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5m
o=gpurun_out/r5m/ab.txt
: > $o
for round in 1 2; do
  for i in 4 14 15 5; do
    echo "default | $(python tools/shape_bench.py --only $i --steps 200 2>/dev/null | tail -1)" >> $o
    echo "GT_LN_XCHG=1 | $(GT_LN_XCHG=1 python tools/shape_bench.py --only $i --steps 200 2>/dev/null | tail -1)" >> $o
  done
  for i in 6 9 12; do
    echo "default | $(python tools/shape_bench.py --only $i --steps 100 2>/dev/null | tail -1)" >> $o
    echo "per-layer wgrad | $(GT_LIB_PATH=$PWD/transformergrooveinfilling_amd/lib/libgroove_perlayer.so python tools/shape_bench.py --only $i --steps 100 2>/dev/null | tail -1)" >> $o
    echo "per-layer wgrad, side stream | $(GT_OVERLAP=1 GT_LIB_PATH=$PWD/transformergrooveinfilling_amd/lib/libgroove_perlayer.so python tools/shape_bench.py --only $i --steps 100 2>/dev/null | tail -1)" >> $o
    echo "per-layer wgrad, side stream, eager | $(GT_OVERLAP=1 GT_LIB_PATH=$PWD/transformergrooveinfilling_amd/lib/libgroove_perlayer.so python tools/shape_bench.py --only $i --steps 100 --no-graph 2>/dev/null | tail -1)" >> $o
  done
done
cat $o
timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -k "test_step_parity and not large" --timeout 300 2>&1 | tail -5
