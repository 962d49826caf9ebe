#!/bin/bash
# final revision: random-shape parity fuzz (general + the round's mid-size kernels forced at random) and the graph + host-sync stress over every shape
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5p
timeout 1200 python tools/fuzz_parity.py --n 30 --seed 11 > gpurun_out/r5p/fuzz.txt 2>&1; tail -3 gpurun_out/r5p/fuzz.txt
timeout 1500 python tools/fuzz_parity.py --mid --n 50 --seed 12 > gpurun_out/r5p/fuzz_mid.txt 2>&1; tail -3 gpurun_out/r5p/fuzz_mid.txt
timeout 1200 python tools/graph_sync_stress.py > gpurun_out/r5p/graph_sync_stress.txt 2>&1; tail -4 gpurun_out/r5p/graph_sync_stress.txt
