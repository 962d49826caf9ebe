#!/bin/bash
# round 5: x0 bf16 shadow (layer 0's in-proj weight gradient joins the transposed-read group)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5n
o=gpurun_out/r5n/ab.txt; : > $o
for round in 1 2; do for i in 9 12 11 13 5; do python tools/shape_bench.py --only $i --steps 100 2>/dev/null | tail -1 >> $o; done; done
cat $o
python tools/class_profile.py 9 > gpurun_out/r5n/class_profile_9.txt 2>&1; cat gpurun_out/r5n/class_profile_9.txt
python tools/class_profile.py 5 > gpurun_out/r5n/class_profile_5.txt 2>&1; cat gpurun_out/r5n/class_profile_5.txt
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "bf16 or precision2 or shadow or autocast" --timeout 600 2>&1 | tail -5
