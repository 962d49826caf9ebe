#!/bin/bash
# round-3 check: rider path parity + A/B timing against the grouped dispatch
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py -q -x -k "sequence_resident or deterministic or train_step" --timeout 600 > gpurun_out/r3a_pytest.log 2>&1; echo "pytest_exit=$?" >> gpurun_out/r3a_pytest.log
tail -5 gpurun_out/r3a_pytest.log
for round in 1 2; do
  for ride in 1 0; do
    for b in 64 32 16 96 128; do
      echo "ride=$ride $(GT_SEQ_RIDE=$ride python tools/shape_bench.py --only 2 --batch $b --steps 200 2>/dev/null | tail -1)"
    done
  done
done 2>&1 | tee gpurun_out/r3a_ab.log
timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/r3a_bench.log 2>&1; tail -1 gpurun_out/r3a_bench.log
