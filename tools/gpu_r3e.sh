#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu --timeout 900 -x 2>&1 | tail -5 | tee gpurun_out/r3e_pytest.log
for round in 1 2; do
  echo "fold=1 $(python tools/shape_bench.py --only 2 --steps 300 2>/dev/null | tail -1)"
  echo "fold=0 $(GT_PACK_FOLD=0 python tools/shape_bench.py --only 2 --steps 300 2>/dev/null | tail -1)"
  for b in 16 32 96 128; do echo "b=$b $(python tools/shape_bench.py --only 2 --batch $b --steps 300 2>/dev/null | tail -1)"; done
  for i in 0 1; do echo "$(python tools/shape_bench.py --only $i --steps 300 2>/dev/null | tail -1)"; done
done | tee gpurun_out/r3e_ab.log
timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/r3e_bench.log 2>&1; tail -1 gpurun_out/r3e_bench.log
