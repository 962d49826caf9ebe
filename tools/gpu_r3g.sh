#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 2>&1 | tail -8 | tee gpurun_out/r3g_pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3g_bench20.log 2>&1; tail -1 gpurun_out/r3g_bench20.log | cut -c1-1500
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3g_bench20b.log 2>&1; tail -1 gpurun_out/r3g_bench20b.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['timing'])"
timeout 600 python bench.py --no-cpu-baseline --force-dp > gpurun_out/r3g_bench_dp.log 2>&1; tail -1 gpurun_out/r3g_bench_dp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['distributed'])"
GT_DP_OVERLAP=1 timeout 600 python bench.py --no-cpu-baseline --force-dp > gpurun_out/r3g_bench_dp_ov.log 2>&1; tail -1 gpurun_out/r3g_bench_dp_ov.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['distributed'])"
