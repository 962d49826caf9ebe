#!/bin/bash
# Run on the GPU box (via gpurun): parity tests, bench, rocprofv3 kernel stats.  usage: tools/gpu_check.sh TAG [pytest-args]
TAG=${1:-run}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests -q -m gpu --timeout 600 "$@" > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest_exit=$?" >> gpurun_out/pytest_$TAG.log
tail -4 gpurun_out/pytest_$TAG.log
timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/bench_$TAG.log 2>&1
tail -1 gpurun_out/bench_$TAG.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seq/s', d['value'], 'ms/step', d['ms_per_step'], 'dom', d['roofline']['kernel'], d['roofline']['achieved'], d['kernel_classes_us_per_step'])" || tail -5 gpurun_out/bench_$TAG.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/benchprof_$TAG.log 2>&1
cd $R
python tools/kstats.py $(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1) 131 | head -40
