cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -x > gpurun_out/pytest_r2a.log 2>&1; echo "pytest_exit=$?" >> gpurun_out/pytest_r2a.log
tail -5 gpurun_out/pytest_r2a.log
timeout 300 python bench.py > gpurun_out/bench_r2a.log 2>&1; tail -1 gpurun_out/bench_r2a.log
timeout 300 python bench.py --gpus 1 --force-dp --no-cpu-baseline > gpurun_out/bench_r2a_dp.log 2>&1; tail -1 gpurun_out/bench_r2a_dp.log | cut -c1-300
timeout 300 python tools/shape_bench.py > gpurun_out/shapes_r2a.log 2>&1; cat gpurun_out/shapes_r2a.log
tools/profile_rev.sh r02a c2 c3 c4 > gpurun_out/profile_r2a.log 2>&1; tail -3 gpurun_out/profile_r2a.log
