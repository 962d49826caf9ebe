cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r2d}
tools/ubench/gemm_bench > gpurun_out/gemm_bench_$TAG.log 2>&1; cat gpurun_out/gemm_bench_$TAG.log
tools/ubench/gemm_bench 2048 1536 512 | head -8
tools/ubench/gemm_bench 16384 512 512 | head -8
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 -x > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest_exit=$?" >> gpurun_out/pytest_$TAG.log
tail -8 gpurun_out/pytest_$TAG.log
timeout 600 python tools/shape_bench.py > gpurun_out/shapes_$TAG.log 2>&1; cat gpurun_out/shapes_$TAG.log
