cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r2f}
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest_exit=$?" >> gpurun_out/pytest_$TAG.log
tail -8 gpurun_out/pytest_$TAG.log
timeout 600 python tools/shape_bench.py > gpurun_out/shapes_$TAG.log 2>&1; cat gpurun_out/shapes_$TAG.log
