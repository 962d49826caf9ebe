cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r2n}
timeout 1200 python -m pytest tests -q -m gpu --timeout 900 -k "sequence_resident or small_models" > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest_exit=$?" >> gpurun_out/pytest_$TAG.log
tail -3 gpurun_out/pytest_$TAG.log
python tools/class_profile.py 2 | head -10
for b in 16 32 64 96 128; do python tools/shape_bench.py --only 2 --batch $b --steps 300 | tail -1; GT_SEQ_SPLIT=0 python tools/shape_bench.py --only 2 --batch $b --steps 300 | tail -1 | sed 's/^/SPLIT=0 /';  GT_SEQ=0 python tools/shape_bench.py --only 2 --batch $b --steps 300 | tail -1 | sed 's/^/SEQ=0 /'; done
python bench.py --steps 200 --warmup 20 2>&1 | tail -1 | cut -c1-300
