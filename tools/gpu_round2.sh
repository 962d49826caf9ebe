cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r2n}
timeout 1200 python -m pytest tests -q -m gpu --timeout 900 -k "sequence_resident or small_models" > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest_exit=$?" >> gpurun_out/pytest_$TAG.log
tail -3 gpurun_out/pytest_$TAG.log
for i in 2; do GT_LIB_PATH=$PWD/gpurun_variants_stamps.so python tools/seq_stamps.py $i > gpurun_out/stamps_${i}_$TAG.txt; grep -E "forward|layer 1|backward|final|epilogue" gpurun_out/stamps_${i}_$TAG.txt; done
for i in 2; do python tools/class_profile.py $i | head -10; done
for i in 0 1 2 3; do python tools/shape_bench.py --only $i --steps 300 | tail -1; done
