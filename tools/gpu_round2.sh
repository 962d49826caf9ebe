cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r2n}
timeout 1200 python -m pytest tests -q -m gpu --timeout 900 -k "sequence_resident or small_models" > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest_exit=$?" >> gpurun_out/pytest_$TAG.log
tail -3 gpurun_out/pytest_$TAG.log
GT_LIB_PATH=$PWD/gpurun_variants_stamps.so python tools/seq_stamps.py 0 > gpurun_out/stamps_c1_$TAG.txt; GT_LIB_PATH=$PWD/gpurun_variants_stamps.so python tools/seq_stamps.py 1 > gpurun_out/stamps_hh_$TAG.txt
grep -E "forward|layer 1|backward|final|epilogue" gpurun_out/stamps_c1_$TAG.txt gpurun_out/stamps_hh_$TAG.txt
python tools/class_profile.py 0 | head -4; python tools/class_profile.py 1 | head -4
for i in 0 1; do python tools/shape_bench.py --only $i --steps 300 | tail -1; GT_SEQ=0 python tools/shape_bench.py --only $i --steps 300 | tail -1 | sed 's/^/GT_SEQ=0 /'; done
