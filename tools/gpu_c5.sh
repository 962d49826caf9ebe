# C5 (bf16 operands, bs512 on one GPU): weight gradients on the transposed-LDS-read kernel vs widened into the fp32 image
L=$PWD/transformergrooveinfilling_amd/lib
for lib in libgroove_nowg32t.so libgroove_hip.so libgroove_nowg32t.so libgroove_hip.so; do
  echo "$lib: $(GT_LIB_PATH=$L/$lib python tools/shape_bench.py --only 11 --steps 30 --warmup 5 2>/dev/null | tail -1)"
done
python tools/class_profile.py 11 10 2>/dev/null | grep -v amdgpu | head -6
