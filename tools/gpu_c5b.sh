L=$PWD/transformergrooveinfilling_amd/lib
for v in "GT_BF16_SHADOWS=0" "GT_BF16_SHADOWS=1" "GT_BF16_SHADOWS=1 GT_LIB_PATH=$L/libgroove_dropfp32.so" "GT_BF16_SHADOWS=0" "GT_BF16_SHADOWS=1 GT_LIB_PATH=$L/libgroove_dropfp32.so"; do
  echo "$v: $(env $v python tools/shape_bench.py --only 11 --steps 30 --warmup 5 2>/dev/null | tail -1)"
done
GT_BF16_SHADOWS=1 GT_LIB_PATH=$L/libgroove_dropfp32.so python tools/class_profile.py 11 10 2>/dev/null | grep -v amdgpu | head -14
