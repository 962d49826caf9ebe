# C5 (bf16 operands, bs512 on one GPU) at the three operand-shadow levels (0: fp32 sources; 1: bf16 copies beside; 2: operand-only tensors in bf16 alone)
for v in "GT_BF16_SHADOWS=0" "GT_BF16_SHADOWS=1" "GT_BF16_SHADOWS=2" "GT_BF16_SHADOWS=0" "GT_BF16_SHADOWS=2"; do
  echo "$v: $(env $v python tools/shape_bench.py --only 11 --steps 30 --warmup 5 2>/dev/null | tail -1)"
done
GT_BF16_SHADOWS=2 python tools/class_profile.py 11 10 2>/dev/null | grep -v amdgpu
