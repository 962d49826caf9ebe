for v in "GT_BF16_WT=0" "GT_BF16_WT=1" "GT_BF16_WT=0" "GT_BF16_WT=1"; do
  for sh in 9 10; do echo "$v: $(env $v python tools/shape_bench.py --only $sh --steps 100 --warmup 10 2>/dev/null | tail -1)"; done
done
python tools/class_profile.py 9 10 2>/dev/null | grep -v amdgpu | head -8
