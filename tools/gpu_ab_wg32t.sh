# round 6 A/B: slab depth / LDS row stride of the bf16 weight-gradient kernel (wgrad32t_group_kernel): BK 32 STR 160 (shipped) vs BK 64 STR 144 vs BK 32 STR 144
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6w2
mkdir -p $O
L=transformergrooveinfilling_amd/lib
python tools/shape_bench.py --only 2 --steps 200 > /dev/null 2>&1
bash tools/ab_variants.sh "11 9" $L/libgroove_wg16.so $L/libgroove_wg32o4.so > $O/ab.txt 2>&1
cat $O/ab.txt
for v in hip wg16 wg32o4; do
  for i in 11 9; do GT_LIB_PATH=$PWD/$L/libgroove_$v.so python tools/class_profile.py $i 2>&1 | grep "kernel time\|gemm_wgrad" | sed "s/^/$v | /" >> $O/classes.txt; done
done
cat $O/classes.txt
GT_LIB_PATH=$PWD/$L/libgroove_wg16.so timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "bf16 or precision or keep_bits" > $O/pytest_wg16.txt 2>&1
tail -3 $O/pytest_wg16.txt
