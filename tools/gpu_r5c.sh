cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c
mkdir -p $O; rm -f $O/*.txt
python tools/shape_bench.py --only 2 --steps 200 > /dev/null 2>&1     # (warm the box)
for s in "2048 512 512" "2048 1536 512" "2048 512 1536" "8192 256 256"; do
  f=$(echo $s | tr ' ' '_')
  ./tools/ubench/gemm_bench $s 2>&1 | grep -E "gemm64|^NN|^C\[" > "$O/gb_deep_$f.txt"
  ./tools/ubench/gemm_bench_m0 $s 2>&1 | grep -E "gemm64|^NN|^C\[" > "$O/gb_two_$f.txt"
done
head -20 $O/gb_*.txt
