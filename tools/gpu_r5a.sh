# round 5, first measurement of the 64x64 ring tiles (gt_gemm64.h): tools/gpu_r5a.sh   (through gpurun)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5a
mkdir -p $O
python tools/shape_bench.py --only 2 --steps 200 > /dev/null 2>&1     # (warm the box)
for s in "2048 512 512" "2048 1536 512" "2048 512 1536" "8192 256 256" "8192 768 256" "16384 512 512"; do ./tools/ubench/gemm_bench $s > "$O/gemm_bench_$(echo $s | tr ' ' '_').txt" 2>&1; done
for i in 6 9 5 4 7 11; do
  python tools/shape_bench.py --only $i --steps 100 2>/dev/null | tail -1 | sed 's/^/new  /' >> $O/shapes.txt
  GT_T64R_MIN=100000000 python tools/shape_bench.py --only $i --steps 100 2>/dev/null | tail -1 | sed 's/^/old  /' >> $O/shapes.txt
done
for i in 6 9 5; do python tools/class_profile.py $i > $O/class_profile_$i.txt 2>&1; done
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
cat $O/shapes.txt
