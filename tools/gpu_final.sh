# final measurements of a revision (run on the GPU box through gpurun): tools/gpu_final.sh TAG
# (build first: csrc/build.sh, tools/build_variant.sh stamps groove_hip,groove_seq_fwd,groove_seq_bwd,groove_seq64 -DGT_SEQ_STAMPS, tools/ubench/gemm_bench)
cd $GRAFT_REPO_ROOT
TAG=${1:-r06_final}
O=gpurun_out/final_$TAG
L=$PWD/transformergrooveinfilling_amd/lib
mkdir -p $O
python tools/shape_bench.py --only 2 --steps 200 > /dev/null 2>&1     # (a fresh box reads 5-14 % low for its first seconds)
python tools/shape_bench.py --steps 200 2>/dev/null > $O/shapes.txt
for i in 0 1 2 3; do GT_SEQ=0 python tools/shape_bench.py --only $i --steps 200 2>/dev/null | tail -1 | sed 's/^/GT_SEQ=0 (one kernel per op) /' >> $O/shapes.txt; done
GT_SEQ_SPLIT=0 python tools/shape_bench.py --only 2 --steps 200 2>/dev/null | tail -1 | sed 's/^/GT_SEQ_SPLIT=0 (one workgroup per sequence) /' >> $O/shapes.txt
GT_SEQ_QUAD=0 python tools/shape_bench.py --only 2 --steps 200 2>/dev/null | tail -1 | sed 's/^/GT_SEQ_QUAD=0 (two workgroups per sequence in every phase: round 3) /' >> $O/shapes.txt
GT_SEQ_QUAD_BWD0=0 python tools/shape_bench.py --only 2 --steps 200 2>/dev/null | tail -1 | sed 's/^/GT_SEQ_QUAD_BWD0=0 (four workgroups per sequence in the forward only) /' >> $O/shapes.txt
GT_SEQ_FUSE_B0=0 python tools/shape_bench.py --only 2 --steps 200 2>/dev/null | tail -1 | sed 's/^/GT_SEQ_FUSE_B0=0 (backward phase 0 as a launch of its own) /' >> $O/shapes.txt
for l in 0 1 2; do GT_BF16_SHADOWS=$l python tools/shape_bench.py --only 11 --steps 30 --warmup 5 2>/dev/null | tail -1 | sed "s/^/GT_BF16_SHADOWS=$l /" >> $O/shapes.txt; done
GT_SEQ_RIDE=0 python tools/shape_bench.py --only 2 --steps 200 2>/dev/null | tail -1 | sed 's/^/GT_SEQ_RIDE=0 (grouped weight gradients at the end) /' >> $O/shapes.txt
GT_PACK_FOLD=0 python tools/shape_bench.py --only 2 --steps 200 2>/dev/null | tail -1 | sed 's/^/GT_PACK_FOLD=0 (packing launch at the head of every step) /' >> $O/shapes.txt
for b in 16 32 80 96 128 192; do python tools/shape_bench.py --only 2 --batch $b --steps 200 2>/dev/null | tail -1 >> $O/shapes.txt; done
# round 5: the 2048-token regime (a GPU's share of configs[3] / [4]) with the switches one by one
for i in 6 9 12; do
  GT_T64R_MIN=100000000 python tools/shape_bench.py --only $i --steps 100 2>/dev/null | tail -1 | sed 's/^/GT_T64R_MIN=inf (no 64x64 ring tiles: round 4 tile rules) /' >> $O/shapes.txt
  GT_LN_XCHG=0 python tools/shape_bench.py --only $i --steps 100 2>/dev/null | tail -1 | sed 's/^/GT_LN_XCHG=0 (LayerNorm as a row pass of its own instead of the in-launch row exchange) /' >> $O/shapes.txt
done
# round 6: LayerNorm inside the 128x128-tile Linears (bs 512 on one GPU)
for i in 7 11 13; do
  GT_LN_XCHG128=0 python tools/shape_bench.py --only $i --steps 40 --warmup 5 2>/dev/null | tail -1 | sed 's/^/GT_LN_XCHG128=0 (LayerNorm row pass instead of the big-tile epilogue) /' >> $O/shapes.txt
done
for i in 7 11 13 5; do
  GT_FFN_KBITS=0 python tools/shape_bench.py --only $i --steps 40 --warmup 5 2>/dev/null | tail -1 | sed 's/^/GT_FFN_KBITS=0 (the FFN2 dgrad reads the activation instead of its keep bits) /' >> $O/shapes.txt
done
python tools/kbits_check.py > $O/kbits_check.txt 2>&1
GT_ROW_FUSE_XCHG=0 python tools/shape_bench.py --only 5 --steps 100 2>/dev/null | tail -1 | sed 's/^/GT_ROW_FUSE_XCHG=0 (C3: row-owning LayerNorm tiles, the path until round 5) /' >> $O/shapes.txt
GT_ROW_FUSE_BIG_MAX_D=0 GT_LN_XCHG=0 python tools/shape_bench.py --only 5 --steps 100 2>/dev/null | tail -1 | sed 's/^/GT_ROW_FUSE_BIG_MAX_D=0 GT_LN_XCHG=0 (C3: 64x64 ring tiles + LayerNorm row pass) /' >> $O/shapes.txt
python bench.py --steps 300 --warmup 30 2> $O/bench.err | tail -1 > $O/bench.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline | tail -1 > $O/bench_driver_style_1.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline | tail -1 > $O/bench_driver_style_2.json 2>/dev/null
python bench.py --no-cpu-baseline --force-dp | tail -1 > $O/bench_force_dp.json 2>/dev/null
GT_DP_GRAPH=1 python bench.py --no-cpu-baseline --force-dp | tail -1 > $O/bench_force_dp_graph.json 2>/dev/null
GT_DP_OVERLAP=1 python bench.py --no-cpu-baseline --force-dp | tail -1 > $O/bench_force_dp_overlap.json 2>/dev/null
GT_DP_GRAPH=1 GT_DP_OVERLAP=1 python bench.py --no-cpu-baseline --force-dp | tail -1 > $O/bench_force_dp_overlap_graph.json 2>/dev/null
for i in 0 1; do GT_SEQ_SPLIT=1 GT_LIB_PATH=$L/libgroove_stamps.so python tools/seq_stamps.py $i > $O/seq_stamps_$i.txt 2>&1; done     # (the shipped path of these shapes: SPLIT)
GT_SEQ_SPLIT=1 GT_LIB_PATH=$L/libgroove_stamps.so python tools/seq_stamps.py 2 > $O/seq_stamps_c2.txt 2>&1
python tools/wg_unit_bench.py 64 > $O/wg_unit_bench.txt 2>&1
for i in 0 1 4 5 6 7 9 11 12 13 14 15 16; do python tools/class_profile.py $i > $O/class_profile_$i.txt 2>&1; done
./tools/ubench/gemm_bench > $O/gemm_bench.txt 2>&1
for s in "2048 512 512" "2048 1536 512" "2048 512 1536" "8192 256 256" "8192 768 256"; do echo "== $s" >> $O/gemm_bench_mid.txt; ./tools/ubench/gemm_bench $s 2>&1 | grep -E "gemm64|gemm32.h|^NN|32x32   <|64x64   <2,2,2,2,BK32" >> $O/gemm_bench_mid.txt; done
python tools/predict_bench.py > $O/predict.txt 2>&1
bash tools/profile_rev.sh $TAG c2 c3 c4 c5 c4s c5s > $O/profile_rev.log 2>&1
tail -3 $O/profile_rev.log
cat $O/shapes.txt | tail -30
# the bench lines again AFTER the PMC pass of this revision, so that roofline.traffic is this revision's own (bench.py refuses a traffic file of another csrc_sha)
mkdir -p profiles && cp gpurun_out/profiles_$TAG/* profiles/ 2>/dev/null
python bench.py --steps 300 --warmup 30 2> $O/bench.err | tail -1 > $O/bench.json
python bench.py --steps 20 --warmup 5 | tail -1 > $O/bench_driver_style_full.json 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
