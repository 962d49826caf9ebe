# final measurements of a revision (run on the GPU box through gpurun): tools/gpu_final.sh TAG
cd $GRAFT_REPO_ROOT
TAG=${1:-r02_final}
O=gpurun_out/final_$TAG
mkdir -p $O
python tools/shape_bench.py --steps 200 > $O/shapes.txt 2>&1
for i in 0 1 2 3; do GT_SEQ=0 python tools/shape_bench.py --only $i --steps 200 | tail -1 | sed 's/^/GT_SEQ=0 (one kernel per op) /' >> $O/shapes.txt; done
GT_SEQ_SPLIT=0 python tools/shape_bench.py --only 2 --steps 200 | tail -1 | sed 's/^/GT_SEQ_SPLIT=0 (one workgroup per sequence) /' >> $O/shapes.txt
for b in 16 32 96 128 192; do python tools/shape_bench.py --only 2 --batch $b --steps 200 | tail -1 >> $O/shapes.txt; done
python bench.py --steps 300 --warmup 30 > $O/bench.json 2> $O/bench.err
# (diagnostic build first: hipcc ... groove_hip.hip -o gpurun_variants_stamps.so -DGT_SEQ_STAMPS)
for i in 0 1 2; do GT_LIB_PATH=$PWD/gpurun_variants_stamps.so python tools/seq_stamps.py $i > $O/seq_stamps_$i.txt 2>&1; done
GT_SEQ_SPLIT=1 GT_LIB_PATH=$PWD/gpurun_variants_stamps.so python tools/seq_stamps.py 2 > $O/seq_stamps_2_split.txt 2>&1
for i in 0 1 2 7 9 11; do python tools/class_profile.py $i > $O/class_profile_$i.txt 2>&1; done
GT_SEQ_SPLIT=0 python tools/class_profile.py 2 > $O/class_profile_2_whole.txt 2>&1
./tools/ubench/gemm_bench > $O/gemm_bench.txt 2>&1
./gpurun_variants_frag_load_bench 2>&1 | head -8 > $O/frag_load_bench.txt
./gpurun_variants_lat_bench 2>&1 | tail -13 > $O/lat_bench.txt
python tools/predict_bench.py > $O/predict.txt 2>&1
bash tools/profile_rev.sh $TAG c2 c4 c5 > $O/profile_rev.log 2>&1
tail -3 $O/profile_rev.log
cat $O/shapes.txt | tail -25
