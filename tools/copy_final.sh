#!/bin/bash
# after `gpurun -- bash tools/gpu_final.sh TAG`: copy what the collection wrote under gpurun_out/ into profiles/ (the judged, committed copies)
TAG=${1:-r06_final}
cd "$(dirname "$0")/.."
P=gpurun_out/profiles_$TAG; O=gpurun_out/final_$TAG
cp $P/* profiles/
for f in bench.json bench_driver_style_1.json bench_driver_style_2.json bench_driver_style_full.json bench_force_dp.json bench_force_dp_graph.json \
         bench_force_dp_overlap.json bench_force_dp_overlap_graph.json shapes.txt gemm_bench.txt gemm_bench_mid.txt predict.txt wg_unit_bench.txt pytest_gpu.txt kbits_check.txt; do
  [ -f $O/$f ] && cp $O/$f profiles/${TAG}_$f
done
for f in $O/class_profile_*.txt; do n=$(basename $f .txt); cp $f profiles/${TAG}_class_profile_shape${n#class_profile_}.txt; done
cp $O/seq_stamps_0.txt profiles/${TAG}_seq_stamps_c1.txt; cp $O/seq_stamps_1.txt profiles/${TAG}_seq_stamps_closedhh.txt; cp $O/seq_stamps_c2.txt profiles/${TAG}_seq_stamps_c2.txt
grep -h "csrc" profiles/${TAG}_c2_mfma_hbm.md | head -1
python - <<P
import json
for n in ("bench", "bench_driver_style_1", "bench_driver_style_2", "bench_driver_style_full"):
    d = json.loads(open("profiles/${TAG}_%s.json" % n).read().strip().splitlines()[-1])
    print(n, d["ms_per_step"], d["value"], d["roofline"].get("traffic"), d.get("csrc_sha"), d.get("git_rev"))
P
