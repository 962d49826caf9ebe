# round-4 A/B helper (run on the GPU box through gpurun): tools/gpu_r4a.sh TAG "ENV=.. ENV=.." "ENV=.." ...   each variant twice, interleaved
TAG=$1; shift
mkdir -p gpurun_out/$TAG
python bench.py --no-cpu-baseline > /dev/null 2>&1     # (a fresh box reads low for its first seconds)
for rep in 1 2; do
  i=0
  for v in "$@"; do
    env $v python bench.py --no-cpu-baseline > gpurun_out/$TAG/bench_v${i}_$rep.json 2>> gpurun_out/$TAG/bench_err.log
    python - "$v" gpurun_out/$TAG/bench_v${i}_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-40s %.4f ms  %s" % (sys.argv[1], d["ms_per_step"], d["kernel_classes_us_per_step"]))
PY
    i=$((i+1))
  done
done
