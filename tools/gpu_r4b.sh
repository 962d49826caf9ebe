# round-4 A/B helper for library variants (run on the GPU box through gpurun): tools/gpu_r4b.sh TAG LIB[,ENV=..] ...   each twice, interleaved;
# headline bench + the ClosedHH / K&S shapes' step times
TAG=$1; shift
mkdir -p gpurun_out/$TAG
python bench.py --no-cpu-baseline > /dev/null 2>&1
for rep in 1 2; do
  for v in "$@"; do
    lib=${v%%,*}; envs=""; [[ "$v" == *,* ]] && envs=${v#*,}
    env GT_LIB_PATH=$PWD/transformergrooveinfilling_amd/lib/libgroove_$lib.so ${envs//,/ } python bench.py --no-cpu-baseline > gpurun_out/$TAG/bench_${lib}_$rep.json 2>> gpurun_out/$TAG/err.log
    python - "$v" gpurun_out/$TAG/bench_${lib}_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-40s %.4f ms  %s" % (sys.argv[1], d["ms_per_step"], d["kernel_classes_us_per_step"]))
PY
  done
done
