#!/bin/bash
# Reproducible roofline evidence for ONE kernel revision (run on the GPU box through gpurun):
#   tools/profile_rev.sh TAG [workloads...]        workloads: c2 (bench.py, the headline command)  c3  c4 (shape_bench rows)
# Per workload FOUR separate rocprofv3 passes (counter passes are never combined with tracing, and each other):
#   trace     --kernel-trace --stats           graph-replayed steps: per-kernel durations + begin/end timeline
#   pmc_sq    --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES      eager launches
#   pmc_fetch --pmc FETCH_SIZE                                       eager launches
#   pmc_write --pmc WRITE_SIZE                                       eager launches
# then tools/profile_summary.py writes profiles/TAG_<workload>_{kernel_stats.csv,mfma_hbm.md,traffic.json,timeline.md}
# (copied back through gpurun_out/profiles_TAG/ as well, since only gpurun_out/ returns from the box).
TAG=${1:?tag}; shift
WL=${@:-c2 c3 c4}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
S=$R/gpurun_out/prof_$TAG
mkdir -p $S
cd /tmp && export TMPDIR=/tmp
for w in $WL; do
  case $w in
    c2) T="$R/bench.py --steps 100 --warmup 10 --no-cpu-baseline"; E="$R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-graph"; I=2 ;;
    c3) T="$R/tools/shape_bench.py --only 5 --steps 30 --warmup 5"; E="$R/tools/shape_bench.py --only 5 --steps 10 --warmup 3 --no-graph"; I=5 ;;
    c4) T="$R/tools/shape_bench.py --only 7 --steps 30 --warmup 5"; E="$R/tools/shape_bench.py --only 7 --steps 10 --warmup 3 --no-graph"; I=7 ;;
    c5) T="$R/tools/shape_bench.py --only 11 --steps 30 --warmup 5"; E="$R/tools/shape_bench.py --only 11 --steps 10 --warmup 3 --no-graph"; I=11 ;;
    c5s) T="$R/tools/shape_bench.py --only 9 --steps 50 --warmup 5"; E="$R/tools/shape_bench.py --only 9 --steps 10 --warmup 3 --no-graph"; I=9 ;;
    c4s) T="$R/tools/shape_bench.py --only 6 --steps 50 --warmup 5"; E="$R/tools/shape_bench.py --only 6 --steps 10 --warmup 3 --no-graph"; I=6 ;;
    *) echo "unknown workload $w"; exit 2 ;;
  esac
  mkdir -p $S/$w
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $S/$w/trace -- python3 $T > $S/$w/trace.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $S/$w/pmc_sq -- python3 $E > $S/$w/pmc_sq.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $S/$w/pmc_fetch -- python3 $E > $S/$w/pmc_fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $S/$w/pmc_write -- python3 $E > $S/$w/pmc_write.log 2>&1
  (cd $R && timeout 300 python3 tools/class_profile.py $I 10 --json $S/$w/class_tflops.json > $S/$w/class_profile.txt 2>&1; cp $S/$w/class_profile.txt $R/profiles/${TAG}_${w}_class_profile.txt)
  (cd $R && python3 tools/profile_summary.py $S $TAG $w)
  # the raw traces are large: keep only what the summary does not hold
  find $S/$w -name "*kernel_trace.csv" -size +20M -delete
  find $S/$w -name "*counter_collection.csv" -delete
done
mkdir -p $R/gpurun_out/profiles_$TAG && cp $R/profiles/${TAG}_* $R/gpurun_out/profiles_$TAG/ 2>/dev/null
ls -la $R/gpurun_out/profiles_$TAG
