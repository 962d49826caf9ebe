# data-parallel step sequence on ONE GPU (1-rank RCCL group): what the split costs besides communication
for v in "GT_DP_GRAPH=0" "GT_DP_GRAPH=1" "GT_DP_GRAPH=0 GT_DP_OVERLAP=1" "GT_DP_GRAPH=1 GT_DP_OVERLAP=1"; do
  r=$(env $v python bench.py --no-cpu-baseline --force-dp 2>gpurun_out/dp_err.log | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms, dp_overhead_us %.1f, fused %.4f' % (d['ms_per_step'], d['distributed']['dp_overhead_us'], d['distributed']['fused_single_process_ms_per_step']))" 2>&1 | tail -1)
  echo "$v: $r"; tail -2 gpurun_out/dp_err.log | grep -i "error\|Traceback" 
done
