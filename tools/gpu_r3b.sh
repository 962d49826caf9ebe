#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
python tools/wg_unit_bench.py 64 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3b_unit.log
for ride in 1 0; do echo "ride=$ride $(GT_SEQ_RIDE=$ride python tools/shape_bench.py --only 2 --steps 200 2>/dev/null | tail -1)"; done | tee gpurun_out/r3b_ab.log
cd /tmp && export TMPDIR=/tmp
for ride in 1 0; do
  GT_SEQ_RIDE=$ride rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3b_trace$ride -- python3 $R/tools/shape_bench.py --only 2 --steps 20 --warmup 5 > /dev/null 2>&1
  f=$(find $R/gpurun_out/r3b_trace$ride -name "*kernel_trace.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"][:40], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Start_Timestamp"]), r.get("Grid_Size"), r.get("Workgroup_Size")) for r in rows]
# last full step: find last sgd
idx = [i for i, k in enumerate(ks) if k[0].startswith("sgd_kernel")]
a, b = idx[-2] + 1, idx[-1] + 1
t0 = ks[a][2]
for k in ks[a:b]:
    print("%-42s start %8.1f us  dur %7.1f us grid %s wg %s" % (k[0], (k[2] - t0) / 1e3, k[1] / 1e3, k[3], k[4]))
PY
done 2>&1 | tee $R/gpurun_out/r3b_trace.log
rm -rf $R/gpurun_out/r3b_trace1 $R/gpurun_out/r3b_trace0
