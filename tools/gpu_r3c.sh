#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
L=$R/transformergrooveinfilling_amd/lib
timeout 600 python -m pytest tests/test_hip_parity.py -q -x -k "sequence_resident or deterministic or train_step" --timeout 600 2>&1 | tail -3 | tee gpurun_out/r3c_pytest.log
for so in hip d2 d3 d6; do GT_LIB_PATH=$L/libgroove_$so.so python tools/wg_unit_bench.py 64 2>&1 | grep "phase [12] ksplit [12]:" | sed "s/^/$so /"; done | tee gpurun_out/r3c_unit.log
for round in 1 2; do
  echo "ride=0 $(GT_SEQ_RIDE=0 python tools/shape_bench.py --only 2 --steps 300 2>/dev/null | tail -1)"
  for so in hip d2 d3 d6; do
    for pct in 50; do echo "$so pct=$pct $(GT_LIB_PATH=$L/libgroove_$so.so GT_SEQ_RIDE_LAST_PCT=$pct python tools/shape_bench.py --only 2 --steps 300 2>/dev/null | tail -1)"; done
  done
  for pct in 0 25 40 60 75 100; do echo "hip pct=$pct $(GT_SEQ_RIDE_LAST_PCT=$pct python tools/shape_bench.py --only 2 --steps 300 2>/dev/null | tail -1)"; done
  for ks in 1 4 8; do echo "hip ks=$ks $(GT_SEQ_TAIL_KS=$ks python tools/shape_bench.py --only 2 --steps 300 2>/dev/null | tail -1)"; done
done | tee gpurun_out/r3c_ab.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3c_trace -- python3 $R/tools/shape_bench.py --only 2 --steps 20 --warmup 5 > /dev/null 2>&1
f=$(find $R/gpurun_out/r3c_trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee $R/gpurun_out/r3c_trace.log
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"][:40], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Start_Timestamp"])) for r in rows]
idx = [i for i, k in enumerate(ks) if k[0].startswith("sgd_kernel")]
a, b = idx[-2] + 1, idx[-1] + 1
t0 = ks[a][2]
for k in ks[a:b]:
    print("%-42s start %8.1f us  dur %7.1f us" % (k[0], (k[2] - t0) / 1e3, k[1] / 1e3))
PY
rm -rf $R/gpurun_out/r3c_trace
