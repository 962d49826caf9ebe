"""Per-kernel MFMA-busy % and HBM GB/s for one workload: combines a rocprofv3 kernel-stats CSV (durations) with --pmc passes
(SQ_VALU_MFMA_BUSY_CYCLES + SQ_BUSY_CU_CYCLES in one pass; FETCH_SIZE and WRITE_SIZE in their own passes).
  MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)
  HBM GB/s    = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950 correction, MI355X_MICROARCH.md "HBM") / average duration
usage: python tools/pmc_mfma_hbm.py KERNEL_STATS.csv DIR_SQ DIR_FETCH DIR_WRITE > table.md"""
import csv
import glob
import os
import sys
from collections import defaultdict


def counters(d):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            a = acc[row["Kernel_Name"]][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
    return {k: {c: v[0] / v[1] for c, v in cs.items()} for k, cs in acc.items()}


dur = {}
for row in csv.DictReader(open(sys.argv[1])):
    dur[row["Name"]] = (float(row["AverageNs"]) / 1e3, int(row["Calls"]), float(row["Percentage"]))
sq, fe, wr = counters(sys.argv[2]), counters(sys.argv[3]), counters(sys.argv[4])
print("| kernel | calls | avg µs | % of GPU time | MFMA busy % | HBM GB/s | HBM MB / launch |")
print("|---|---|---|---|---|---|---|")
for name, (us, calls, pct) in sorted(dur.items(), key=lambda kv: -kv[1][2]):
    if pct < 0.5:
        continue
    s = sq.get(name, {})
    busy = 100.0 * s["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * s["SQ_BUSY_CU_CYCLES"]) if s.get("SQ_BUSY_CU_CYCLES") else float("nan")
    kib = 2.0 * fe.get(name, {}).get("FETCH_SIZE", float("nan")) + wr.get(name, {}).get("WRITE_SIZE", float("nan"))
    print("| `%s` | %d | %.2f | %.1f | %.1f | %.0f | %.2f |" % (name[:72], calls, us, pct, busy, kib * 1024 / us / 1e3, kib * 1024 / 1e6))
