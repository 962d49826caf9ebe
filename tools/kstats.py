"""Summarise a rocprofv3 *_kernel_stats.csv: per-kernel calls, average, share.  usage: kstats.py FILE [steps]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    n = r["Name"].replace("void gemm_kernel", "gemm").replace("(GemmArgs)", "")[:60]
    print("%-60s calls/step %6.2f avg %8.2f us  %5.1f%%" % (n, float(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3,
                                                          100 * float(r["TotalDurationNs"]) / tot))
print("total kernel time per step: %.1f us" % (tot / 1e3 / steps))
