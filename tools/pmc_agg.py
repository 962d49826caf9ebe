"""Average rocprofv3 --pmc counter values per kernel over all dispatches.
usage: python tools/pmc_agg.py DIR [DIR ...]   (each DIR = output of one `rocprofv3 --pmc ... --output-format csv -d DIR` pass)
prints one line per kernel: launches, then COUNTER=mean for every counter found."""
import csv
import glob
import os
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            a = acc[row["Kernel_Name"]][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
for k in sorted(acc, key=lambda k: -max(v[0] for v in acc[k].values())):
    c = acc[k]
    n = max(v[1] for v in c.values())
    print("%-62s n=%-5d " % (k[:62], n) + "  ".join("%s=%.4g" % (name, v[0] / v[1]) for name, v in sorted(c.items())))
