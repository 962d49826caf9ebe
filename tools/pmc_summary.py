"""Per-kernel-class HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, eager launches),
corrected as MI355X_MICROARCH.md "HBM" prescribes for gfx950: FETCH_SIZE under-reports wide coalesced reads by exactly 2x,
WRITE_SIZE is exact; both in KiB.  The classes are the ones bench.py's live profile uses, so bench.py can look up
`roofline.traffic` for its dominant kernel class.
usage: python tools/pmc_summary.py OUT.json DIR_FETCH DIR_WRITE [RAW_OUT.json]"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

EPI = {0: None, 1: "gemm_wgrad", 2: "gemm_fwd_input", 3: "gemm_fwd_ffn1", 4: "gemm_fwd_heads", 5: "gemm_dgrad_ffn2",
       6: "gemm_dgrad_input", 7: "gemm_fwd_res_ln", 8: "gemm_dgrad_lnbwd"}
PLAIN = {"wgrad_group_kernel": "gemm_wgrad", "attn_fwd_mfma_kernel": "attn_fwd", "attn_bwd_mfma_kernel": "attn_bwd",
         "attn_fwd_kernel": "attn_fwd", "attn_bwd_kernel": "attn_bwd", "attn_decode_kernel": "attn_decode",
         "ln_bwd_kernel": "ln_bwd", "ln_fwd_kernel": "ln_fwd", "loss_kernel": "loss", "sgd_kernel": "optimizer",
         "adam_kernel": "optimizer", "ln_param_reduce_kernel": "ln_param_reduce",
         }


def klass(name):
    m = re.match(r"void gemm_kernel<(.*?)>", name)
    if m:
        a = [x.strip() for x in m.group(1).split(",")]
        epi, bkm = int(a[-1]), a[-2] == "true"
        return EPI[epi] or ("gemm_dgrad" if bkm else "gemm_fwd_bias")
    for k, v in PLAIN.items():
        if k in name:
            return v
    return None


def read(d):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            a = acc[row["Kernel_Name"]][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
    return acc


raw = defaultdict(dict)
for d in sys.argv[2:4]:
    for kern, ctrs in read(d).items():
        for c, (tot, n) in ctrs.items():
            raw[kern][c] = tot / n
            raw[kern]["launches"] = n
out = {}
for kern, v in raw.items():
    k = klass(kern)
    if not k or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    o = out.setdefault(k, {"launches": 0, "fetch_kib_raw": 0.0, "write_kib": 0.0})
    n = v["launches"]
    o["fetch_kib_raw"] += v["FETCH_SIZE"] * n
    o["write_kib"] += v["WRITE_SIZE"] * n
    o["launches"] += n
for k, o in out.items():
    n = o.pop("launches")
    o["fetch_kib_raw"] /= n
    o["write_kib"] /= n
    o["traffic_bytes_per_launch"] = (2.0 * o["fetch_kib_raw"] + o["write_kib"]) * 1024.0
json.dump({"note": "HBM-side bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction), C2 workload bs=64, eager launches",
           "classes": out}, open(sys.argv[1], "w"), indent=1)
if len(sys.argv) > 4:
    json.dump(raw, open(sys.argv[4], "w"), indent=1)
for k, o in sorted(out.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"]):
    print("%-20s %8.2f MB per launch" % (k, o["traffic_bytes_per_launch"] / 1e6))
