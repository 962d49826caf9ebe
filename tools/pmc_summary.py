"""Turn the per-kernel FETCH_SIZE / WRITE_SIZE averages of two rocprofv3 --pmc passes (gpurun_out/pmc_summary.json, written by
the collection one-liner in profiles/README.md) into per-kernel-class HBM traffic, corrected as MI355X_MICROARCH.md
"HBM" prescribes for gfx950: FETCH_SIZE under-reports wide coalesced reads by exactly 2x, WRITE_SIZE is exact; both in KiB.
usage: python tools/pmc_summary.py gpurun_out/pmc_summary.json profiles/r01_traffic.json"""
import json
import re
import sys

EPI = {0: None, 1: "gemm_wgrad", 2: "gemm_fwd_input", 3: "gemm_fwd_ffn1", 4: "gemm_fwd_heads", 5: "gemm_dgrad_ffn2",
       6: "gemm_dgrad_input", 7: "gemm_fwd_res_ln", 8: "gemm_dgrad_lnbwd"}


def klass(name):
    m = re.match(r"void gemm_kernel<(.*?)>", name)
    if m:
        a = [x.strip() for x in m.group(1).split(",")]
        epi, bkm = int(a[-1]), a[-2] == "true"
        return EPI[epi] or ("gemm_dgrad" if bkm else "gemm_fwd_bias")
    for k in ("wgrad_group_kernel", "attn_fwd_kernel", "attn_bwd_kernel", "ln_bwd_kernel", "ln_fwd_kernel", "loss_kernel", "sgd_kernel",
              "adam_kernel", "ln_param_reduce_kernel", "chain_fwd_kernel", "chain_bwd_kernel"):
        if k in name:
            return {"wgrad_group_kernel": "gemm_wgrad", "sgd_kernel": "optimizer", "adam_kernel": "optimizer"}.get(k, k.replace("_kernel", ""))
    return None


src = json.load(open(sys.argv[1]))
out = {}
for name, v in src.items():
    k = klass(name)
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
           "classes": out}, open(sys.argv[2], "w"), indent=1)
for k, o in sorted(out.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"]):
    print("%-20s %8.2f MB per launch" % (k, o["traffic_bytes_per_launch"] / 1e6))
