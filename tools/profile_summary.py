"""Turn the rocprofv3 passes of tools/profile_rev.sh into the committed evidence under profiles/.

usage: python tools/profile_summary.py SCRATCH_DIR TAG WORKLOAD [STEPS_TRACED]
  SCRATCH_DIR/WORKLOAD/{trace,pmc_sq,pmc_fetch,pmc_write}  = one rocprofv3 pass each (separate runs, never combined)
writes profiles/TAG_WORKLOAD_kernel_stats.csv   (rocprofv3 --kernel-trace --stats summary, copied)
       profiles/TAG_WORKLOAD_mfma_hbm.md        per-kernel MFMA-busy %, HBM GB/s, HBM bytes per launch
       profiles/TAG_WORKLOAD_traffic.json       per kernel class HBM-side bytes per launch + csrc_sha (bench.py reads the c2 one
                                                and refuses it when the kernel sources have changed since)
       profiles/TAG_WORKLOAD_timeline.md        where a step's wall time goes: kernel durations vs. the gaps between one
                                                dispatch's end and the next one's begin (graph-replayed run)
Counter corrections follow MI355X_MICROARCH.md "HBM": FETCH_SIZE (KiB) under-reports wide coalesced reads by exactly 2x on
gfx950, WRITE_SIZE is exact -> HBM bytes = (2 FETCH_SIZE + WRITE_SIZE) * 1024.  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES /
(4 SIMDs x SQ_BUSY_CU_CYCLES)."""
import csv
import glob
import json
import os
import re
import shutil
import statistics
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

EPI = {0: None, 1: "gemm_wgrad", 2: "gemm_fwd_input", 3: "gemm_fwd_ffn1", 4: "gemm_fwd_heads", 5: "gemm_dgrad_ffn2",
       6: "gemm_dgrad_input", 7: "gemm_fwd_res_ln", 8: "gemm_dgrad_lnbwd"}
PLAIN = [("seq_tail_kernel", "seq_tail"), ("seq_update_pack_kernel", "optimizer"), ("wgrad_group_kernel", "gemm_wgrad"), ("wgrad_reduce", "gemm_wgrad_reduce"), ("attn_fwd", "attn_fwd"), ("attn_bwd", "attn_bwd"),
         ("attn_decode", "attn_decode"), ("ln_bwd", "ln_bwd"), ("ln_fwd", "ln_fwd"), ("heads_loss", "heads_loss"), ("loss_kernel", "loss"),
         ("sgd_kernel", "optimizer"), ("adam_kernel", "optimizer"), ("ln_param_reduce", "ln_param_reduce"),
         ("encoder_small", "encoder_small"), ("seq_fwd_kernel", "seq_fwd"), ("seq_fb_kernel", "seq_fwd"), ("seq_bwd_kernel", "seq_bwd"),
         ("seq_pack_kernel", "seq_pack"), ("wgrad32_group_kernel", "gemm_wgrad"), ("voice_metrics", "voice_metrics"),
         ("gather_batch", "gather_batch")]


def klass(name):
    """kernel name -> the class label the library's live profile (gt_profile_report) uses"""
    m = re.match(r"void (gemm_kernel|gemm32_kernel)<(.*?)>", name)
    if m:
        a = [x.strip() for x in m.group(2).split(",")]
        try:
            # gemm_kernel<WM, WN, TM, TN, BK, AKM, BKM, EPI, PREC>; gemm32_kernel<BKM, EPI, PREC>
            bkm, epi = (a[6] == "true", int(a[7])) if m.group(1) == "gemm_kernel" else (a[0] == "true", int(a[1]))
            return EPI.get(epi) or ("gemm_dgrad" if bkm else "gemm_fwd_bias")
        except (ValueError, IndexError):
            return "gemm"
    m = re.match(r"void (gemm32h_kernel|gemm64h_kernel)<(\d+)>", name)          # bf16-source kernels: NT for forward AND dgrad (the transposed weight copy)
    if m:
        return EPI.get(int(m.group(2))) or "gemm_fwd_bias"      # (EPI 0 = the plain store: forward Linears and dgrads share the kernel)
    m = re.match(r"void gemm64_kernel<(.*?)>", name)            # gemm64_kernel<BKM, EPI, PREC>
    if m:
        a = [x.strip() for x in m.group(1).split(",")]
        try:
            return EPI.get(int(a[1])) or ("gemm_dgrad" if a[0] == "true" else "gemm_fwd_bias")
        except (ValueError, IndexError):
            return "gemm"
    m = re.match(r"void gemm32row_kernel<(.*?)>", name)         # gemm32row_kernel<BN, BMW, BKM, EPI>: the LayerNorm-fused row tiles
    if m:
        a = [x.strip() for x in m.group(1).split(",")]
        try:
            return EPI.get(int(a[3])) or "gemm"
        except (ValueError, IndexError):
            return "gemm"
    if "wgrad32t_group_kernel" in name:
        return "gemm_wgrad"
    for k, v in PLAIN:
        if k in name:
            return v
    return None


def short(name):
    return re.sub(r"\(.*\)$", "", name.replace("void ", ""))[:70]


def find(d, pat):
    fs = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    return fs[0] if fs else None


def counters(d):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            a = acc[row["Kernel_Name"]][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
    return {k: {c: (v[0] / v[1], v[1]) for c, v in cs.items()} for k, cs in acc.items()}


def timeline(trace_csv, out_md, workload):
    rows = [r for r in csv.DictReader(open(trace_csv)) if r["Kind"] == "KERNEL_DISPATCH"]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ours = [r for r in rows if klass(r["Kernel_Name"]) or "gt_" in r["Kernel_Name"] or "_kernel" in r["Kernel_Name"]]
    # a step ends with the optimizer launch
    ends = [i for i, r in enumerate(rows) if klass(r["Kernel_Name"]) == "optimizer"]
    if len(ends) < 12:
        open(out_md, "w").write("# timeline (%s)\n\nnot enough optimizer launches in the trace (%d)\n" % (workload, len(ends)))
        return
    # steady state: the last 60 % of the steps (skip warm-up, graph capture and the eager profiling pass's neighbourhood)
    lo, hi = ends[len(ends) // 3], ends[-2]
    seg = rows[lo + 1:hi + 1]
    nsteps = sum(1 for r in seg if klass(r["Kernel_Name"]) == "optimizer")
    kern = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    wall = int(seg[-1]["End_Timestamp"]) - int(rows[lo]["End_Timestamp"])
    gaps, by_prev = [], defaultdict(list)
    prev = rows[lo]
    for r in seg:
        g = int(r["Start_Timestamp"]) - int(prev["End_Timestamp"])
        gaps.append(g)
        by_prev[klass(prev["Kernel_Name"]) or short(prev["Kernel_Name"])].append(g)
        prev = r
    # steps replayed back to back vs. step boundaries (host-side graph launch): gaps after the optimizer are the inter-step ones
    inter = by_prev.get("optimizer", [])
    intra = [g for k, v in by_prev.items() if k != "optimizer" for g in v]
    with open(out_md, "w") as f:
        f.write("# Kernel timeline, %s (steps as bench.py enqueues them (a replayed hipGraph, or direct launches for the dozen-launch sequence-resident step), rocprofv3 --kernel-trace begin/end timestamps)\n\n" % workload)
        f.write("%d steady-state steps, %.1f launches per step.\n\n" % (nsteps, len(seg) / nsteps))
        f.write("| per step | µs |\n|---|---|\n")
        f.write("| wall (first begin to last end, steps back to back) | %.1f |\n" % (wall / 1e3 / nsteps))
        f.write("| sum of kernel durations (begin to end of each dispatch) | %.1f |\n" % (kern / 1e3 / nsteps))
        f.write("| sum of gaps (end of one dispatch to begin of the next) | %.1f |\n" % (sum(gaps) / 1e3 / nsteps))
        f.write("| ... of which between steps (after the optimizer launch) | %.1f |\n" % (sum(inter) / 1e3 / nsteps))
        f.write("\nGap between consecutive dispatches inside a step: median %.2f µs, mean %.2f µs, p90 %.2f µs, max %.2f µs (negative = overlap).\n"
                % (statistics.median(intra) / 1e3, statistics.mean(intra) / 1e3, sorted(intra)[int(0.9 * len(intra))] / 1e3, max(intra) / 1e3))
        f.write("Gap between steps: median %.2f µs.\n\n" % (statistics.median(inter) / 1e3 if inter else float("nan")))
        f.write("| gap after a kernel of class | launches per step | mean gap µs | mean duration of that class µs |\n|---|---|---|---|\n")
        dur = defaultdict(list)
        for r in seg:
            dur[klass(r["Kernel_Name"]) or short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, v in sorted(by_prev.items(), key=lambda kv: -sum(kv[1])):
            f.write("| %s | %.1f | %.2f | %.2f |\n" % (k, len(v) / nsteps, statistics.mean(v) / 1e3, statistics.mean(dur.get(k, [0])) / 1e3))


def main():
    scratch, tag, wl = sys.argv[1], sys.argv[2], sys.argv[3]
    base = os.path.join(scratch, wl)
    prof = os.path.join(ROOT, "profiles")
    pre = os.path.join(prof, "%s_%s" % (tag, wl))
    import bench
    sha = bench.csrc_sha()
    try:
        rev = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL, text=True).strip()
    except Exception:
        rev = None
    if not rev:          # (the GPU box gets the tree without .git: tools/gpu_final_r05.sh stamps the revision into this file before it ships)
        try:
            rev = open(os.path.join(ROOT, "tools", ".git_rev")).read().strip() or None
        except OSError:
            rev = None
    stats = find(os.path.join(base, "trace"), "*kernel_stats.csv")
    trace = find(os.path.join(base, "trace"), "*kernel_trace.csv")
    if stats:
        shutil.copy(stats, pre + "_kernel_stats.csv")
    if trace:
        timeline(trace, pre + "_timeline.md", wl)
    dur = {}
    if stats:
        for row in csv.DictReader(open(stats)):
            dur[row["Name"]] = (float(row["AverageNs"]) / 1e3, int(row["Calls"]), float(row["Percentage"]))
    sq, fe, wr = counters(os.path.join(base, "pmc_sq")), counters(os.path.join(base, "pmc_fetch")), counters(os.path.join(base, "pmc_write"))
    ctf = {}
    cj = os.path.join(base, "class_tflops.json")          # tools/class_profile.py --json: flop rate per kernel class (live, eager pass)
    if os.path.exists(cj):
        ctf = json.load(open(cj))
        shutil.copy(cj, pre + "_class_tflops.json")
    with open(pre + "_mfma_hbm.md", "w") as f:
        f.write("# %s, workload %s: per-kernel MFMA-busy %% and HBM traffic (csrc %s, rev %s)\n\n" % (tag, wl, sha, rev))
        f.write("Durations: the kernel-trace pass (steps as the engine enqueues them).  Counters: three separate --pmc passes of the same workload, eager launches.\n\n")
        f.write("TFLOP/s: the algorithmic flops of the kernel's CLASS (2 M N K of its launches, tagged by the library) over the class's time in a live eager pass "
                "(tools/class_profile.py); kernels of one class share the figure.\n\n")
        f.write("| kernel | calls | avg µs | % of GPU time | class | TFLOP/s of the class | % of MFMA peak | MFMA busy % | HBM GB/s | HBM MB / launch |\n|---|---|---|---|---|---|---|---|---|---|\n")
        for name, (us, calls, pct) in sorted(dur.items(), key=lambda kv: -kv[1][2]):
            if pct < 0.4:
                continue
            s = sq.get(name, {})
            busy = 100.0 * s["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (4.0 * s["SQ_BUSY_CU_CYCLES"][0]) if s.get("SQ_BUSY_CU_CYCLES", (0,))[0] else float("nan")
            kib = 2.0 * fe.get(name, {}).get("FETCH_SIZE", (float("nan"),))[0] + wr.get(name, {}).get("WRITE_SIZE", (float("nan"),))[0]
            kc = klass(name) or ""
            c = ctf.get("classes", {}).get(kc, {})
            tfs = ("%.1f" % c["tflops"]) if c.get("tflops") else "-"
            frs = ("%.1f" % (100 * c["frac_of_mfma_peak"])) if c.get("tflops") else "-"
            f.write("| `%s` | %d | %.2f | %.1f | %s | %s | %s | %.1f | %.0f | %.2f |\n" % (short(name), calls, us, pct, kc, tfs, frs, busy, kib * 1024 / us / 1e3, kib * 1024 / 1e6))
    classes = {}
    for kern in set(fe) | set(wr):
        k = klass(kern)
        if not k or "FETCH_SIZE" not in fe.get(kern, {}) or "WRITE_SIZE" not in wr.get(kern, {}):
            continue
        n = fe[kern]["FETCH_SIZE"][1]
        o = classes.setdefault(k, {"launches": 0, "fetch_kib_raw": 0.0, "write_kib": 0.0})
        o["fetch_kib_raw"] += fe[kern]["FETCH_SIZE"][0] * n
        o["write_kib"] += wr[kern]["WRITE_SIZE"][0] * n
        o["launches"] += n
    for o in classes.values():
        n = o["launches"]
        o["fetch_kib_raw"] /= n
        o["write_kib"] /= n
        o["traffic_bytes_per_launch"] = (2.0 * o["fetch_kib_raw"] + o["write_kib"]) * 1024.0
    json.dump({"note": "HBM-side bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction), eager launches, separate --pmc passes",
               "workload": wl, "tag": tag, "csrc_sha": sha, "git_rev": rev, "classes": classes}, open(pre + "_traffic.json", "w"), indent=1)
    print("wrote", pre + "_{kernel_stats.csv,mfma_hbm.md,traffic.json,timeline.md}")


if __name__ == "__main__":
    main()
