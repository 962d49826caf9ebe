"""bench.py -- HVO sequences/sec per train step (BASELINE.json metric) on N MI355X of one node.

    python bench.py [--gpus N --steps K --warmup W]          (N>1: launched by torch.distributed.run)

A step = one pass of the hot path over one batch of synthetic 32x16 MSO inputs / 32x27 HVO targets
already resident in HBM: forward, BCE+MSE loss, backward, (N>1: RCCL all-reduce of the flat gradient
buffer), SGD update -- gt_train_step of libgroove_hip.so, replayed as one hipGraph.
Workload (N=1 and per GPU for N>1, weak scaling): BASELINE configs[1] -- InfillingClosedHH_training.yaml
hyper-parameters with the BASELINE shape overrides d_model=128 / 4 heads / 3 layers, bs=64, fp32.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORK = dict(d_model=128, n_heads=4, dim_feedforward=512, num_encoder_layers=3, num_decoder_layers=0,
            dropout=0.24, embedding_size_src=16)          # YAML: dropout .24, lr .07, penalty .38, sgd
BATCH, LR, PENALTY = 64, 0.07, 0.38
FP32_MATRIX_PEAK_TFLOPS = 157.3                            # MI355X_MICROARCH.md "Peak FP32 (matrix)"


def f_train_per_seq(w, T=32):
    """F_train = 3 * F_fwd (fwd + dgrad + wgrad GEMM flops), per sequence."""
    d, F, S, L, Ld = w["d_model"], w["dim_feedforward"], w["embedding_size_src"], w["num_encoder_layers"], w["num_decoder_layers"]
    f = 2 * T * S * d + L * (8 * T * d * d + 4 * T * T * d + 4 * T * d * F) + 2 * T * d * 27
    if Ld:
        f += 2 * T * 27 * d + Ld * (16 * T * d * d + 8 * T * T * d + 4 * T * d * F)
    return 3.0 * f


def cpu_baseline(budget_s=15.0):
    """The oracle's stock-torch restatement (oracle/torch_groove.py) timed on this box's host cores: the
    same config, batch and synthetic data, torch CPU fp32, all cores.  A reported baseline only."""
    import torch
    from oracle import numpy_groove as ng
    from oracle import torch_groove as tg
    m = tg.build(WORK, seed=0)
    opt = torch.optim.SGD(m.parameters(), lr=LR)
    x, y = ng.synthetic_batch(BATCH, WORK["embedding_size_src"], seed=1234)
    x, y = torch.from_numpy(x), torch.from_numpy(y)
    m.train()
    # pick the thread count that is fastest for this (small) model: all cores is NOT it on a many-core host
    ncpu = os.cpu_count() or 1
    best = (float("inf"), 1)
    for th in sorted({t for t in (4, 8, 16, 32, 64, ncpu) if t <= ncpu}):
        torch.set_num_threads(th)
        tg.train_step(m, opt, x, y, PENALTY)
        t0 = time.perf_counter()
        for _ in range(3):
            tg.train_step(m, opt, x, y, PENALTY)
        dt = (time.perf_counter() - t0) / 3
        if dt < best[0]:
            best = (dt, th)
        if dt > 4 * best[0]:
            break
    cores = best[1]
    torch.set_num_threads(cores)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        tg.train_step(m, opt, x, y, PENALTY)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": BATCH * n / dt, "unit": "sequences/s", "cores": cores, "kind": "port",
            "sample": "%d train steps of the same workload (bs %d) in %.1f s, torch %s CPU fp32, %d threads (fastest of a 4..%d sweep)"
                      % (n, BATCH, dt, torch.__version__, cores, ncpu)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--force-dp", action="store_true",
                    help="1 GPU only: run the data-parallel step sequence (graph, RCCL all-reduce in a 1-rank group, update) to "
                         "measure its non-communication overhead")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from oracle import numpy_groove as ng
    from transformergrooveinfilling_amd.engine import StepEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N>1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local
    if world > 1 or args.force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29533"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device(dev))

    eng = StepEngine(batch_size=BATCH, optimizer="sgd", learning_rate=LR, hit_loss_penalty=PENALTY, seed=1234 + rank,
                     device=dev, world_size=world, use_graph=not args.no_graph, **WORK)
    eng.force_dp = bool(args.force_dp)
    eng.load_named(ng.init_params(WORK, seed=0))                       # identical replicas
    x, y = ng.synthetic_batch(BATCH, WORK["embedding_size_src"], seed=1234 + rank)
    eng.x.copy_(torch.from_numpy(x)); eng.y.copy_(torch.from_numpy(y))   # inputs resident in HBM

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.train_step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.train_step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(eng.stats[0].item())

    out = None
    if rank == 0:
        seq_s = world * BATCH * args.steps / dt
        ftrain = f_train_per_seq(WORK)
        # dominant kernel, measured live: eager pass with HIP events around every launch on the launch stream
        prof = eng.profile(20)
        dom = max(prof.items(), key=lambda kv: kv[1][1])
        cnt, ms, fl, by = dom[1]
        achieved = (fl / cnt) / (ms / cnt * 1e-3) / 1e12 if ms > 0 else 0.0
        tot_ms = sum(v[1] for v in prof.values())
        # HBM-side bytes per launch of that kernel class: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same
        # workload (separate runs, see profiles/README.md), gfx950-corrected; null if the class was not measured
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
                traffic = json.load(f)["classes"][dom[0]]["traffic_bytes_per_launch"]
        except Exception:
            pass
        out = {
            "metric": "HVO sequences/sec (32-step, d_model=128) per train step", "value": seq_s, "unit": "sequences/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: InfillingClosedHH_training.yaml + overrides d_model=128/4 heads/3 layers, "
                                   "dim_feedforward=512, bs=64 per GPU, dropout=0.24, SGD lr=0.07, hit_loss_penalty=0.38, S=16, encoder-only",
                       "global_batch": world * BATCH, "parallelism": "dp%d" % world, "hipgraph": not args.no_graph},
            "roofline": {"bound": "mfma", "kernel": dom[0], "achieved": achieved, "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP32_MATRIX_PEAK_TFLOPS, "traffic": traffic,
                         "launches_per_step": cnt / 20.0, "avg_launch_us": 1e3 * ms / cnt,
                         "flops_per_launch": fl / cnt, "share_of_kernel_time": ms / tot_ms},
            "step_roofline": {"f_train_mflop_per_seq": ftrain / 1e6, "achieved_tflops": seq_s * ftrain / 1e12,
                              "frac_of_fp32_mfma_peak": seq_s * ftrain / 1e12 / (FP32_MATRIX_PEAK_TFLOPS * world)},
            "kernel_classes_us_per_step": {k: round(1e3 * v[1] / 20.0, 2) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])},
            "final_loss": loss,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    if world > 1 or args.force_dp:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its version banner through C stdio, which would otherwise be flushed at exit -- AFTER this line.
        # The JSON must be the last line on stdout.
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
