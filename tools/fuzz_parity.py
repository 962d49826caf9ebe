"""Randomised parity sweep on the GPU: odd model shapes through parity.check_step / check_predict / check_bucketed_backward
(full step against the fp64 oracle).  usage: python tools/fuzz_parity.py [--n 24] [--seed 0]"""
import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401  (first: one HIP runtime)

import parity  # noqa: E402
from harness import cfg_dict  # noqa: E402


def run(n, seed, backend="hip", verbose=True):
    """-> number of failing shapes among n random draws (some draws are skipped as invalid)."""
    rnd = random.Random(seed)
    fails = 0
    for k in range(n):
        H = rnd.choice([1, 2, 3, 4, 6, 8, 16])
        hd = rnd.choice([1, 2, 4, 6, 8, 16, 24, 32, 64])
        d = H * hd
        if d % 2 or d > 512 or d < 4:      # d_model 2: LayerNorm over two features is +-1 with cancelling gradients -- the
            continue                       # oracle's own fp32 run is 10 % off its fp64 run there, nothing to compare against
        F = rnd.choice([7, 16, 24, 48, 100, 128, 512, 640])
        L = rnd.choice([1, 2, 3])
        Ld = rnd.choice([0, 0, 1, 2])
        B = rnd.choice([1, 2, 3, 5, 9, 17, 33, 64])
        S = rnd.choice([16, 27, 5])
        p = rnd.choice([0.0, 0.1, 0.3])
        cfg = cfg_dict(d, H, F, L, Ld, embedding_size_src=S)
        tag = "d%d H%d F%d L%d+%d B%d S%d p%.1f" % (d, H, F, L, Ld, B, S, p)
        t0 = time.time()
        try:
            parity.check_step(backend, cfg, B, p, seed=k)
            parity.check_predict(backend, cfg, min(B, 4), True)
            # the sequence-resident kernels (csrc/gt_seq.h) run the whole backward in one launch: a single gradient bucket
            seq = Ld == 0 and d % 16 == 0 and d <= 128 and F % 16 == 0 and F <= 512 and S <= 32 and (hd < 16 or hd in (16, 32, 64))
            nbk = len(parity.Runner(cfg, min(B, 8), backend).lib.grad_buckets(parity.Runner(cfg, min(B, 8), backend).c))   # (riders: 2 at d_model 128)
            assert nbk == (2 if (Ld or L >= 2) and not seq else 1) or (seq and d == 128)
            parity.check_bucketed_backward(backend, cfg, min(B, 8), p, nbk, exact=False)
            if verbose:
                print("ok   %-40s %.1fs" % (tag, time.time() - t0), flush=True)
        except Exception as e:  # noqa: BLE001
            fails += 1
            print("FAIL %-40s %s: %s" % (tag, type(e).__name__, str(e)[:200]), flush=True)
    return fails


def run_riders(n, seed, backend="hip", verbose=True):
    """d_model-128 shapes through the SPLIT phases with rider weight gradients (gt_seq_wg.h): odd batches (odd slab counts per wave),
    dim_feedforward with partial column tiles, 27- / 5-wide inputs (packed / scalar staging in the tail), 1-4 layers (bucket cut)."""
    rnd = random.Random(1000 + seed)
    fails = 0
    for k in range(n):
        H = rnd.choice([2, 4, 8, 16, 32])
        F = rnd.choice([16, 48, 64, 112, 128, 320, 512])
        L = rnd.choice([1, 2, 3, 4])
        B = rnd.choice([1, 2, 3, 5, 9, 17, 33, 64, 80])
        S = rnd.choice([16, 27, 5])
        p = rnd.choice([0.0, 0.1, 0.3])
        mode = rnd.choice(["split", "split", "split-noride", True])
        cfg = cfg_dict(128, H, F, L, 0, embedding_size_src=S)
        tag = "riders d128 H%d F%d L%d B%d S%d p%.1f %s" % (H, F, L, B, S, p, mode)
        t0 = time.time()
        try:
            r, _, _, _ = parity.check_step(backend, cfg, B, p, seed=k, seq=mode)
            nb = len(r.lib.grad_buckets(r.c))
            parity.check_train_step(backend, cfg, min(B, 16), p, seq=mode)
            parity.check_bucketed_backward(backend, cfg, min(B, 16), p, nb if B <= 16 else len(parity.Runner(cfg, min(B, 16), backend, seq=mode).lib.grad_buckets(
                parity.Runner(cfg, min(B, 16), backend, seq=mode).c)), exact=False, seq=mode)
            if verbose:
                print("ok   %-50s %d buckets %.1fs" % (tag, nb, time.time() - t0), flush=True)
        except Exception as e:  # noqa: BLE001
            fails += 1
            print("FAIL %-50s %s: %s" % (tag, type(e).__name__, str(e)[:200]), flush=True)
    return fails


def run_mid(n, seed, backend="hip", verbose=True):
    """round 5: the 64x64 ring tiles (gt_gemm64.h; run with GT_T64R_MIN=1 so that every eligible problem takes them), the LayerNorm row exchange
    (forced on at random), fp32 / bf16 operands / precision 2, d_model 256 / 384 / 512, 64 ... 2048 tokens, encoder-only and encoder-decoder."""
    rnd = random.Random(2000 + seed)
    fails = 0
    lib = parity.Runner(cfg_dict(32, 4, 16, 1), 1, backend).lib
    for k in range(n):
        d = rnd.choice([256, 256, 384, 512, 512])
        H = rnd.choice([h for h in (2, 4, 8, 16) if (d // h) in (16, 24, 32, 48, 64, 96, 128, 192)])
        F = rnd.choice([128, 256, 384, 512, 640])
        L = rnd.choice([1, 2])
        Ld = rnd.choice([0, 0, 0, 1])
        B = rnd.choice([2, 4, 6, 8, 16, 64])
        S = rnd.choice([16, 27])
        p = rnd.choice([0.0, 0.1, 0.3])
        prec = rnd.choice([0, 0, 1, 2])
        xchg = rnd.choice([-1, 1])
        if B == 64 and (L + Ld > 2 or d == 384):
            L, Ld = 1, 0
        cfg = cfg_dict(d, H, F, L, Ld, embedding_size_src=S)
        tag = "mid d%d H%d F%d L%d+%d B%d S%d p%.1f prec %d xchg %d" % (d, H, F, L, Ld, B, S, p, prec, xchg)
        t0 = time.time()
        lib.cdll.gt_set_ln_exchange(xchg)
        try:
            if prec:
                r, _, _ = parity.check_step_bf16(backend, cfg, B, p, seed=k, precision=prec)
                tag += " (in force %d)" % r.precision_in_force()
            else:
                parity.check_step(backend, cfg, B, p, seed=k)
                if B <= 16:
                    parity.check_train_step(backend, cfg, B, p, seq=False)
            if verbose:
                print("ok   %-64s %.1fs" % (tag, time.time() - t0), flush=True)
        except Exception as e:  # noqa: BLE001
            fails += 1
            print("FAIL %-64s %s: %s" % (tag, type(e).__name__, str(e)[:200]), flush=True)
        finally:
            lib.cdll.gt_set_ln_exchange(-1)
    return fails


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=24)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--riders", action="store_true", help="d_model-128 SPLIT / rider shapes only")
    ap.add_argument("--mid", action="store_true", help="round 5: d_model 256 ... 512 through the 64x64 ring tiles / row exchange / precision 2 (set GT_T64R_MIN=1)")
    args = ap.parse_args()
    fails = run_mid(args.n, args.seed) if args.mid else run_riders(args.n, args.seed) if args.riders else run(args.n, args.seed)
    print("failures:", fails)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
