"""Inference throughput of model.predict (ref:evaluator.py:173 hands the whole evaluation set over at once):
encoder-only = one forward + threshold; encoder-decoder = encoder once + 32 greedy decoder passes.
usage: python tools/predict_bench.py [--n 4096]   (one GPU)"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from transformergrooveinfilling_amd import layout as ng  # noqa: E402
from transformergrooveinfilling_amd.engine import StepEngine  # noqa: E402

SHAPES = [
    ("C2 d128/H4/F512/L3 enc-only", dict(d_model=128, n_heads=4, dim_feedforward=512, num_encoder_layers=3, num_decoder_layers=0, dropout=0.24)),
    ("K&S/Random yaml d256/H2/F512/L6 enc-only", dict(d_model=256, n_heads=2, dim_feedforward=512, num_encoder_layers=6, num_decoder_layers=0, dropout=0.3)),
    ("C4 d512/H8/F512/L6 enc-only", dict(d_model=512, n_heads=8, dim_feedforward=512, num_encoder_layers=6, num_decoder_layers=0, dropout=0.3)),
    ("C3 d256/H2/F512/L6+6 enc-dec (greedy, 32 decoder passes)", dict(d_model=256, n_heads=2, dim_feedforward=512, num_encoder_layers=6, num_decoder_layers=6, dropout=0.3)),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4096, help="sequences per predict call")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    for name, dims in SHAPES:
        dims = dict(dims, embedding_size_src=16)
        eng = StepEngine(batch_size=8, optimizer="sgd", learning_rate=0.05, hit_loss_penalty=0.5, seed=1, **dims)
        eng.load_named(ng.init_params(dims, seed=0))
        x, _ = ng.synthetic_batch(args.n, 16, seed=2)
        xd = torch.from_numpy(x).cuda()
        out = eng.predict(xd)                       # warm-up (code objects, allocator)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            out = eng.predict(xd)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        hits = float(out[:, :, :9].mean())
        print("%-58s N=%5d  %8.2f ms/call  %10.0f seq/s   (mean hit rate %.3f)" % (name, args.n, 1e3 * dt, args.n / dt, hits), flush=True)
        del eng
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
