"""Step time of the HIP path for other shapes than the headline workload (C1..C4, the shipped YAML shapes).
usage: python tools/shape_bench.py [--steps 100]   (one GPU)"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from transformergrooveinfilling_amd import layout  # noqa: E402
from transformergrooveinfilling_amd.engine import StepEngine  # noqa: E402

SHAPES = [
    ("C1 HH_testing yaml d32/H4/F16/L6 bs32", dict(d_model=32, n_heads=4, dim_feedforward=16, num_encoder_layers=6, num_decoder_layers=0, dropout=0.18), 32),
    ("ClosedHH yaml d32/H16/F512/L6 bs16", dict(d_model=32, n_heads=16, dim_feedforward=512, num_encoder_layers=6, num_decoder_layers=0, dropout=0.24), 16),
    ("C2 d128/H4/F512/L3 bs64", dict(d_model=128, n_heads=4, dim_feedforward=512, num_encoder_layers=3, num_decoder_layers=0, dropout=0.24), 64),
    ("C2 shape bs256", dict(d_model=128, n_heads=4, dim_feedforward=512, num_encoder_layers=3, num_decoder_layers=0, dropout=0.24), 256),
    ("K&S/Random yaml d256/H2/F512/L6 bs32", dict(d_model=256, n_heads=2, dim_feedforward=512, num_encoder_layers=6, num_decoder_layers=0, dropout=0.3), 32),
    ("C3 enc-dec d256/H2/F512/L6+6 bs256", dict(d_model=256, n_heads=2, dim_feedforward=512, num_encoder_layers=6, num_decoder_layers=6, dropout=0.3), 256),
    ("C4 d512/H8/F512/L6 bs64 (per-GPU share of 512)", dict(d_model=512, n_heads=8, dim_feedforward=512, num_encoder_layers=6, num_decoder_layers=0, dropout=0.3), 64),
    ("C4 d512/H8/F512/L6 bs512", dict(d_model=512, n_heads=8, dim_feedforward=512, num_encoder_layers=6, num_decoder_layers=0, dropout=0.3), 512),
    ("C5 shape in fp32: symbolic input S=27, d512/H8/F512/L6 bs64", dict(d_model=512, n_heads=8, dim_feedforward=512, num_encoder_layers=6, num_decoder_layers=0, dropout=0.3, embedding_size_src=27), 64),
    ("C5 bf16 operands: symbolic S=27, d512/H8/F512/L6 bs64", dict(d_model=512, n_heads=8, dim_feedforward=512, num_encoder_layers=6, num_decoder_layers=0, dropout=0.3, embedding_size_src=27, precision="bf16"), 64),
    ("C5 bf16 operands: audio (MSO) S=16, d512/H8/F512/L6 bs64", dict(d_model=512, n_heads=8, dim_feedforward=512, num_encoder_layers=6, num_decoder_layers=0, dropout=0.3, precision="bf16"), 64),
    ("C5 bf16 operands, S=27, bs512 on one GPU", dict(d_model=512, n_heads=8, dim_feedforward=512, num_encoder_layers=6, num_decoder_layers=0, dropout=0.3, embedding_size_src=27, precision="bf16"), 512),
    ("C5 precision 2 (bf16 storage of the Linear outputs): S=27, d512/H8/F512/L6 bs64", dict(d_model=512, n_heads=8, dim_feedforward=512, num_encoder_layers=6, num_decoder_layers=0, dropout=0.3, embedding_size_src=27, precision="autocast"), 64),
    ("C5 precision 2, S=27, bs512 on one GPU", dict(d_model=512, n_heads=8, dim_feedforward=512, num_encoder_layers=6, num_decoder_layers=0, dropout=0.3, embedding_size_src=27, precision="autocast"), 512),
    ("RandomLow_lm yaml d256/H2/F2048/L8 bs32", dict(d_model=256, n_heads=2, dim_feedforward=2048, num_encoder_layers=8, num_decoder_layers=0, dropout=0.16), 32),
    ("Random_test_large yaml d256/H16/F64/L11 bs16", dict(d_model=256, n_heads=16, dim_feedforward=64, num_encoder_layers=11, num_decoder_layers=0, dropout=0.15), 16),
    # the reference CLI's own defaults (ref:train.py:43-62 = configs/hyperparameter_defaults.yaml: what `train.py --experiment X` runs without a --config)
    ("CLI defaults d64/H16/F256/L7 bs16", dict(d_model=64, n_heads=16, dim_feedforward=256, num_encoder_layers=7, num_decoder_layers=0, dropout=0.2), 16),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--only", type=int, default=-1, help="index of the single shape to run")
    ap.add_argument("--no-graph", action="store_true", help="eager launches (for rocprofv3 --pmc passes)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=0, help="override the batch size of the selected shape")
    args = ap.parse_args()
    for i, (name, dims, B) in enumerate(SHAPES):
        if args.only >= 0 and i != args.only:
            continue
        dims = dict(dict(embedding_size_src=16), **dims)
        if args.batch:
            B, name = args.batch, "%s [batch %d]" % (name, args.batch)
        eng = StepEngine(batch_size=B, optimizer="sgd", learning_rate=0.05, hit_loss_penalty=0.5, seed=1, use_graph=False if args.no_graph else "auto", **dims)
        eng.load_named(layout.init_params(dims, seed=0))
        x, y = layout.synthetic_batch(B, dims["embedding_size_src"], seed=2)
        eng.x.copy_(torch.from_numpy(x)); eng.y.copy_(torch.from_numpy(y))
        for _ in range(args.warmup):
            eng.train_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.train_step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        ft = bench.f_train_per_seq(dims)
        bf = dims.get("precision") in ("bf16", "autocast")
        peak = 2500.0 if bf else 157.3            # dense MFMA peak of the operand type (MI355X_MICROARCH.md)
        print("%-58s %8.3f ms/step %10.0f seq/s %7.2f TFLOP/s (%.1f %% of %s MFMA peak)  loss %.3f" %
              (name, 1e3 * dt, B / dt, B / dt * ft / 1e12, 100 * B / dt * ft / 1e12 / peak, "bf16" if bf else "fp32", float(eng.stats[0])), flush=True)
        del eng
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
