"""Stress on the GPU (round 5, after the memset-node bug): every shape of tools/shape_bench.py plus encoder-decoder variants, the fused step REPLAYED AS
A hipGraph with a host synchronisation after every step, against the same steps enqueued eagerly -- parameters must stay finite and agree
(fp32 atomics in the weight gradients: last bits differ).  usage: python tools/graph_sync_stress.py [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from shape_bench import SHAPES  # noqa: E402
from transformergrooveinfilling_amd import layout  # noqa: E402
from transformergrooveinfilling_amd.engine import StepEngine  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
extra = [("enc-dec d64/H2/F64/L2+2 bs8", dict(d_model=64, n_heads=2, dim_feedforward=64, num_encoder_layers=2, num_decoder_layers=2, dropout=0.1), 8),
         ("enc-dec d128/H4/F256/L1+3 bs32", dict(d_model=128, n_heads=4, dim_feedforward=256, num_encoder_layers=1, num_decoder_layers=3, dropout=0.2), 32),
         ("enc-dec d512/H8/F512/L1+1 bs64", dict(d_model=512, n_heads=8, dim_feedforward=512, num_encoder_layers=1, num_decoder_layers=1, dropout=0.3), 64),
         ("enc-dec d256 bf16 L2+2 bs64", dict(d_model=256, n_heads=2, dim_feedforward=512, num_encoder_layers=2, num_decoder_layers=2, dropout=0.3, precision="bf16"), 64),
         ("adam d512/H8/F512/L2 bs64", dict(d_model=512, n_heads=8, dim_feedforward=512, num_encoder_layers=2, num_decoder_layers=0, dropout=0.3, optimizer="adam"), 64)]
bad = 0
for name, dims, B in list(SHAPES) + extra:
    dims = dict(dict(embedding_size_src=16), **dims)
    opt = dims.pop("optimizer", "sgd")
    if "bs512" in name or "bs256" in name and "C3" in name:
        dims = dict(dims, num_encoder_layers=min(2, dims["num_encoder_layers"]), num_decoder_layers=min(2, dims["num_decoder_layers"]))     # (time: the big shapes at two layers)
    x, y = layout.synthetic_batch(B, dims["embedding_size_src"], seed=2)
    out = []
    for graph in (True, False):
        eng = StepEngine(batch_size=B, optimizer=opt, learning_rate=0.02 if opt == "sgd" else 1e-3, hit_loss_penalty=0.5, seed=1, use_graph=graph, **dims)
        eng.load_named(layout.init_params(dims, seed=0))
        eng.x.copy_(torch.from_numpy(x)); eng.y.copy_(torch.from_numpy(y))
        ok = True
        for _ in range(steps):
            eng.train_step()
            if graph:
                torch.cuda.synchronize()
                ok = ok and bool(torch.isfinite(eng.params).all())
        torch.cuda.synchronize()
        out.append((eng.params.clone(), ok and bool(torch.isfinite(eng.params).all()), float(eng.stats[0])))
        del eng
    rel = float((out[0][0] - out[1][0]).abs().max() / out[1][0].abs().max())
    good = out[0][1] and out[1][1] and rel < 5e-3
    bad += not good
    print("%s %-70s finite %s/%s  max rel diff graph vs eager %.2e  loss %.4f / %.4f" % ("ok  " if good else "FAIL", name[:70], out[0][1], out[1][1], rel, out[0][2], out[1][2]), flush=True)
    torch.cuda.empty_cache()
print("failures:", bad)
sys.exit(1 if bad else 0)
