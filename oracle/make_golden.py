"""Generate tests/golden/*.npz (run HERE, in the build container; outputs are committed).

    python -m oracle.make_golden

What produces the expected values: stock ``torch.nn`` modules on CPU fp32 wired by
oracle/torch_groove.py (the third-party dependency the reference's arithmetic lives in), fed
 (a) the reference's own demo checkpoint (ref:demo/transformer_run_171tyqit_Epoch_1.Model) when
     /root/reference is present -> golden/demo_ckpt.npz (weights dump + forward on a seeded input), and
 (b) seeded parameter sets from oracle.numpy_groove.init_params -> golden/g2_*.npz (forward, loss
     stats for three penalties, all parameter gradients, parameters after one SGD and one Adam step).
Fixtures hold data only (inputs, weights, expected outputs).
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import numpy_groove as ng  # noqa: E402
from oracle import torch_groove as tg  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
CKPT = "/root/reference/demo/transformer_run_171tyqit_Epoch_1.Model"

G2 = {
    "enc_d32h4": dict(d_model=32, n_heads=4, dim_feedforward=16, num_encoder_layers=2, num_decoder_layers=0, embedding_size_src=16),
    "enc_d32h16_sym": dict(d_model=32, n_heads=16, dim_feedforward=64, num_encoder_layers=2, num_decoder_layers=0, embedding_size_src=27),
    "enc_d64h2": dict(d_model=64, n_heads=2, dim_feedforward=32, num_encoder_layers=2, num_decoder_layers=0, embedding_size_src=16),
    "encdec_d32h4": dict(d_model=32, n_heads=4, dim_feedforward=16, num_encoder_layers=2, num_decoder_layers=2, embedding_size_src=16),
    # BASELINE configs[1] shape (d128/H4/F512/L3); weights regenerated from the seed, grads kept as norms + samples
    "enc_c2": dict(d_model=128, n_heads=4, dim_feedforward=512, num_encoder_layers=3, num_decoder_layers=0, embedding_size_src=16),
}
# BASELINE configs[2] / [3] / [4] at FULL depth (batch 8): stock torch forward, loss, gradient norms + 64 samples per tensor
G2_FULL = {
    "encdec_c3": (dict(d_model=256, n_heads=2, dim_feedforward=512, num_encoder_layers=6, num_decoder_layers=6, embedding_size_src=16), 8),
    "enc_c4": (dict(d_model=512, n_heads=8, dim_feedforward=512, num_encoder_layers=6, num_decoder_layers=0, embedding_size_src=16), 8),
    "enc_c5_sym": (dict(d_model=512, n_heads=8, dim_feedforward=512, num_encoder_layers=6, num_decoder_layers=0, embedding_size_src=27), 8),
}


def load_into(model, P):
    sd = model.state_dict()
    for k, v in P.items():
        sd[k] = torch.tensor(v)
    model.load_state_dict(sd, strict=True)


def demo_ckpt():
    ck = torch.load(CKPT, weights_only=True, map_location="cpu")
    cfg = dict(d_model=32, n_heads=4, dim_feedforward=16, num_encoder_layers=6, num_decoder_layers=0,
               dropout=0.0, embedding_size_src=16)
    out = {"epoch": ck["epoch"], "loss": ck["loss"],
           "sgd_lr": ck["optimizer_state_dict"]["param_groups"][0]["lr"]}
    x, y = ng.synthetic_batch(4, 16, seed=7)
    out["x"], out["y"] = x, y
    for H in (4, 16):   # n_heads is not recoverable from the state dict; both load
        cfg["n_heads"] = H
        m = tg.build(cfg)
        print("strict load H=%d:" % H, m.load_state_dict(ck["model_state_dict"], strict=True))
        m.eval()
        with torch.no_grad():
            h, v, o = m(torch.tensor(x))
        out["h_H%d" % H], out["v_H%d" % H], out["o_H%d" % H] = h.numpy(), v.numpy(), o.numpy()
    for k, v in ck["model_state_dict"].items():
        out["sd/" + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "demo_ckpt.npz"), **out)


def g2(name, cfg, B=4, seed=11, with_predict=True):
    full = cfg["d_model"] <= 64
    cfg = dict(cfg, dropout=0.0)
    P = ng.init_params(cfg, seed=seed, perturb=0.05)
    x, y = ng.synthetic_batch(B, cfg["embedding_size_src"], seed=seed + 1)
    enc_only = cfg["num_decoder_layers"] == 0
    out = {"seed": seed, "x": x, "y": y, "cfg_keys": np.array(sorted(cfg)), "cfg_vals": np.array([float(cfg[k]) for k in sorted(cfg)])}
    m = tg.build(cfg)
    load_into(m, P)
    m.train()
    xt, yt = torch.tensor(x), torch.tensor(y)
    bce, mse = torch.nn.BCEWithLogitsLoss(reduction="none"), torch.nn.MSELoss(reduction="none")
    for pen in (1.0, 0.47, 0.0):
        m.zero_grad()
        pred = m(xt) if enc_only else m(xt, tg.shift_right(yt))
        st = tg.calculate_loss(pred, yt, bce, mse, pen)
        out["stats_pen%g" % pen] = np.array([st[0].item()] + list(st[1:]), np.float64)
    out["h"], out["v"], out["o"] = [t.detach().numpy() for t in pred]
    # gradients at penalty 0.47
    m.zero_grad()
    pred = m(xt) if enc_only else m(xt, tg.shift_right(yt))
    tg.calculate_loss(pred, yt, bce, mse, 0.47)[0].backward()
    grads = {k: p.grad.numpy().copy() for k, p in m.named_parameters()}
    r = np.random.default_rng(5)
    for k, g in grads.items():
        if full:
            out["grad/" + k] = g
        else:
            idx = r.integers(0, g.size, size=min(64, g.size))
            out["gidx/" + k], out["gval/" + k] = idx, g.reshape(-1)[idx]
            out["gnorm/" + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
    if full:
        for k, v in P.items():
            out["param/" + k] = v
        sgd = torch.optim.SGD(m.parameters(), lr=0.094)
        sgd.step()
        for k, p in m.named_parameters():
            out["sgd/" + k] = p.detach().numpy().copy()
        load_into(m, P)           # grads are kept
        adam = torch.optim.Adam(m.parameters(), lr=1e-3)
        adam.step()
        for k, p in m.named_parameters():
            out["adam/" + k] = p.detach().numpy().copy()
        load_into(m, P)
    if with_predict:
        ph, pv, po = m.predict(xt)
        out["pred_h"], out["pred_v"], out["pred_o"] = ph.numpy(), pv.numpy(), po.numpy()
        _, margin = ng.predict(P, cfg, x)
        out["pred_margin_min"] = margin.min()
    np.savez_compressed(os.path.join(OUT, "g2_%s.npz" % name), **out)
    print(name, "loss", out["stats_pen0.47"][0])


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(1)
    if os.path.exists(CKPT):
        demo_ckpt()
    only = sys.argv[1:]
    for n, c in G2.items():
        if not only or n in only:
            g2(n, c)
    torch.set_num_threads(8)
    for n, (c, B) in G2_FULL.items():
        if not only or n in only:
            g2(n, c, B=B, with_predict=False)
