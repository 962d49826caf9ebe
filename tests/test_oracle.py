"""The oracle against its pins: the reference's demo checkpoint and the golden vectors that stock
torch modules produced (oracle/make_golden.py).  CPU only."""
import glob
import os

import numpy as np
import pytest

from oracle import numpy_groove as ng

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_g2(path):
    z = np.load(path)
    cfg = {k: (float(v) if k == "dropout" else int(v)) for k, v in zip(z["cfg_keys"], z["cfg_vals"])}
    return z, cfg


def test_demo_checkpoint_forward():
    z = np.load(os.path.join(GOLD, "demo_ckpt.npz"))
    P = {k[3:]: z[k] for k in z.files if k.startswith("sd/")}
    pe = P.pop("InputLayerEncoder.PositionalEncoding.pe")
    assert pe.shape == (1, 32, 32)
    assert np.abs(pe[0] - ng.positional_encoding(32)).max() < 1e-6          # buffer == recomputed sinusoid
    assert float(z["sgd_lr"]) == pytest.approx(0.094) and int(z["epoch"]) == 0
    for H in (4, 16):
        cfg = dict(d_model=32, n_heads=H, dim_feedforward=16, num_encoder_layers=6, num_decoder_layers=0,
                   embedding_size_src=16)
        assert [n for n, _ in ng.param_names(cfg)] == list(P.keys())          # state-dict order
        (h, v, o), _ = ng.forward(P, cfg, z["x"])
        for a, k in ((h, "h"), (v, "v"), (o, "o")):
            assert np.abs(a - z["%s_H%d" % (k, H)]).max() < 2e-5


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "g2_*.npz"))))
def test_numpy_oracle_vs_torch_golden(path):
    z, cfg = load_g2(path)
    P = ng.init_params(cfg, seed=int(z["seed"]), perturb=0.05)
    if "param/OutputLayer.Linear.bias" in z.files:                             # seeded init is reproducible
        for k in P:
            assert np.array_equal(P[k], z["param/" + k])
    x, y = z["x"], z["y"]
    enc_only = cfg["num_decoder_layers"] == 0
    tgt = None if enc_only else np.concatenate([np.zeros_like(y[:, :1]), y[:, :-1]], 1)
    (h, v, o), C = ng.forward(P, cfg, x, tgt=tgt)
    for a, k in ((h, "h"), (v, "v"), (o, "o")):
        assert np.abs(a - z[k]).max() < 2e-5
    for pen in (1.0, 0.47, 0.0):
        st, dpred = ng.calculate_loss((h, v, o), y, pen)
        ref = z["stats_pen%g" % pen]
        assert np.allclose(np.array(st, np.float64), ref, rtol=2e-5, atol=1e-6)
    st, dpred = ng.calculate_loss((h, v, o), y, 0.47)
    G = ng.backward(P, cfg, C, dpred)
    for k in P:
        if "grad/" + k in z.files:
            ref = z["grad/" + k]
            assert np.abs(G[k] - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-7, k
        else:
            idx, val = z["gidx/" + k], z["gval/" + k]
            assert np.abs(G[k].reshape(-1)[idx] - val).max() <= 2e-4 * np.abs(val).max() + 1e-7, k
            assert np.sqrt((G[k].astype(np.float64) ** 2).sum()) == pytest.approx(float(z["gnorm/" + k]), rel=1e-4)
    if "sgd/OutputLayer.Linear.bias" in z.files:
        Ps = ng.sgd_step(P, G, 0.094)
        Pa, _, _ = ng.adam_step(P, G, {k: 0 * P[k] for k in P}, {k: 0 * P[k] for k in P}, 1, 1e-3)
        for k in P:
            assert np.abs(Ps[k] - z["sgd/" + k]).max() < 1e-5, k
            live = np.abs(G[k]) > 1e-6        # Adam divides by |g|: a ~0 gradient (softmax-invariant key bias) is all noise
            assert np.abs(Pa[k] - z["adam/" + k])[live].max(initial=0) < 2e-5, k
    if "pred_h" not in z.files:                                                # full-size goldens (make_golden.py G2_FULL) carry no predict()
        return
    (ph, pv, po), margin = ng.predict(P, cfg, x)
    sure = margin > 1e-4
    assert np.array_equal(ph[sure], z["pred_h"][sure])                          # hit mask bit-exact
    if enc_only:                                                                # (greedy decode can diverge after a flipped hit)
        assert np.abs(pv - z["pred_v"]).max() < 2e-5 and np.abs(po - z["pred_o"]).max() < 2e-5


@pytest.mark.skipif(not os.path.exists("/root/reference/demo/transformer_run_171tyqit_Epoch_1.Model"),
                    reason="reference checkpoint not present on this box")
def test_reference_checkpoint_strict_loads_into_torch_restatement():
    import torch
    from oracle import torch_groove as tg
    ck = torch.load("/root/reference/demo/transformer_run_171tyqit_Epoch_1.Model", weights_only=True, map_location="cpu")
    m = tg.build(dict(d_model=32, n_heads=4, dim_feedforward=16, num_encoder_layers=6, num_decoder_layers=0))
    res = m.load_state_dict(ck["model_state_dict"], strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert len(list(m.parameters())) == len(ck["optimizer_state_dict"]["param_groups"][0]["params"]) == 78


def test_dropout_hash_statistics():
    m = ng.keep_mask((1234, 99, 3), ng.layer_site(2, ng.S_FFN), 200000, 0.24)
    keep = (m > 0).mean()
    assert abs(keep - 0.76) < 0.005
    assert np.allclose(m[m > 0], 1 / (1 - float(np.float32(0.24))))
    m2 = ng.keep_mask((1234, 99, 4), ng.layer_site(2, ng.S_FFN), 200000, 0.24)
    assert abs(((m > 0) == (m2 > 0)).mean() - (0.76 ** 2 + 0.24 ** 2)) < 0.01   # steps are independent
