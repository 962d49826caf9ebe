"""Parity checks shared by the emulator tests (CPU, small) and the GPU tests (real HIP library).
Tolerances (fp32 path, stated per BASELINE.json north_star): outputs max-abs <= 1e-4 vs the fp64
oracle (we assert 2e-5), gradients <= 1e-3 relative to the tensor's max (we assert 2e-4), hit masks
bit-exact wherever |sigmoid(logit) - thres| exceeds 1e-4; ReLU decisions identical wherever the oracle's
|pre-activation| exceeds 1e-5 (adopt_device_kinks)."""
import glob
import os

import numpy as np

from harness import Runner
from oracle import numpy_groove as ng

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OUT_TOL, GRAD_TOL = 2e-5, 2e-4


def shift_right(y):
    return np.concatenate([np.zeros_like(y[:, :1]), y[:, :-1]], 1)


def rel_err(a, ref):
    return float(np.abs(a - ref).max() / (np.abs(ref).max() + 1e-12))


KINK = 1e-5


def adopt_device_kinks(r, C, cfg):
    """ReLU has no derivative at 0.  Where the fp64 oracle's pre-activation lies within KINK of the kink, the fp32 device
    may round to the other side (about one element per 1e6 pre-activations: BASELINE configs[1] at full size has 1e6 per
    layer) and both subgradients are legitimate -- the same reasoning as the hit-threshold margin above.  Take the
    device's decision for exactly those elements so that both backward passes differentiate the same function; a
    differing decision anywhere else is an error.  Returns the number of adopted elements."""
    n_enc, adopted = len(C["enc"]), 0
    todo = [(c, "hpre", c["mf"], r.ws_get("hact", l)) for l, c in enumerate(C["enc"])]
    todo += [(c, "hpre", c["mf"], r.ws_get("hact", n_enc + l)) for l, c in enumerate(C["dec"])]
    todo.append((C["in_enc"], "a", None, r.ws_get("a0")))
    for c, key, keep, dev in todo:
        pre = c[key]
        on = dev.reshape(pre.shape) > 0
        flips = on != (pre > 0)
        if keep is not None:
            flips &= keep != 0                      # dropped elements are zero on both sides whatever the sign
        if flips.any():
            assert (np.abs(pre[flips]) < KINK).all(), "ReLU decision differs away from the kink: max |pre| %g" % np.abs(pre[flips]).max()
            pre[flips] = np.where(on[flips], 1e-30, -1e-30)
            adopted += int(flips.sum())
    return adopted


def check_step(backend, cfg, B, p=0.0, penalty=0.47, seed=3, check_ws=True, chain=False):
    """forward + loss + backward (+ saved activations) against the fp64 numpy oracle."""
    cfg = dict(cfg, dropout=p)
    P = ng.init_params(cfg, seed=seed, perturb=0.05)
    x, y = ng.synthetic_batch(B, cfg["embedding_size_src"], seed=seed + 2)
    Ld = cfg.get("num_decoder_layers", 0)
    tgt = shift_right(y) if Ld else None
    rng = (1234, 99, 7)
    r = Runner(cfg, B, backend, rng=rng, chain=chain)
    r.set_params(P)
    hvo = r.forward(x, tgt, train=p > 0)
    (h, v, o), C = ng.forward(P, cfg, x, tgt=tgt, rng=rng if p > 0 else None, dtype=np.float64)
    ref = np.concatenate([h, v, o], -1)
    assert np.abs(hvo - ref).max() < OUT_TOL, "forward max-abs %g" % np.abs(hvo - ref).max()
    if check_ws:
        c0 = C["enc"][0]
        assert rel_err(r.ws_get("x0").reshape(-1), c0["x_in"].reshape(-1)) < 1e-5
        assert rel_err(r.ws_get("P", 0).reshape(-1), c0["attn"]["P"].reshape(-1)) < 1e-5
        assert rel_err(r.ws_get("hact", 0).reshape(-1), c0["hact"].reshape(-1)) < 1e-5
        assert rel_err(r.ws_get("memory").reshape(-1), C["memory"].reshape(-1)) < 1e-5
    stats, d_hvo = r.loss(y, penalty)
    rstats, dpred = ng.calculate_loss((h, v, o), y.astype(np.float64), penalty)
    for i in (0, 3, 4, 5):
        assert abs(stats[i] - rstats[i]) < 1e-5 * max(1.0, abs(rstats[i])), (i, stats[i], rstats[i])
    # hit accuracy counts (h > 0) == y_h: a logit within 1e-4 of 0 may fall on the other side in fp32, one count each
    near = int((np.abs(h) < 1e-4).sum())
    assert abs(stats[1] - rstats[1]) <= (near + 0.01) / h.size + 1e-6, (stats[1], rstats[1], near)
    assert rel_err(d_hvo, np.concatenate(dpred, -1)) < 1e-5
    G = r.backward(train=p > 0)
    adopted = adopt_device_kinks(r, C, cfg)
    assert adopted <= 4 + r.M * cfg["dim_feedforward"] * (len(C["enc"]) + len(C["dec"])) // 100000, adopted
    Gr = ng.backward(P, cfg, C, dpred, dtype=np.float64)
    assert set(G) == set(Gr)
    for k in Gr:
        # relative to the tensor's largest entry, with an absolute floor: a tensor that is numerically zero (|g| < 1e-5
        # everywhere -- e.g. LayerNorm gammas at d_model 2, where the normalised outputs are +-1 and the sum cancels; the
        # oracle's own fp32 run is 10 % off its fp64 run there) is compared absolutely
        err = float(np.abs(G[k] - Gr[k]).max() / max(np.abs(Gr[k]).max(), 1e-5))
        assert err < GRAD_TOL, (k, err)
    return r, P, G, Gr


# ---- bf16 operand path (gt_config.precision = 1, BASELINE configs[4]) -------------------------------------------------------
# Compared with the fp64 oracle fed bf16-ROUNDED operands (oracle.numpy_groove "bf16 operand mode": same rounding points as
# the device).  What is left between the two is (a) fp32 accumulation and (b) rounding-boundary flips: an operand the device
# computed 6e-8 (relative) away from the oracle's value rounds to the neighbouring bf16 with probability ~1.5e-5; one flip
# moves that row's outputs by ~1e-3 and the rest of its sequence with it.  Without a flip the two agree to 1e-7 (small
# shapes); at realistic sizes every sequence sees a few.  The bar is therefore RELATIVE TO THE bf16 EFFECT ITSELF:
#   E_q = RMS(fp32-operand oracle - bf16-operand oracle)         (what rounding the operands changes)
#   forward:   RMS(device - bf16 oracle) <= 0.35 E_q + 2e-6   and   max-abs <= 2e-2
#   gradients: per tensor RMS(device - bf16 oracle) <= 0.35 E_q(tensor) + 1e-4 RMS(g)   and   max-abs <= 2e-2 max|g|
#   loss: 2e-3 relative.
# A device that did not round (or rounded at other points) sits at ~1.0 E_q and fails.
BF16_FRAC, BF16_OUT_MAX, BF16_GRAD_MAX = 0.35, 2e-2, 2e-2


def _rms(a):
    return float(np.sqrt((np.asarray(a, np.float64) ** 2).mean()))


def check_step_bf16(backend, cfg, B, p=0.0, penalty=0.47, seed=3):
    cfg = dict(cfg, dropout=p, precision=1)
    cfg0 = dict(cfg, precision=0)
    P = ng.init_params(cfg, seed=seed, perturb=0.05)
    x, y = ng.synthetic_batch(B, cfg["embedding_size_src"], seed=seed + 2)
    Ld = cfg.get("num_decoder_layers", 0)
    tgt = shift_right(y) if Ld else None
    rng = (1234, 99, 7)
    r = Runner(cfg, B, backend, rng=rng)
    r.set_params(P)
    hvo = r.forward(x, tgt, train=p > 0)
    (h, v, o), C = ng.forward(P, cfg, x, tgt=tgt, rng=rng if p > 0 else None, dtype=np.float64)
    (h0, v0, o0), C0 = ng.forward(P, cfg0, x, tgt=tgt, rng=rng if p > 0 else None, dtype=np.float64)
    ref, ref0 = np.concatenate([h, v, o], -1), np.concatenate([h0, v0, o0], -1)
    eq = _rms(ref0 - ref)
    assert eq > 1e-5, "the bf16 rounding has no visible effect on this case: pick another"
    assert np.abs(hvo - ref).max() < BF16_OUT_MAX, "forward max-abs %g" % np.abs(hvo - ref).max()
    assert _rms(hvo - ref) <= BF16_FRAC * eq + 2e-6, "forward rms %g vs bf16 effect %g" % (_rms(hvo - ref), eq)
    stats, d_hvo = r.loss(y, penalty)
    rstats, dpred = ng.calculate_loss((h, v, o), y.astype(np.float64), penalty)
    _, dpred0 = ng.calculate_loss((h0, v0, o0), y.astype(np.float64), penalty)
    assert abs(stats[0] - rstats[0]) < 2e-3 * max(1.0, abs(rstats[0]))
    G = r.backward(train=p > 0)
    adopt_device_kinks(r, C, cfg)
    Gr = ng.backward(P, cfg, C, dpred, dtype=np.float64)
    G0 = ng.backward(P, cfg0, C0, dpred0, dtype=np.float64)
    for k in Gr:
        scale = max(np.abs(Gr[k]).max(), 1e-5)
        assert np.abs(G[k] - Gr[k]).max() / scale < BF16_GRAD_MAX, (k, np.abs(G[k] - Gr[k]).max() / scale)
        assert _rms(G[k] - Gr[k]) <= BF16_FRAC * _rms(G0[k] - Gr[k]) + 1e-4 * _rms(Gr[k]) + 1e-9, \
            (k, _rms(G[k] - Gr[k]), _rms(G0[k] - Gr[k]), _rms(Gr[k]))
    return r, P, G, Gr


def check_train_step_bf16(backend, cfg, B, p):
    """gt_train_step with precision = 1: two SGD steps against the bf16-operand oracle (fp32 master weights)."""
    cfg = dict(cfg, dropout=p, precision=1)
    P = ng.init_params(cfg, seed=9, perturb=0.05)
    x, y = ng.synthetic_batch(B, cfg["embedding_size_src"], seed=4)
    tgt = shift_right(y) if cfg.get("num_decoder_layers", 0) else None
    r = Runner(cfg, B, backend, rng=(77, 5, 0), lr=0.05)
    r.set_params(P)
    cur = {k: v.astype(np.float64) for k, v in P.items()}
    for step in range(2):
        stats = r.train_step(x, y, 0.38, algo=0)
        (h, v, o), C = ng.forward(cur, cfg, x, tgt=tgt, rng=(77, 5, step) if p > 0 else None, dtype=np.float64)
        rstats, dpred = ng.calculate_loss((h, v, o), y.astype(np.float64), 0.38)
        assert abs(stats[0] - rstats[0]) < 1e-3 * max(1, abs(rstats[0])), (step, stats[0], rstats[0])
        G = ng.backward(cur, cfg, C, dpred, dtype=np.float64)
        cur = {k: cur[k] - 0.05 * G[k] for k in cur}
        got = r.unflatten(r.params.numpy())
        for k in cur:
            assert np.abs(got[k] - cur[k]).max() < 0.05 * BF16_GRAD_MAX * max(np.abs(G[k]).max(), 1e-5) * (step + 1) + 1e-6, (step, k)
    assert r.step_state().step == 2


def check_optimizers(backend, cfg, B):
    cfg = dict(cfg, dropout=0.0)
    r, P, G, Gr = check_step(backend, cfg, B, check_ws=False)
    Gf = {k: G[k].astype(np.float32) for k in G}
    new = r.optimizer_step(0)
    want = ng.sgd_step(P, Gf, 0.094)
    for k in P:
        assert np.abs(new[k] - want[k]).max() < 1e-6, k
    assert r.step_state().step == 8
    # adam from the SGD-updated point with the same grads, two steps (bias correction uses step+1 relative to start)
    r2 = Runner(cfg, B, backend, rng=(1, 2, 0), lr=1e-3)
    r2.set_params(P)
    r2.grads = r2.Buf(r2.flatten(Gf))
    m = {k: np.zeros_like(P[k]) for k in P}
    v = {k: np.zeros_like(P[k]) for k in P}
    cur = P
    for t in (1, 2):
        got = r2.optimizer_step(1)
        cur, m, v = ng.adam_step(cur, Gf, m, v, t, 1e-3)
        for k in P:
            assert np.abs(got[k] - cur[k]).max() < 2e-6, (t, k)


def check_train_step(backend, cfg, B, p, algo=0, chain=False):
    """gt_train_step == forward+loss+backward+update with the oracle's masks; second step uses step+1."""
    cfg = dict(cfg, dropout=p)
    P = ng.init_params(cfg, seed=9, perturb=0.05)
    x, y = ng.synthetic_batch(B, cfg["embedding_size_src"], seed=4)
    Ld = cfg.get("num_decoder_layers", 0)
    tgt = shift_right(y) if Ld else None
    r = Runner(cfg, B, backend, rng=(77, 5, 0), lr=0.05, chain=chain)
    r.set_params(P)
    cur = {k: v.astype(np.float64) for k, v in P.items()}
    for step in range(2):
        stats = r.train_step(x, y, 0.38, algo=0)
        (h, v, o), C = ng.forward(cur, cfg, x, tgt=tgt, rng=(77, 5, step) if p > 0 else None, dtype=np.float64)
        rstats, dpred = ng.calculate_loss((h, v, o), y.astype(np.float64), 0.38)
        assert abs(stats[0] - rstats[0]) < 2e-5 * max(1, abs(rstats[0])), (step, stats[0], rstats[0])
        G = ng.backward(cur, cfg, C, dpred, dtype=np.float64)
        cur = {k: cur[k] - 0.05 * G[k] for k in cur}
        got = r.unflatten(r.params.numpy())
        for k in cur:
            assert np.abs(got[k] - cur[k]).max() < 2e-5 * max(1.0, np.abs(cur[k]).max()), (step, k)
    assert r.step_state().step == 2


def check_predict(backend, cfg, B, use_thres=True, thres=0.5, out_tol=OUT_TOL, margin_tol=1e-4):
    """out_tol / margin_tol: the fp32 bars by default; the bf16 operand path passes its own (hits compared where the decision
    margin exceeds what a bf16 rounding flip can move a probability by)."""
    cfg = dict(cfg, dropout=0.3)      # predict is eval mode: dropout must be ignored
    P = ng.init_params(cfg, seed=21, perturb=0.05)
    x, _ = ng.synthetic_batch(B, cfg["embedding_size_src"], seed=8)
    r = Runner(cfg, B, backend)
    r.set_params(P)
    hvo = r.predict(x, thres=thres, use_thres=use_thres)
    (h, v, o), margin = ng.predict(P, cfg, x, use_thres=use_thres, thres=thres, dtype=np.float64)
    if use_thres:
        sure = margin > margin_tol
        if cfg.get("num_decoder_layers", 0):
            # greedy decoding: a flipped low-margin hit changes every LATER step of that sequence, so each sequence is
            # compared up to (not including) its first step with a decision inside the margin -- bit-exact hits, v / o
            # within tolerance; sequences that are sure everywhere are compared whole
            vo = np.concatenate([v, o], -1)
            compared = 0
            for b in range(B):
                unsure = np.flatnonzero(~sure[b].reshape(32, -1).all(1))
                t_end = int(unsure[0]) if len(unsure) else 32
                compared += t_end
                assert np.array_equal(hvo[b, :t_end, :9], h[b, :t_end]), (b, t_end)
                if t_end:
                    assert np.abs(hvo[b, :t_end, 9:] - vo[b, :t_end]).max() < out_tol, (b, t_end)
            assert compared > 0, "no comparable decode step: every sequence has a low-margin decision at step 0"
            return
        assert np.array_equal(hvo[..., :9][sure], h[sure])                      # bit-exact hit mask
        assert set(np.unique(hvo[..., :9])) <= {0.0, 1.0}
    else:
        assert np.abs(hvo[..., :9] - h).max() < out_tol
    assert np.abs(hvo[..., 9:] - np.concatenate([v, o], -1)).max() < out_tol


def check_golden(backend, path):
    """HIP/emulated path against the committed torch-generated vectors."""
    z = np.load(path)
    cfg = {k: (float(v) if k == "dropout" else int(v)) for k, v in zip(z["cfg_keys"], z["cfg_vals"])}
    P = ng.init_params(cfg, seed=int(z["seed"]), perturb=0.05)
    x, y = z["x"], z["y"]
    B = x.shape[0]
    tgt = shift_right(y) if cfg["num_decoder_layers"] else None
    r = Runner(cfg, B, backend, lr=0.094)
    r.set_params(P)
    hvo = r.forward(x, tgt)
    assert np.abs(hvo - np.concatenate([z["h"], z["v"], z["o"]], -1)).max() < OUT_TOL
    for pen in (1.0, 0.0, 0.47):
        stats, _ = r.loss(y, pen)
        ref = z["stats_pen%g" % pen]
        for i in (0, 1, 3, 4, 5):
            assert abs(stats[i] - ref[i]) < 2e-5 * max(1.0, abs(ref[i])), (pen, i)
        assert abs(np.exp(stats[3]) - ref[2]) < 1e-4 * ref[2]                   # perplexity = exp(bce)
    G = r.backward()
    for k in G:
        if "grad/" + k in z.files:
            assert rel_err(G[k], z["grad/" + k]) < GRAD_TOL, k
        else:
            idx, val = z["gidx/" + k], z["gval/" + k]
            assert np.abs(G[k].reshape(-1)[idx] - val).max() <= GRAD_TOL * np.abs(val).max() + 1e-7, k
            assert abs(np.sqrt((G[k].astype(np.float64) ** 2).sum()) - float(z["gnorm/" + k])) < 1e-4 * float(z["gnorm/" + k])
    if "sgd/OutputLayer.Linear.bias" in z.files:
        new = r.optimizer_step(0)
        for k in new:
            assert np.abs(new[k] - z["sgd/" + k]).max() < 1e-5, k


def check_demo_ckpt(backend):
    z = np.load(os.path.join(GOLD, "demo_ckpt.npz"))
    P = {k[3:]: z[k] for k in z.files if k.startswith("sd/") and not k.endswith(".pe")}
    for H in (4, 16):
        cfg = dict(d_model=32, n_heads=H, dim_feedforward=16, num_encoder_layers=6, num_decoder_layers=0, embedding_size_src=16)
        r = Runner(cfg, 4, backend)
        r.set_params(P)
        hvo = r.forward(z["x"])
        ref = np.concatenate([z["h_H%d" % H], z["v_H%d" % H], z["o_H%d" % H]], -1)
        assert np.abs(hvo - ref).max() < OUT_TOL


def golden_files():
    return sorted(glob.glob(os.path.join(GOLD, "g2_*.npz")))


def check_bucketed_backward(backend, cfg, B, p, n_buckets, exact):
    """gt_train_step(skip_update=2) must leave bucket 0 of gt_grad_buckets FINAL (that is what the data-parallel path
    all-reduces while skip_update=3 runs), and 2 followed by 3 must equal the one-call backward (skip_update=1)."""
    cfg = dict(cfg, dropout=p)
    P = ng.init_params(cfg, seed=11, perturb=0.05)
    x, y = ng.synthetic_batch(B, cfg["embedding_size_src"], seed=12)
    whole = Runner(cfg, B, backend, rng=(5, 6, 0))
    whole.set_params(P)
    whole.train_step(x, y, 0.47, skip_update=1)
    gw = whole.grads.numpy().copy()
    r = Runner(cfg, B, backend, rng=(5, 6, 0))
    r.set_params(P)
    buckets = r.lib.grad_buckets(r.c)
    assert len(buckets) == n_buckets
    assert sum(c for _, c in buckets) == r.total and min(o for o, _ in buckets) == 0
    r.train_step(x, y, 0.47, skip_update=2)
    g2 = r.grads.numpy().copy()
    r.train_step(x, y, 0.47, skip_update=3)
    g3 = r.grads.numpy().copy()
    tol = 0.0 if exact else 2e-6 * np.abs(gw).max()
    o0, c0 = buckets[0]
    assert np.abs(g2[o0:o0 + c0] - gw[o0:o0 + c0]).max() <= tol, "bucket 0 not final after the first half"
    assert np.abs(g3 - gw).max() <= tol, "two halves differ from the whole backward"
    if n_buckets == 2:
        o1, c1 = buckets[1]
        assert np.abs(gw[o1:o1 + c1]).max() > 0 and np.abs(g2[o1:o1 + c1] - gw[o1:o1 + c1]).max() > 0   # the first half really stops early
