"""Parity checks shared by the emulator tests (CPU, small) and the GPU tests (real HIP library).
Tolerances (fp32 path, stated per BASELINE.json north_star): outputs max-abs <= 1e-4 vs the fp64
oracle (we assert 2e-5), gradients <= 1e-3 relative to the tensor's max (we assert 2e-4), hit masks
bit-exact wherever |sigmoid(logit) - thres| exceeds 1e-4; ReLU decisions identical wherever the oracle's
|pre-activation| exceeds 1e-5 (adopt_device_kinks)."""
import glob
import os

import ctypes

import numpy as np

import harness
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


def check_step(backend, cfg, B, p=0.0, penalty=0.47, seed=3, check_ws=True, seq=True):
    """forward + loss + backward (+ saved activations) against the fp64 numpy oracle."""
    cfg = dict(cfg, dropout=p)
    P = ng.init_params(cfg, seed=seed, perturb=0.05)
    x, y = ng.synthetic_batch(B, cfg["embedding_size_src"], seed=seed + 2)
    Ld = cfg.get("num_decoder_layers", 0)
    tgt = shift_right(y) if Ld else None
    rng = (1234, 99, 7)
    r = Runner(cfg, B, backend, rng=rng, seq=seq)
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
# Two bars.
# (1) PER OPERATION, teacher-forced (check_ops_bf16): every Linear of the forward and of the backward is recomputed in fp64
#     from the DEVICE's own saved inputs (workspace activations / gradient temporaries), its two operands rounded to bf16 at
#     exactly the points the device rounds them (oracle.numpy_groove.round_bf16), and compared with the device's output of
#     that one operation.  Nothing propagates, so the bar is the fp32 one: 5e-5 of the tensor's largest entry (fp32
#     accumulation over up to 16384 terms), LayerNorm / activation epilogues included.  This is the parity proof.
# (2) END TO END against the bf16-operand oracle (oracle "bf16 operand mode").  Here differences DO propagate: an operand the
#     device computed 6e-8 (relative) away from the oracle's value rounds to the neighbouring bf16 now and then, that moves
#     its row by ~1e-3, and from there on the rest of the sequence rounds differently too -- two legitimate realisations of
#     the same rounding noise decorrelate.  So this bar is a sanity bound, relative to the bf16 effect itself
#     E_q = RMS(fp32-operand oracle - bf16-operand oracle):  RMS(device - bf16 oracle) <= 1.5 E_q + 1e-5, max-abs <= 5e-2,
#     loss within 1 %.
BF16_OP_TOL, BF16_E2E_FRAC, BF16_OUT_MAX = 5e-5, 1.5, 5e-2


def _rms(a):
    return float(np.sqrt((np.asarray(a, np.float64) ** 2).mean()))


def _close(dev, ref, what, tol=BF16_OP_TOL, where=None, stored16=False, rms_tol=None):
    """stored16: the device keeps this tensor in bf16 alone (operand-only tensors, gt_set_operand_shadows level 2) -- what is read back
    is the RNE rounding of the value the bar applies to, so each element may additionally be off by half a bf16 ulp of itself (2^-9;
    2^-8 allowed: the fp32 value may sit a hair on the other side of a rounding boundary)."""
    dev, ref = np.asarray(dev, np.float64).reshape(ref.shape), np.asarray(ref, np.float64)
    err = np.abs(dev - ref)
    if stored16:
        err = np.maximum(err - np.abs(ref) * 2.0 ** -8, 0.0)
    if where is not None:
        err = err * where
    scale = max(float(np.abs(ref).max()), 1e-6)
    assert err.max() <= tol * scale, "%s: max |err| %.3g of max |ref| %.3g (ratio %.3g)" % (what, err.max(), scale, err.max() / scale)
    if rms_tol is not None:      # behind a HIDDEN bf16 rounding single elements may sit one ulp off (max bar 2^-8-ish); on average nothing may
        assert _rms(err) <= rms_tol * scale, "%s: rms err %.3g of max |ref| %.3g (ratio %.3g)" % (what, _rms(err), scale, _rms(err) / scale)


def check_ops_bf16(r, P, cfg, x, tgt, rng, p, G=None):
    """Teacher-forced per-operation parity of the bf16 path (bar (1) above).  r: Runner after forward (and, with G = the device's
    gradients, after loss + backward).  Returns the number of operations checked."""
    rb = ng.round_bf16
    f64 = lambda a: np.asarray(a, np.float64)
    P = {k: f64(v) for k, v in P.items()}
    d, F, H = cfg["d_model"], cfg["dim_feedforward"], cfg["n_heads"]
    L, Ld = cfg["num_encoder_layers"], cfg.get("num_decoder_layers", 0)
    M = r.M
    pe = np.tile(f64(ng.positional_encoding(d)), (M // 32, 1))
    scale = 1.0 / (1.0 - float(np.float32(p))) if p > 0 else 1.0

    def mask(site, n, shape=None):
        m = ng.keep_mask(rng, site, n, p) if p > 0 else None
        return 1.0 if m is None else (m if shape is None else m.reshape(shape))

    def ws(name, layer=0, cols=None):
        a = f64(r.ws_get(name, layer))
        return a.reshape(M, -1) if cols is None else a.reshape(-1, cols)

    def lin(a, w, b=None):
        y = rb(a) @ rb(w).T
        return y if b is None else y + b

    def ln(z, g, b):
        mu = z.mean(-1, keepdims=True)
        var = ((z - mu) ** 2).mean(-1, keepdims=True)
        xh = (z - mu) / np.sqrt(var + 1e-5)
        return xh * g + b, xh

    def ln_bwd(dy, xh, rstd, g):
        gdy = dy * g
        return rstd * (gdy - gdy.mean(-1, keepdims=True) - xh * (gdy * xh).mean(-1, keepdims=True)), (dy * xh).sum(0), dy.sum(0)

    # precision 2 (encoder layers): Linear outputs are STORED in bf16 -- qkv and dqkv observably (stored16), the out-proj / linear2 outputs
    # ahead of their LayerNorm and the dgrad outputs ahead of a LayerNorm backward only transiently: the oracle rounds the same value (hrb)
    # and an element whose fp32 and fp64 values straddle a rounding boundary lands one bf16 ulp apart -- max bar 2^-7 of the tensor's
    # largest entry, RMS bar 2e-4 (the flips are rare)
    P2 = r.precision_in_force() == 2
    hrb = (lambda a: rb(a)) if P2 else (lambda a: a)
    HID = dict(tol=2.0 ** -7, rms_tol=2e-4) if P2 else {}
    heads, hd = cfg["n_heads"], cfg["d_model"] // cfg["n_heads"]

    def split_heads(a):                                   # (M, d) -> (B, H, 32, hd)
        return a.reshape(M // 32, 32, heads, hd).transpose(0, 2, 1, 3)

    def merge_heads(a):
        return a.transpose(0, 2, 1, 3).reshape(M, heads * hd)

    n = 0
    # ---------------------------------------------------------------------------------------------- forward
    def input_layer(xin, pre, a0name, outname, site):
        nonlocal n
        a = lin(f64(xin).reshape(M, -1), P[pre + "weight"], P[pre + "bias"])
        safe = np.abs(a) > 1e-4
        _close(ws(a0name), a, pre + "pre-activation")
        _close(ws(outname), (np.maximum(a, 0) + pe) * mask(site, a.size, a.shape), pre + "output", where=safe)
        n += 2

    def attn_block(name, gl, xin, inw, qkvname, ctxname, outname, xhname, normname, site, q_rows=None):
        nonlocal n
        enc = P2 and gl < L
        out = lin(ws(ctxname, gl), P[name + "out_proj.weight"], P[name + "out_proj.bias"])
        out = (hrb(out) if enc else out) * mask(site, M * d, (M, d))
        y, xh = ln(xin + out, P[normname + ".weight"], P[normname + ".bias"])
        _close(ws(outname, gl), y, "%s out-proj + norm (layer %d)" % (name, gl), **(HID if enc else dict(tol=2e-5)))
        _close(ws(xhname, gl), xh, "%s xhat (layer %d)" % (name, gl), **(HID if enc else dict(tol=2e-5)))
        n += 2
        return ws(outname, gl)

    def ffn_block(pre, gl, xin, normname, outname="xout"):
        nonlocal n
        hp = lin(xin, P[pre + "linear1.weight"], P[pre + "linear1.bias"])
        _close(ws("hact", gl), np.maximum(hp, 0) * mask(ng.layer_site(gl, ng.S_FFN), hp.size, hp.shape),
               pre + "linear1", where=np.abs(hp) > 1e-4, stored16=r.bf16_only("hact", gl))
        enc = P2 and gl < L - 1          # (the top encoder layer's linear2 feeds the two-norm pass from fp32)
        f = lin(ws("hact", gl), P[pre + "linear2.weight"], P[pre + "linear2.bias"])
        f = (hrb(f) if enc else f) * mask(ng.layer_site(gl, ng.S_DROPF), M * d, (M, d))
        y, xh = ln(xin + f, P[pre + normname + ".weight"], P[pre + normname + ".bias"])
        _close(ws(outname, gl), y, pre + "linear2 + norm", **(HID if enc else dict(tol=2e-5)))
        n += 2
        return ws(outname, gl)

    input_layer(x, "InputLayerEncoder.Linear.", "a0", "x0", ng.SITE_PE_ENC)
    cur = ws("x0")
    enc_in = []
    for l in range(L):
        pre = "Encoder.Encoder.layers.%d." % l
        enc_in.append(cur)
        _close(ws("qkv", l), lin(cur, P[pre + "self_attn.in_proj_weight"], P[pre + "self_attn.in_proj_bias"]), pre + "in_proj",
               stored16=r.bf16_only("qkv", l))
        n += 1
        if P2:      # the attention core on bf16-STORED q / k / v (fp32 arithmetic; attn_fwd_lds_kernel): probabilities and ctx
            qkv = ws("qkv", l)
            q_, k_, v_ = (split_heads(qkv[:, i * d:(i + 1) * d]) for i in range(3))
            sc = q_ @ k_.transpose(0, 1, 3, 2) / np.sqrt(hd)
            pr = np.exp(sc - sc.max(-1, keepdims=True)); pr /= pr.sum(-1, keepdims=True)
            _close(f64(r.ws_get("P", l)).reshape(pr.shape), pr, pre + "attention probabilities over bf16-stored q / k", tol=2e-5)
            pm = pr * mask(ng.layer_site(l, ng.S_ATTN), pr.size, pr.shape)
            _close(ws("ctx", l), merge_heads(pm @ v_), pre + "attention output over bf16-stored v", stored16=True)
            n += 2
        x1 = attn_block(pre + "self_attn.", l, cur, None, "qkv", "ctx", "x1", "xhat1", pre + "norm1", ng.layer_site(l, ng.S_DROP1))
        cur = ffn_block(pre, l, x1, "norm2")
    mem, _ = ln(cur, P["Encoder.Encoder.norm.weight"], P["Encoder.Encoder.norm.bias"])
    _close(ws("memory"), mem, "final encoder norm", tol=2e-5)
    final = ws("memory")
    dec_in = []
    if Ld:
        input_layer(tgt, "InputLayerDecoder.Linear.", "b0", "y0", ng.SITE_PE_DEC)
        ycur = ws("y0")
        for l in range(Ld):
            pre, gl = "Decoder.Decoder.layers.%d." % l, L + l
            dec_in.append(ycur)
            _close(ws("qkv", gl), lin(ycur, P[pre + "self_attn.in_proj_weight"], P[pre + "self_attn.in_proj_bias"]), pre + "self in_proj")
            y1 = attn_block(pre + "self_attn.", gl, ycur, None, "qkv", "ctx", "x1", "xhat1", pre + "norm1", ng.layer_site(gl, ng.S_DROP1))
            wx, bx = P[pre + "multihead_attn.in_proj_weight"], P[pre + "multihead_attn.in_proj_bias"]
            _close(ws("qx", gl), lin(y1, wx[:d], bx[:d]), pre + "cross q in_proj")
            _close(ws("kvx", gl), lin(final, wx[d:], bx[d:]), pre + "cross kv in_proj")
            n += 3
            y2 = attn_block(pre + "multihead_attn.", gl, y1, None, "qx", "ctxx", "x2", "xhatx", pre + "norm2", ng.layer_site(gl, ng.S_DROP2))
            ycur = ffn_block(pre, gl, y2, "norm3")
        fin, _ = ln(ycur, P["Decoder.Decoder.norm.weight"], P["Decoder.Decoder.norm.bias"])
        _close(ws("dec_final"), fin, "final decoder norm", tol=2e-5)
        final = ws("dec_final")
    logits = lin(final, P["OutputLayer.Linear.weight"], P["OutputLayer.Linear.bias"])
    want = np.concatenate([logits[:, :9], 1 / (1 + np.exp(-logits[:, 9:18])), 0.5 * np.tanh(logits[:, 18:])], 1)
    _close(r.hvo.numpy().reshape(M, 27), want, "output layer + heads")
    n += 1
    if G is None:
        return n
    # ---------------------------------------------------------------------------------------------- backward
    G = {k: f64(v) for k, v in G.items()}

    def tmp(name, gl, cols):
        return f64(r.ws_get(name if (p > 0 or not name.endswith("m")) else name[:-1], gl)).reshape(M, cols)

    def wgrad(dy, xin, wname, bname):
        nonlocal n
        _close(G[wname], rb(dy).T @ rb(xin), "grad " + wname)
        _close(G[bname], rb(dy).sum(0), "grad " + bname)
        n += 2

    dlog = ws("dlogits")
    wgrad(dlog, final, "OutputLayer.Linear.weight", "OutputLayer.Linear.bias")

    def ffn_bwd(pre, gl, xin, dz_name, prev_xh, prev_rstd, prev_norm, dzprev_name):
        nonlocal n
        dz, dzm = tmp(dz_name, gl, d), tmp(dz_name + "m", gl, d)
        wgrad(dzm, ws("hact", gl), pre + "linear2.weight", pre + "linear2.bias")
        dh = (rb(dzm) @ rb(P[pre + "linear2.weight"])) * np.where(ws("hact", gl) != 0, scale, 0.0)
        _close(tmp("dhid", gl, F), dh, pre + "linear2 dgrad (relu / dropout mask)", stored16=r.bf16_only("dhid", gl))
        dhid = tmp("dhid", gl, F)
        wgrad(dhid, xin, pre + "linear1.weight", pre + "linear1.bias")
        enc = P2 and gl < L
        pre_ln = rb(dhid) @ rb(P[pre + "linear1.weight"])
        pre_ln = (hrb(pre_ln) if enc else pre_ln) + dz
        want, dg, db = ln_bwd(pre_ln, ws(prev_xh, gl), ws(prev_rstd, gl, 1), P[pre + prev_norm + ".weight"])
        HB = dict(tol=2.0 ** -7, rms_tol=4e-4) if enc else dict(tol=1e-4)
        _close(tmp(dzprev_name, gl, d), want, pre + "linear1 dgrad + " + prev_norm + " backward", **HB)
        _close(G[pre + prev_norm + ".weight"], dg, "grad " + pre + prev_norm + ".weight", **HB)
        _close(G[pre + prev_norm + ".bias"], db, "grad " + pre + prev_norm + ".bias", **HB)
        n += 4

    if Ld:
        # encoder half of the encoder-decoder backward: the memory gradient = the sum of the decoder layers' cross-attention k / v
        # dgrads; through the final encoder norm and the top layer's closing norm it becomes that layer's dz (one fused row pass on the device)
        dmem = np.zeros((M, d))
        for l in range(Ld):
            dkvx = f64(r.ws_get("dqkvx", L + l))[M * d:].reshape(M, 2 * d)
            dmem += rb(dkvx) @ rb(P["Decoder.Decoder.layers.%d.multihead_attn.in_proj_weight" % l][d:])
        _close(ws("dmem"), dmem, "memory gradient (sum of the cross-attention k / v dgrads)", tol=1e-4)
        g_top, dg, db = ln_bwd(ws("dmem"), ws("enc_xhat"), ws("enc_rstd", 0, 1), P["Encoder.Encoder.norm.weight"])
        _close(G["Encoder.Encoder.norm.weight"], dg, "grad Encoder.Encoder.norm.weight", tol=1e-4)
        _close(G["Encoder.Encoder.norm.bias"], db, "grad Encoder.Encoder.norm.bias", tol=1e-4)
        want, _, _ = ln_bwd(g_top, ws("xhat2", L - 1), ws("rstd2", L - 1, 1), P["Encoder.Encoder.layers.%d.norm2.weight" % (L - 1)])
        _close(tmp("dzA", L - 1, d), want, "final encoder norm + top layer's norm2 backward", tol=1e-4)
        n += 4
    for l in reversed(range(L)):
        pre = "Encoder.Encoder.layers.%d." % l
        ffn_bwd(pre, l, ws("x1", l), "dzA", "xhat1", "rstd1", "norm1", "dzB")
        dz1, dz1m = tmp("dzB", l, d), tmp("dzBm", l, d)
        if p > 0:
            _close(dz1m, dz1 * mask(ng.layer_site(l, ng.S_DROP1), M * d, (M, d)), pre + "dropout1 mask on the gradient", tol=1e-6,
                   stored16=r.bf16_only("dzBm", l))
        wgrad(dz1m, ws("ctx", l), pre + "self_attn.out_proj.weight", pre + "self_attn.out_proj.bias")
        dqkv = tmp("dqkv", l, 3 * d)
        if P2:      # attention backward over bf16-stored q / k / v / dctx (attn_bwd_lds_kernel, IN16): dctx = the out-proj dgrad, itself stored in bf16
            dctx = hrb(rb(dz1m) @ rb(P[pre + "self_attn.out_proj.weight"]))
            qkv = ws("qkv", l)
            q_, k_, v_ = (split_heads(qkv[:, i * d:(i + 1) * d]) for i in range(3))
            pr = f64(r.ws_get("P", l)).reshape(M // 32, heads, 32, 32)
            mk = mask(ng.layer_site(l, ng.S_ATTN), pr.size, pr.shape)
            do = split_heads(dctx)
            dp = (do @ v_.transpose(0, 1, 3, 2)) * mk
            ds = pr * (dp - (dp * pr).sum(-1, keepdims=True)) / np.sqrt(hd)
            want = np.concatenate([merge_heads(ds @ k_), merge_heads(ds.transpose(0, 1, 3, 2) @ q_), merge_heads((pr * mk).transpose(0, 1, 3, 2) @ do)], 1)
            _close(dqkv, want, pre + "attention backward over bf16-stored operands", tol=2.0 ** -6, rms_tol=2e-3)
            n += 1
        wgrad(dqkv, enc_in[l], pre + "self_attn.in_proj_weight", pre + "self_attn.in_proj_bias")
        dx = rb(dqkv) @ rb(P[pre + "self_attn.in_proj_weight"])
        dx = (hrb(dx) if (P2 and l > 0) else dx) + dz1
        if l > 0:
            pp = "Encoder.Encoder.layers.%d." % (l - 1)
            want, _, _ = ln_bwd(dx, ws("xhat2", l - 1), ws("rstd2", l - 1, 1), P[pp + "norm2.weight"])
            _close(tmp("dzA", l - 1, d), want, pre + "in_proj dgrad + previous layer's norm2 backward",
                   **(dict(tol=2.0 ** -7, rms_tol=4e-4) if P2 else dict(tol=1e-4)))
        else:
            da = dx * mask(ng.SITE_PE_ENC, M * d, (M, d)) * (ws("a0") > 0)
            _close(ws("dctx"), da, "InputLayer backward (in_proj dgrad, dropout, relu mask)", tol=1e-4)
            wgrad(ws("dctx"), f64(x).reshape(M, -1), "InputLayerEncoder.Linear.weight", "InputLayerEncoder.Linear.bias")
        n += 1
    for l in reversed(range(Ld)):
        pre, gl = "Decoder.Decoder.layers.%d." % l, L + l
        ffn_bwd(pre, gl, ws("x2", gl), "dzA", "xhatx", "rstdx", "norm2", "dzB")
        wgrad(tmp("dzBm", gl, d), ws("ctxx", gl), pre + "multihead_attn.out_proj.weight", pre + "multihead_attn.out_proj.bias")
        dqkvx = f64(r.ws_get("dqkvx", gl))
        dqx, dkvx = dqkvx[:M * d].reshape(M, d), dqkvx[M * d:].reshape(M, 2 * d)
        gw, gb = G[pre + "multihead_attn.in_proj_weight"], G[pre + "multihead_attn.in_proj_bias"]
        _close(gw[:d], rb(dqx).T @ rb(ws("x1", gl)), "grad " + pre + "cross q in_proj weight")
        _close(gw[d:], rb(dkvx).T @ rb(ws("memory")), "grad " + pre + "cross kv in_proj weight")
        _close(gb, np.concatenate([rb(dqx).sum(0), rb(dkvx).sum(0)]), "grad " + pre + "cross in_proj bias")
        wgrad(tmp("dzCm", gl, d), ws("ctx", gl), pre + "self_attn.out_proj.weight", pre + "self_attn.out_proj.bias")
        wgrad(tmp("dqkv", gl, 3 * d), dec_in[l], pre + "self_attn.in_proj_weight", pre + "self_attn.in_proj_bias")
        n += 3
    return n


def check_bf16_shadows(backend, cfg, B, p):
    """precision = 1 where the Linears run on the big-tile kernel (csrc/groove_hip.hip bf16_shadows): the producers of every GEMM operand
    also write a bf16 copy and the GEMMs stage those (gemm32h_kernel) -- each shadow must be, bit for bit, the RNE rounding of the fp32
    tensor beside it (the value the fp32-source kernel rounds at fragment assembly), for activations, gradients and both weight copies.
    With that the shadow path's results are those of the fp32-source path; check_step_bf16 on the same shape compares them with the oracle."""
    cfg = dict(cfg, dropout=p, precision=1)
    P = ng.init_params(cfg, seed=3, perturb=0.05)
    x, y = ng.synthetic_batch(B, cfg["embedding_size_src"], seed=5)
    lib = harness.emu_lib() if backend == "emu" else harness._lib.get_lib()
    lib.cdll.gt_set_operand_shadows(1)               # (off by default; process-wide switch: restored below)
    try:
        return _check_bf16_shadows(backend, cfg, B, p, P, x, y)
    finally:
        lib.cdll.gt_set_operand_shadows(-1)


def _check_bf16_shadows(backend, cfg, B, p, P, x, y):
    r = Runner(cfg, B, backend, rng=(1234, 99, 7))
    r.set_params(P)
    r.forward(x, train=p > 0)
    _, dpred = r.loss(y, 0.47)
    r.backward(dpred, train=p > 0)
    L, d, F = cfg["num_encoder_layers"], cfg["d_model"], cfg["dim_feedforward"]

    def bf(a):
        return (ng.round_bf16(np.asarray(a, np.float32)).view(np.uint32) >> 16).astype(np.uint16)

    def shadow(name, l):
        o, c = r.lib.ws_find(r.c, name + "16", l)
        return r.ws.numpy()[o:o + c].view(np.uint16)

    n = 0
    want = bf(r.ws_get("x0", 0))                          # the input layer's output (operand of layer 0's in-proj and of its weight gradient)
    got = shadow("x0", 0)[:want.size]
    assert np.array_equal(got, want), ("x0", int((got != want).sum()), want.size)
    n += 1
    for l in range(L):
        for name in ["ctx", "x1", "hact"] + (["xout"] if l + 1 < L else []) + ["dhid", "dqkv", "dzAm" if p > 0 else "dzA", "dzBm" if p > 0 else "dzB"]:
            want = bf(r.ws_get(name, l))
            got = shadow(name, l)[:want.size]
            assert np.array_equal(got, want), (name, l, int((got != want).sum()), want.size)
            n += 1
        pre = "Encoder.Encoder.layers.%d." % l
        o, c = r.lib.ws_find(r.c, "w16", l)
        w16 = r.ws.numpy()[o:o + c].view(np.uint16)
        o, c = r.lib.ws_find(r.c, "w16t", l)
        w16t = r.ws.numpy()[o:o + c].view(np.uint16)
        at = 0
        for wn in ("self_attn.in_proj_weight", "self_attn.out_proj.weight", "linear1.weight", "linear2.weight"):
            W = np.asarray(P[pre + wn], np.float32)
            assert np.array_equal(w16[at:at + W.size], bf(W).reshape(-1)), (wn, l)
            assert np.array_equal(w16t[at:at + W.size], bf(W.T.copy()).reshape(-1)), (wn, l, "transposed")
            at += W.size
            n += 2
    return n


def check_step_bf16(backend, cfg, B, p=0.0, penalty=0.47, seed=3, precision=1):
    """precision = 2: the same two bars; the end-to-end sanity bound is taken against the bf16-OPERAND oracle with a wider factor (the
    storage roundings of precision 2 are additional noise of the same size: 2.5 x the bf16 effect)"""
    cfg = dict(cfg, dropout=p, precision=precision)
    cfg0 = dict(cfg, precision=0)
    P = ng.init_params(cfg, seed=seed, perturb=0.05)
    x, y = ng.synthetic_batch(B, cfg["embedding_size_src"], seed=seed + 2)
    Ld = cfg.get("num_decoder_layers", 0)
    tgt = shift_right(y) if Ld else None
    rng = (1234, 99, 7)
    r = Runner(cfg, B, backend, rng=rng)
    r.set_params(P)
    hvo = r.forward(x, tgt, train=p > 0)
    # (2) end to end, sanity bound
    (h, v, o), C = ng.forward(P, cfg, x, tgt=tgt, rng=rng if p > 0 else None, dtype=np.float64)
    (h0, v0, o0), _ = ng.forward(P, cfg0, x, tgt=tgt, rng=rng if p > 0 else None, dtype=np.float64)
    ref, ref0 = np.concatenate([h, v, o], -1), np.concatenate([h0, v0, o0], -1)
    eq = _rms(ref0 - ref)
    assert eq > 1e-5, "the bf16 rounding has no visible effect on this case: pick another"
    assert np.abs(hvo - ref).max() < BF16_OUT_MAX, "forward max-abs %g" % np.abs(hvo - ref).max()
    frac = BF16_E2E_FRAC if r.precision_in_force() < 2 else 2.5
    assert _rms(hvo - ref) <= frac * eq + 1e-5, "forward rms %g vs bf16 effect %g" % (_rms(hvo - ref), eq)
    stats, d_hvo = r.loss(y, penalty)
    rstats, _ = ng.calculate_loss((h, v, o), y.astype(np.float64), penalty)
    assert abs(stats[0] - rstats[0]) < 1e-2 * max(1.0, abs(rstats[0]))
    G = r.backward(train=p > 0)
    # (1) per operation, teacher-forced: the parity proof
    nops = check_ops_bf16(r, P, cfg, x, tgt, rng, p, G)
    assert nops >= 10 * (cfg["num_encoder_layers"] + Ld)
    return r, P, G


def check_autocast_anchor(backend, cfg, B, seed=3):
    """precision = 2 against the third-party definition it is anchored on: the eval forward of the stock torch modules under
    torch.autocast(bfloat16) (oracle.torch_groove.forward_autocast).  Two realisations of "bf16 where the bytes are" cannot agree element
    for element (autocast also runs the attention products in bf16 and rounds more tensors); the device must sit within the bf16 effect
    itself: RMS(device - autocast) and RMS(device - fp32) <= 1.5 x RMS(fp32 - autocast)."""
    from oracle import torch_groove as tg
    cfg = dict(cfg, dropout=0.0, precision=2)
    P = ng.init_params(cfg, seed=seed, perturb=0.05)
    x, y = ng.synthetic_batch(B, cfg["embedding_size_src"], seed=seed + 2)
    r = Runner(cfg, B, backend)
    assert r.precision_in_force() == 2, "precision 2 is not in force for this shape"
    r.set_params(P)
    hvo = r.forward(x, None, train=False)
    ac, ref = tg.forward_autocast(P, cfg, x)
    eq = _rms(ref - ac)
    assert eq > 1e-5
    assert _rms(hvo - ac) <= 1.5 * eq + 1e-5 and _rms(hvo - ref) <= 1.5 * eq + 1e-5, (_rms(hvo - ac), _rms(hvo - ref), eq)
    assert np.abs(hvo - ac).max() < BF16_OUT_MAX
    return _rms(hvo - ac), _rms(hvo - ref), eq


def check_train_step_bf16(backend, cfg, B, p, precision=1):
    """gt_train_step with precision = 1 (fp32 master weights): after each of two SGD steps the per-operation checks hold on the
    step's own saved state, the update is exactly  w -= lr * g  of the device's gradients ... observed through the parameters:
    they move along the bf16-operand oracle's gradient within the end-to-end bound."""
    cfg = dict(cfg, dropout=p, precision=precision)
    P = ng.init_params(cfg, seed=9, perturb=0.05)
    x, y = ng.synthetic_batch(B, cfg["embedding_size_src"], seed=4)
    tgt = shift_right(y) if cfg.get("num_decoder_layers", 0) else None
    r = Runner(cfg, B, backend, rng=(77, 5, 0), lr=0.05)
    r.set_params(P)
    cur = {k: v.astype(np.float64) for k, v in P.items()}
    for step in range(2):
        before = r.unflatten(r.params.numpy())
        stats = r.train_step(x, y, 0.38, algo=0)
        (h, v, o), C = ng.forward(before, cfg, x, tgt=tgt, rng=(77, 5, step) if p > 0 else None, dtype=np.float64)
        rstats, dpred = ng.calculate_loss((h, v, o), y.astype(np.float64), 0.38)
        assert abs(stats[0] - rstats[0]) < 1e-2 * max(1, abs(rstats[0])), (step, stats[0], rstats[0])
        check_ops_bf16(r, before, cfg, x, tgt, (77, 5, step), p)                    # forward state of THIS step
        G = ng.backward(before, cfg, C, dpred, dtype=np.float64)
        got = r.unflatten(r.params.numpy())
        for k in G:
            upd = (before[k].astype(np.float64) - got[k]) / 0.05                  # the gradient the device applied
            assert np.abs(upd - G[k]).max() <= 5e-2 * max(np.abs(G[k]).max(), 1e-4) + 1e-5, (step, k)
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


def expected_packs(Pd, cfg):
    """Fragment-ordered weight copies of the sequence-resident kernels (csrc/gt_seq.h, seq_pack_kernel) from a name -> array dict:
    per layer [in_w | out_w | w1 | w2]; pack[((t * nkt + u) * 64 + lane) * 4 + j] = B(16 u + 4 lg + j, 16 t + l16), lane = l16 + 16 lg;
    forward copy: B(k, n) = W[n][k], dgrad copy: B(k, n) = W[k][n]."""
    pf, pb = [], []
    for l in range(cfg["num_encoder_layers"]):
        pre = "Encoder.Encoder.layers.%d." % l
        for name in ("self_attn.in_proj_weight", "self_attn.out_proj.weight", "linear1.weight", "linear2.weight"):
            W = np.asarray(Pd[pre + name], np.float32)
            R, C = W.shape
            pf.append(W.reshape(R // 16, 16, C // 16, 4, 4).transpose(0, 2, 3, 1, 4).reshape(-1))      # (t, l16, u, lg, j) -> (t, u, lg, l16, j)
            pb.append(W.reshape(R // 16, 4, 4, C // 16, 16).transpose(3, 0, 1, 4, 2).reshape(-1))      # (u, lg, j, t, l16) -> (t, u, lg, l16, j)
    return np.concatenate(pf), np.concatenate(pb)


def check_train_step(backend, cfg, B, p, algo=0, seq=True):
    """gt_train_step == forward+loss+backward+update with the oracle's masks; second step uses step+1."""
    cfg = dict(cfg, dropout=p)
    P = ng.init_params(cfg, seed=9, perturb=0.05)
    x, y = ng.synthetic_batch(B, cfg["embedding_size_src"], seed=4)
    Ld = cfg.get("num_decoder_layers", 0)
    tgt = shift_right(y) if Ld else None
    r = Runner(cfg, B, backend, rng=(77, 5, 0), lr=0.05, seq=seq)
    r.set_params(P)
    cur = {k: v.astype(np.float64) for k, v in P.items()}
    folded = r.lib.cdll.gt_step_launches(ctypes.byref(r.c)) > 0     # sequence-resident path: the update writes the next step's weight packs
    for step in range(3 if folded else 2):
        # from the second step on: GT_STEP_PACKS_CURRENT (4) -- the previous step's update left this step's fragment-ordered weights
        stats = r.train_step(x, y, 0.38, algo=algo, skip_update=4 if (folded and step > 0) else 0)
        (h, v, o), C = ng.forward(cur, cfg, x, tgt=tgt, rng=(77, 5, step) if p > 0 else None, dtype=np.float64)
        rstats, dpred = ng.calculate_loss((h, v, o), y.astype(np.float64), 0.38)
        assert abs(stats[0] - rstats[0]) < 2e-5 * max(1, abs(rstats[0])), (step, stats[0], rstats[0])
        G = ng.backward(cur, cfg, C, dpred, dtype=np.float64)
        if algo == 0:
            cur = {k: cur[k] - 0.05 * G[k] for k in cur}
        else:
            if step == 0:
                am, av = {k: np.zeros_like(v) for k, v in cur.items()}, {k: np.zeros_like(v) for k, v in cur.items()}
            cur, am, av = ng.adam_step(cur, G, am, av, step + 1, 0.05)
        got = r.unflatten(r.params.numpy())
        for k in cur:
            # (adam: an element whose gradient is numerically zero moves by lr * g / (|g| + eps) -- its sign is noise; skip those)
            live = np.abs(G[k]) > 1e-6 if algo == 1 else np.ones(G[k].shape, bool)
            assert np.abs(got[k] - cur[k])[live].max(initial=0) < (1e-3 if algo == 1 else 2e-5) * max(1.0, np.abs(cur[k]).max()), (step, k)
            if algo == 1:
                # Adam divides by sqrt(v): an element's error is the gradient's error (bar 2e-4 of the tensor's largest entry) over |g|.
                # Where the gradient is well conditioned -- within a factor 10 of the largest -- the update keeps a tight bar
                strong = np.abs(G[k]) > 0.1 * np.abs(G[k]).max()
                assert np.abs(got[k] - cur[k])[strong].max(initial=0) < 1e-4 * max(1.0, np.abs(cur[k]).max()), (step, k, "well-conditioned elements")
        if folded and cfg["d_model"] % 16 == 0:
            # the folded update + pack kernel wrote the NEXT step's fragment-ordered weights: bitwise what packing the updated parameters gives
            ef, eb = expected_packs(got, cfg)
            for name, want in (("pack_f", ef), ("pack_b", eb)):
                o, c = r.lib.ws_find(r.c, name)
                assert c == want.size, (name, c, want.size)
                assert np.array_equal(r.ws.numpy()[o:o + c].view(np.uint32), want.view(np.uint32)), (step, name)
    assert r.step_state().step == (3 if folded else 2)
    try:                                             # QUAD forward: no pair exchange timed out (error word of the region's header)
        o, _ = r.lib.ws_find(r.c, "seq_xchg")
        assert r.ws.numpy()[o:o + 1].view(np.uint32)[0] == 0
    except Exception as e:
        assert "seq_xchg" in str(e) or "absent" in str(e) or "unknown" in str(e), e


def check_predict(backend, cfg, B, use_thres=True, thres=0.5, out_tol=OUT_TOL, margin_tol=1e-4, pd_seed=None):
    """out_tol / margin_tol: the fp32 bars by default; the bf16 operand path passes its own (hits compared where the decision
    margin exceeds what a bf16 rounding flip can move a probability by)."""
    cfg = dict(cfg, dropout=0.3)      # predict is eval mode: dropout must be ignored
    P = ng.init_params(cfg, seed=21, perturb=0.05)
    x, _ = ng.synthetic_batch(B, cfg["embedding_size_src"], seed=8)
    r = Runner(cfg, B, backend)
    r.set_params(P)
    hvo = r.predict(x, thres=thres, use_thres=use_thres, pd_seed=pd_seed)         # pd_seed: the reference's use_pd (sampled hits)
    (h, v, o), margin = ng.predict(P, cfg, x, use_thres=use_thres, thres=thres, dtype=np.float64, pd_seed=pd_seed)
    if pd_seed is not None:
        frac = float(hvo[..., :9].mean())
        assert 0.02 < frac < 0.98, frac                                            # really sampled: neither all hits nor none
    if use_thres or pd_seed is not None:
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


def check_golden_bf16(backend, path):
    """The committed stock-torch vectors (full-depth C4 / C5 models) on the bf16-operand path (precision = 1): the per-operation
    teacher-forced check -- the parity proof of this mode -- on the golden's own parameters and inputs, and the torch fp32 outputs /
    loss as the end-to-end sanity bound (what separates the two is the bf16 rounding of the operands, nothing else)."""
    z = np.load(path)
    cfg = {k: (float(v) if k == "dropout" else int(v)) for k, v in zip(z["cfg_keys"], z["cfg_vals"])}
    cfg = dict(cfg, precision=1)
    P = ng.init_params(cfg, seed=int(z["seed"]), perturb=0.05)
    x, y = z["x"], z["y"]
    tgt = shift_right(y) if cfg["num_decoder_layers"] else None
    r = Runner(cfg, x.shape[0], backend, lr=0.094)
    r.set_params(P)
    hvo = r.forward(x, tgt)
    assert np.abs(hvo - np.concatenate([z["h"], z["v"], z["o"]], -1)).max() < BF16_OUT_MAX
    stats, _ = r.loss(y, 0.47)
    assert abs(stats[0] - z["stats_pen0.47"][0]) < 1e-2 * max(1.0, abs(z["stats_pen0.47"][0]))
    G = r.backward()
    nops = check_ops_bf16(r, P, cfg, x, tgt, (1234, 99, 0), 0.0, G)
    assert nops >= 10 * (cfg["num_encoder_layers"] + cfg["num_decoder_layers"])


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


def check_bucketed_backward(backend, cfg, B, p, n_buckets, exact, seq=True):
    """gt_train_step(skip_update=2) must leave bucket 0 of gt_grad_buckets FINAL (that is what the data-parallel path
    all-reduces while skip_update=3 runs), and 2 followed by 3 must equal the one-call backward (skip_update=1)."""
    cfg = dict(cfg, dropout=p)
    P = ng.init_params(cfg, seed=11, perturb=0.05)
    x, y = ng.synthetic_batch(B, cfg["embedding_size_src"], seed=12)
    whole = Runner(cfg, B, backend, rng=(5, 6, 0), seq=seq)
    whole.set_params(P)
    whole.train_step(x, y, 0.47, skip_update=1)
    gw = whole.grads.numpy().copy()
    r = Runner(cfg, B, backend, rng=(5, 6, 0), seq=seq)
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
