"""Explicit numpy restatement of the GrooveTransformer hot path, with manual backward.

TEST INFRASTRUCTURE ONLY -- never imported by ``transformergrooveinfilling_amd``.

Every array is (M, features) with M = B*32 rows, row m = b*32 + t (batch-first, which is what
the dataset hands over: ref:dataset.py:355-356, ref:train.py:156-158).  Parameter names are the
demo checkpoint's state-dict keys (ref:demo/transformer_run_171tyqit_Epoch_1.Model).

Third-party algorithm restated (PyTorch, pinned 1.10.2 at ref:environment.yaml:61):
  Linear                    y = x W^T + b
  MultiheadAttention        torch:nn/functional.py:5820-5850 (packed q,k,v in-projection; cross-
                            attention: q from tgt with W[0:d], k,v from memory with W[d:3d]),
                            :6504-6642 (scale 1/sqrt(hd), additive mask, softmax, dropout, P.V,
                            out-projection)
  TransformerEncoderLayer   post-norm, torch:nn/modules/transformer.py:951-956,961-982
  TransformerDecoderLayer   post-norm, torch:nn/modules/transformer.py:1143-1153
  LayerNorm                 eps=1e-5, biased variance
  final encoder/decoder LN  torch:nn/modules/transformer.py:550-551,655-656
  BCEWithLogitsLoss         max(x,0) - x*y + log1p(exp(-|x|))        (ref:train.py:176-179)
  MSELoss                   (x-y)^2
  SGD / Adam                torch.optim defaults (ckpt: lr, momentum 0; Adam b=(0.9,0.999) eps 1e-8)
Glue restated from the un-vendored submodule as recalled (PARITY UNPINNED, see oracle/__init__):
  InputLayer  = Linear -> ReLU -> +pe[t] -> dropout;  OutputLayer = Linear -> [h | sigmoid | 0.5 tanh]
  calculate_loss, predict (threshold / greedy decode).

Dropout uses the SAME counter-based hash as the HIP kernels (include/groove_hip.h,
"dropout RNG"), so train-mode parity with p>0 is exact in which elements are dropped.
"""
import math

import numpy as np

T = 32
NV = 9  # voices; HVO = [h(9) | v(9) | o(9)] (ref:utils.py:38-47)

# ---- dropout sites (must match include/groove_hip.h) ---------------------------------------
SITE_PE_ENC, SITE_PE_DEC, SITE_LAYER0, SITE_STRIDE = 0, 1, 16, 8
S_ATTN, S_DROP1, S_FFN, S_DROPF, S_XATTN, S_DROP2 = 0, 1, 2, 3, 4, 5


def layer_site(global_layer, kind):
    return SITE_LAYER0 + global_layer * SITE_STRIDE + kind


def _fmix32(h):
    h = h.astype(np.uint64)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    return h


def site_key(seed_lo, seed_hi, step, site):
    m = np.uint64(0xFFFFFFFF)
    s = (np.uint64(seed_lo) ^ _fmix32(np.array(step, dtype=np.uint64))) & m
    k = (s ^ ((np.uint64(site) * np.uint64(0x9E3779B9)) & m)) & m
    k = _fmix32(k) ^ np.uint64(seed_hi)
    k = _fmix32((k + np.uint64(0x7F4A7C15)) & m)
    return k


def keep_mask(rng, site, n, p):
    """rng = (seed_lo, seed_hi, step).  Returns float mask in {0, 1/(1-p)} for idx 0..n-1."""
    if rng is None or p <= 0.0:
        return None
    key = site_key(rng[0], rng[1], rng[2], site)
    idx = np.arange(n, dtype=np.uint64)
    r = _fmix32(((idx * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)) ^ key)
    thr = np.uint64(int(np.float32(p) * np.float32(16777216.0)))
    keep = (r >> np.uint64(8)) >= thr
    return keep.astype(np.float64) * (1.0 / (1.0 - float(np.float32(p))))


# ---- parameter bookkeeping -------------------------------------------------------------------
def param_names(cfg):
    """Trainable tensors in state-dict order (pe buffer excluded)."""
    d, F, S, L, Ld = cfg["d_model"], cfg["dim_feedforward"], cfg["embedding_size_src"], \
        cfg["num_encoder_layers"], cfg.get("num_decoder_layers", 0)
    out = [("InputLayerEncoder.Linear.weight", (d, S)), ("InputLayerEncoder.Linear.bias", (d,))]
    for l in range(L):
        p = "Encoder.Encoder.layers.%d." % l
        out += [(p + "self_attn.in_proj_weight", (3 * d, d)), (p + "self_attn.in_proj_bias", (3 * d,)),
                (p + "self_attn.out_proj.weight", (d, d)), (p + "self_attn.out_proj.bias", (d,)),
                (p + "linear1.weight", (F, d)), (p + "linear1.bias", (F,)),
                (p + "linear2.weight", (d, F)), (p + "linear2.bias", (d,)),
                (p + "norm1.weight", (d,)), (p + "norm1.bias", (d,)),
                (p + "norm2.weight", (d,)), (p + "norm2.bias", (d,))]
    out += [("Encoder.Encoder.norm.weight", (d,)), ("Encoder.Encoder.norm.bias", (d,))]
    if Ld:
        out += [("InputLayerDecoder.Linear.weight", (d, 27)), ("InputLayerDecoder.Linear.bias", (d,))]
        for l in range(Ld):
            p = "Decoder.Decoder.layers.%d." % l
            out += [(p + "self_attn.in_proj_weight", (3 * d, d)), (p + "self_attn.in_proj_bias", (3 * d,)),
                    (p + "self_attn.out_proj.weight", (d, d)), (p + "self_attn.out_proj.bias", (d,)),
                    (p + "multihead_attn.in_proj_weight", (3 * d, d)), (p + "multihead_attn.in_proj_bias", (3 * d,)),
                    (p + "multihead_attn.out_proj.weight", (d, d)), (p + "multihead_attn.out_proj.bias", (d,)),
                    (p + "linear1.weight", (F, d)), (p + "linear1.bias", (F,)),
                    (p + "linear2.weight", (d, F)), (p + "linear2.bias", (d,)),
                    (p + "norm1.weight", (d,)), (p + "norm1.bias", (d,)),
                    (p + "norm2.weight", (d,)), (p + "norm2.bias", (d,)),
                    (p + "norm3.weight", (d,)), (p + "norm3.bias", (d,))]
        out += [("Decoder.Decoder.norm.weight", (d,)), ("Decoder.Decoder.norm.bias", (d,))]
    out += [("OutputLayer.Linear.weight", (27, d)), ("OutputLayer.Linear.bias", (27,))]
    return out


def positional_encoding(d, max_len=T, dtype=np.float32):
    """fp32 arithmetic exactly as the torch buffer is built (checked against ckpt pe to 6e-8)."""
    pe = np.zeros((max_len, d), np.float32)
    pos = np.arange(max_len, dtype=np.float32)[:, None]
    div = np.exp(np.arange(0, d, 2).astype(np.float32) * np.float32(-math.log(10000.0) / d)).astype(np.float32)
    pe[:, 0::2] = np.sin(pos * div)
    pe[:, 1::2] = np.cos(pos * div)
    return pe.astype(dtype)


# ---- bf16 operand mode (gt_config.precision = 1; BASELINE configs[4]) ---------------------------------------------------
# Every Linear's forward / dgrad / wgrad matmul takes its two operands rounded to bf16 (round to nearest even, from their
# fp32 values -- the device converts fp32 tensors on the way into the matrix cores) and accumulates exactly (fp32 on the
# device, the oracle's dtype here); bias gradients are column sums of the ROUNDED output gradient (the device sums the
# staged bf16 slab).  Attention core, LayerNorm, softmax, loss, optimizer: unchanged.  cfg["precision"] switches it on.
_BF16 = [False]


def round_bf16(a):
    a32 = np.ascontiguousarray(a, dtype=np.float32)
    u = a32.view(np.uint32).astype(np.uint64)
    r = ((u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) >> np.uint64(16)) << np.uint64(16)
    return r.astype(np.uint32).view(np.float32).astype(np.asarray(a).dtype).reshape(np.shape(a))


def _mm(a, b):
    """a @ b of a Linear (forward, dgrad or wgrad)."""
    if _BF16[0]:
        return round_bf16(a) @ round_bf16(b)
    return a @ b


def _cs(dy):
    """bias gradient: column sums of the output gradient."""
    return (round_bf16(dy) if _BF16[0] else dy).sum(0)


class _precision:
    def __init__(self, cfg):
        self.on = bool(cfg.get("precision", 0))

    def __enter__(self):
        self.prev, _BF16[0] = _BF16[0], self.on

    def __exit__(self, *a):
        _BF16[0] = self.prev


# ---- primitive ops ---------------------------------------------------------------------------
def _ln_fwd(z, g, b, eps=1e-5):
    mu = z.mean(-1, keepdims=True)
    var = ((z - mu) ** 2).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (z - mu) * rstd
    return xhat * g + b, xhat, rstd


def _ln_bwd(dy, xhat, rstd, g):
    gdy = dy * g
    m1 = gdy.mean(-1, keepdims=True)
    m2 = (gdy * xhat).mean(-1, keepdims=True)
    return rstd * (gdy - m1 - xhat * m2), (dy * xhat).sum(0), dy.sum(0)


def _drop(a, rng, site, p):
    m = keep_mask(rng, site, a.size, p)
    if m is None:
        return a, None
    m = m.reshape(a.shape).astype(a.dtype)
    return a * m, m


def _attn_fwd(q_in, kv_in, Win, bin_, Wo, bo, H, causal, rng, site, p):
    """q_in (M,d) supplies queries, kv_in (M,d) keys/values.  Returns out (M,d) and cache."""
    M, d = q_in.shape
    B, hd = M // T, d // H
    q = _mm(q_in, Win[:d].T) + bin_[:d]
    k = _mm(kv_in, Win[d:2 * d].T) + bin_[d:2 * d]
    v = _mm(kv_in, Win[2 * d:].T) + bin_[2 * d:]
    qh = q.reshape(B, T, H, hd).transpose(0, 2, 1, 3)
    kh = k.reshape(B, T, H, hd).transpose(0, 2, 1, 3)
    vh = v.reshape(B, T, H, hd).transpose(0, 2, 1, 3)
    S = (qh @ kh.transpose(0, 1, 3, 2)) * q_in.dtype.type(1.0 / math.sqrt(hd))
    if causal:
        S = S + np.triu(np.full((T, T), -np.inf, dtype=S.dtype), 1)
    S = S - S.max(-1, keepdims=True)
    E = np.exp(S)
    P = E / E.sum(-1, keepdims=True)                      # (B,H,T,T); idx = ((b*H+h)*T+i)*T+j
    Pd, mask = _drop(P, rng, site, p)
    ctx = (Pd @ vh).transpose(0, 2, 1, 3).reshape(M, d)
    out = _mm(ctx, Wo.T) + bo
    return out, dict(q_in=q_in, kv_in=kv_in, qh=qh, kh=kh, vh=vh, P=P, Pd=Pd, mask=mask, ctx=ctx,
                     q=q, k=k, v=v)


def _attn_bwd(dout, c, Win, Wo, H):
    M, d = dout.shape
    B, hd = M // T, d // H
    scale = dout.dtype.type(1.0 / math.sqrt(hd))
    g = {}
    g["out_w"] = _mm(dout.T, c["ctx"])
    g["out_b"] = _cs(dout)
    dctx = _mm(dout, Wo)
    dch = dctx.reshape(B, T, H, hd).transpose(0, 2, 1, 3)
    dPd = dch @ c["vh"].transpose(0, 1, 3, 2)
    dvh = c["Pd"].transpose(0, 1, 3, 2) @ dch
    dP = dPd * c["mask"] if c["mask"] is not None else dPd
    P = c["P"]
    dS = P * (dP - (dP * P).sum(-1, keepdims=True)) * scale
    dqh = dS @ c["kh"]
    dkh = dS.transpose(0, 1, 3, 2) @ c["qh"]
    dq = dqh.transpose(0, 2, 1, 3).reshape(M, d)
    dk = dkh.transpose(0, 2, 1, 3).reshape(M, d)
    dv = dvh.transpose(0, 2, 1, 3).reshape(M, d)
    dW = np.concatenate([_mm(dq.T, c["q_in"]), _mm(dk.T, c["kv_in"]), _mm(dv.T, c["kv_in"])], 0)
    db = np.concatenate([_cs(dq), _cs(dk), _cs(dv)])
    g["in_w"], g["in_b"] = dW, db
    d_q_in = _mm(dq, Win[:d])
    d_kv_in = _mm(dk, Win[d:2 * d]) + _mm(dv, Win[2 * d:])
    return d_q_in, d_kv_in, g, dict(dq=dq, dk=dk, dv=dv, dctx=dctx)


def _input_fwd(x, W, b, pe, rng, site, p):
    a = _mm(x, W.T) + b
    r = np.maximum(a, 0)
    e = r + np.tile(pe, (x.shape[0] // T, 1))
    out, mask = _drop(e, rng, site, p)
    return out, dict(x=x, a=a, mask=mask)


def _input_bwd(dout, c):
    de = dout * c["mask"] if c["mask"] is not None else dout
    da = de * (c["a"] > 0)
    return _mm(da.T, c["x"]), _cs(da)


# ---- model ------------------------------------------------------------------------------------
def forward(P, cfg, x, tgt=None, rng=None, dtype=np.float32):
    with _precision(cfg):
        return _forward(P, cfg, x, tgt, rng, dtype)


def _forward(P, cfg, x, tgt=None, rng=None, dtype=np.float32):
    """P: dict name->array.  x (B,T,S); tgt (B,T,27) for the encoder-decoder.  rng=(lo,hi,step)
    enables dropout with p=cfg['dropout'].  Returns (h,v,o) each (B,T,9) and the cache."""
    P = {k: np.asarray(v, dtype) for k, v in P.items()}
    d, H = cfg["d_model"], cfg["n_heads"]
    L, Ld = cfg["num_encoder_layers"], cfg.get("num_decoder_layers", 0)
    p = float(cfg.get("dropout", 0.0)) if rng is not None else 0.0
    B = x.shape[0]
    M = B * T
    pe = positional_encoding(d, T, dtype)
    C = dict(enc=[], dec=[])
    h0, C["in_enc"] = _input_fwd(np.asarray(x, dtype).reshape(M, -1), P["InputLayerEncoder.Linear.weight"],
                                 P["InputLayerEncoder.Linear.bias"], pe, rng, SITE_PE_ENC, p)
    xcur = h0
    for l in range(L):
        n = "Encoder.Encoder.layers.%d." % l
        c = {"x_in": xcur}
        ao, c["attn"] = _attn_fwd(xcur, xcur, P[n + "self_attn.in_proj_weight"], P[n + "self_attn.in_proj_bias"],
                                  P[n + "self_attn.out_proj.weight"], P[n + "self_attn.out_proj.bias"],
                                  H, False, rng, layer_site(l, S_ATTN), p)
        ao, c["m1"] = _drop(ao, rng, layer_site(l, S_DROP1), p)
        x1, c["xhat1"], c["rstd1"] = _ln_fwd(xcur + ao, P[n + "norm1.weight"], P[n + "norm1.bias"])
        c["x1"] = x1
        c["hpre"] = _mm(x1, P[n + "linear1.weight"].T) + P[n + "linear1.bias"]
        hact, c["mf"] = _drop(np.maximum(c["hpre"], 0), rng, layer_site(l, S_FFN), p)
        c["hact"] = hact
        f = _mm(hact, P[n + "linear2.weight"].T) + P[n + "linear2.bias"]
        f, c["m2"] = _drop(f, rng, layer_site(l, S_DROPF), p)
        xcur, c["xhat2"], c["rstd2"] = _ln_fwd(x1 + f, P[n + "norm2.weight"], P[n + "norm2.bias"])
        c["x_out"] = xcur
        C["enc"].append(c)
    mem, C["enc_xhat"], C["enc_rstd"] = _ln_fwd(xcur, P["Encoder.Encoder.norm.weight"], P["Encoder.Encoder.norm.bias"])
    C["memory"] = mem
    final = mem
    if Ld:
        ycur, C["in_dec"] = _input_fwd(np.asarray(tgt, dtype).reshape(M, -1), P["InputLayerDecoder.Linear.weight"],
                                       P["InputLayerDecoder.Linear.bias"], pe, rng, SITE_PE_DEC, p)
        for l in range(Ld):
            n = "Decoder.Decoder.layers.%d." % l
            gl = L + l
            c = {"x_in": ycur}
            sa, c["attn"] = _attn_fwd(ycur, ycur, P[n + "self_attn.in_proj_weight"], P[n + "self_attn.in_proj_bias"],
                                      P[n + "self_attn.out_proj.weight"], P[n + "self_attn.out_proj.bias"],
                                      H, True, rng, layer_site(gl, S_ATTN), p)
            sa, c["m1"] = _drop(sa, rng, layer_site(gl, S_DROP1), p)
            y1, c["xhat1"], c["rstd1"] = _ln_fwd(ycur + sa, P[n + "norm1.weight"], P[n + "norm1.bias"])
            c["x1"] = y1
            ca, c["xattn"] = _attn_fwd(y1, mem, P[n + "multihead_attn.in_proj_weight"], P[n + "multihead_attn.in_proj_bias"],
                                       P[n + "multihead_attn.out_proj.weight"], P[n + "multihead_attn.out_proj.bias"],
                                       H, False, rng, layer_site(gl, S_XATTN), p)
            ca, c["mx"] = _drop(ca, rng, layer_site(gl, S_DROP2), p)
            y2, c["xhatx"], c["rstdx"] = _ln_fwd(y1 + ca, P[n + "norm2.weight"], P[n + "norm2.bias"])
            c["x2"] = y2
            c["hpre"] = _mm(y2, P[n + "linear1.weight"].T) + P[n + "linear1.bias"]
            hact, c["mf"] = _drop(np.maximum(c["hpre"], 0), rng, layer_site(gl, S_FFN), p)
            c["hact"] = hact
            f = _mm(hact, P[n + "linear2.weight"].T) + P[n + "linear2.bias"]
            f, c["m2"] = _drop(f, rng, layer_site(gl, S_DROPF), p)
            ycur, c["xhat2"], c["rstd2"] = _ln_fwd(y2 + f, P[n + "norm3.weight"], P[n + "norm3.bias"])
            c["x_out"] = ycur
            C["dec"].append(c)
        final, C["dec_xhat"], C["dec_rstd"] = _ln_fwd(ycur, P["Decoder.Decoder.norm.weight"], P["Decoder.Decoder.norm.bias"])
    C["final"] = final
    logits = _mm(final, P["OutputLayer.Linear.weight"].T) + P["OutputLayer.Linear.bias"]
    C["logits"] = logits
    h = logits[:, :NV]
    v = 1.0 / (1.0 + np.exp(-logits[:, NV:2 * NV]))
    o = 0.5 * np.tanh(logits[:, 2 * NV:])
    C["v"], C["o"] = v, o
    rs = lambda a: a.reshape(B, T, NV)
    return (rs(h), rs(v), rs(o)), C


def calculate_loss(pred, y, penalty):
    """Returns (loss, hit_accuracy, hit_perplexity, bce, mse_v, mse_o), (dh, dv, do) w.r.t. (h,v,o)."""
    h, v, o = pred
    dt = h.dtype
    y = np.asarray(y, dt)
    y_h, y_v, y_o = y[..., :NV], y[..., NV:2 * NV], y[..., 2 * NV:]
    pen = np.where(y_h == 1, dt.type(1), dt.type(penalty))
    n = h.shape[0] * h.shape[1]
    bce_el = np.maximum(h, 0) - h * y_h + np.log1p(np.exp(-np.abs(h)))
    bce = (bce_el * pen).sum() / n
    mse_v = (((v - y_v) ** 2) * pen).sum() / n
    mse_o = (((o - y_o) ** 2) * pen).sum() / n
    hit = (h > 0).astype(dt)                                  # sigmoid(h) > 0.5  <=>  h > 0
    acc = (hit == y_h).mean()
    sig = 1.0 / (1.0 + np.exp(-h))
    dh = (sig - y_h) * pen / n
    dv = 2 * (v - y_v) * pen / n
    do = 2 * (o - y_o) * pen / n
    return (bce + mse_v + mse_o, acc, np.exp(bce), bce, mse_v, mse_o), (dh, dv, do)


def _ffn_bwd(dz, c, W1, W2, xin):
    """dz: grad of the pre-LN sum (residual + dropped ffn out).  Returns dx_in_from_ffn, grads."""
    df = dz * c["m2"] if c["m2"] is not None else dz
    g = {"w2": _mm(df.T, c["hact"]), "b2": _cs(df)}
    dh = _mm(df, W2)
    if c["mf"] is not None:
        dh = dh * c["mf"]
    dh = dh * (c["hpre"] > 0)
    g["w1"], g["b1"] = _mm(dh.T, xin), _cs(dh)
    return _mm(dh, W1), g, dict(dhid=dh)


def backward(P, cfg, C, dpred, dtype=np.float32):
    with _precision(cfg):
        return _backward(P, cfg, C, dpred, dtype)


def _backward(P, cfg, C, dpred, dtype=np.float32):
    """Manual backward of ``forward``.  dpred = (dh, dv, do).  Returns dict name->grad."""
    P = {k: np.asarray(v, dtype) for k, v in P.items()}
    H = cfg["n_heads"]
    L, Ld = cfg["num_encoder_layers"], cfg.get("num_decoder_layers", 0)
    G = {}
    dh, dv, do = [np.asarray(a, dtype).reshape(-1, NV) for a in dpred]
    v, o = C["v"], C["o"]
    dlog = np.concatenate([dh, dv * v * (1 - v), do * (0.5 - 2 * o * o)], 1)
    C["dlogits"] = dlog
    G["OutputLayer.Linear.weight"] = _mm(dlog.T, C["final"])
    G["OutputLayer.Linear.bias"] = _cs(dlog)
    dfin = _mm(dlog, P["OutputLayer.Linear.weight"])
    dmem = 0.0
    if Ld:
        dy, G["Decoder.Decoder.norm.weight"], G["Decoder.Decoder.norm.bias"] = \
            _ln_bwd(dfin, C["dec_xhat"], C["dec_rstd"], P["Decoder.Decoder.norm.weight"])
        for l in reversed(range(Ld)):
            n = "Decoder.Decoder.layers.%d." % l
            c = C["dec"][l]
            dz3, G[n + "norm3.weight"], G[n + "norm3.bias"] = _ln_bwd(dy, c["xhat2"], c["rstd2"], P[n + "norm3.weight"])
            dx2, g, _ = _ffn_bwd(dz3, c, P[n + "linear1.weight"], P[n + "linear2.weight"], c["x2"])
            G[n + "linear1.weight"], G[n + "linear1.bias"] = g["w1"], g["b1"]
            G[n + "linear2.weight"], G[n + "linear2.bias"] = g["w2"], g["b2"]
            dz2, G[n + "norm2.weight"], G[n + "norm2.bias"] = _ln_bwd(dx2 + dz3, c["xhatx"], c["rstdx"], P[n + "norm2.weight"])
            dca = dz2 * c["mx"] if c["mx"] is not None else dz2
            dq_in, dkv_in, g, _ = _attn_bwd(dca, c["xattn"], P[n + "multihead_attn.in_proj_weight"],
                                            P[n + "multihead_attn.out_proj.weight"], H)
            G[n + "multihead_attn.in_proj_weight"], G[n + "multihead_attn.in_proj_bias"] = g["in_w"], g["in_b"]
            G[n + "multihead_attn.out_proj.weight"], G[n + "multihead_attn.out_proj.bias"] = g["out_w"], g["out_b"]
            dmem = dmem + dkv_in
            dz1, G[n + "norm1.weight"], G[n + "norm1.bias"] = _ln_bwd(dq_in + dz2, c["xhat1"], c["rstd1"], P[n + "norm1.weight"])
            dsa = dz1 * c["m1"] if c["m1"] is not None else dz1
            dq_in, dkv_in, g, _ = _attn_bwd(dsa, c["attn"], P[n + "self_attn.in_proj_weight"],
                                            P[n + "self_attn.out_proj.weight"], H)
            G[n + "self_attn.in_proj_weight"], G[n + "self_attn.in_proj_bias"] = g["in_w"], g["in_b"]
            G[n + "self_attn.out_proj.weight"], G[n + "self_attn.out_proj.bias"] = g["out_w"], g["out_b"]
            dy = dq_in + dkv_in + dz1
        G["InputLayerDecoder.Linear.weight"], G["InputLayerDecoder.Linear.bias"] = _input_bwd(dy, C["in_dec"])
        dfin = dmem
    dx, G["Encoder.Encoder.norm.weight"], G["Encoder.Encoder.norm.bias"] = \
        _ln_bwd(dfin, C["enc_xhat"], C["enc_rstd"], P["Encoder.Encoder.norm.weight"])
    for l in reversed(range(L)):
        n = "Encoder.Encoder.layers.%d." % l
        c = C["enc"][l]
        dz2, G[n + "norm2.weight"], G[n + "norm2.bias"] = _ln_bwd(dx, c["xhat2"], c["rstd2"], P[n + "norm2.weight"])
        dx1, g, dbg = _ffn_bwd(dz2, c, P[n + "linear1.weight"], P[n + "linear2.weight"], c["x1"])
        G[n + "linear1.weight"], G[n + "linear1.bias"] = g["w1"], g["b1"]
        G[n + "linear2.weight"], G[n + "linear2.bias"] = g["w2"], g["b2"]
        dz1, G[n + "norm1.weight"], G[n + "norm1.bias"] = _ln_bwd(dx1 + dz2, c["xhat1"], c["rstd1"], P[n + "norm1.weight"])
        dao = dz1 * c["m1"] if c["m1"] is not None else dz1
        dq_in, dkv_in, g, dbg2 = _attn_bwd(dao, c["attn"], P[n + "self_attn.in_proj_weight"],
                                           P[n + "self_attn.out_proj.weight"], H)
        G[n + "self_attn.in_proj_weight"], G[n + "self_attn.in_proj_bias"] = g["in_w"], g["in_b"]
        G[n + "self_attn.out_proj.weight"], G[n + "self_attn.out_proj.bias"] = g["out_w"], g["out_b"]
        c["bwd"] = dict(dz2=dz2, dz1=dz1, dhid=dbg["dhid"], **dbg2)
        dx = dq_in + dkv_in + dz1
    G["InputLayerEncoder.Linear.weight"], G["InputLayerEncoder.Linear.bias"] = _input_bwd(dx, C["in_enc"])
    return G


def sgd_step(P, G, lr, grad_scale=1.0):
    return {k: (P[k] - np.float32(lr) * (G[k] * np.float32(grad_scale))).astype(P[k].dtype) if k in G else P[k]
            for k in P}


def adam_step(P, G, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (no weight decay, no amsgrad); step counts from 1."""
    out, m2, v2 = {}, {}, {}
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    for k in P:
        if k not in G:
            out[k] = P[k]
            continue
        g = G[k]
        m2[k] = b1 * m[k] + (1 - b1) * g
        v2[k] = b2 * v[k] + (1 - b2) * g * g
        denom = np.sqrt(v2[k]) / math.sqrt(bc2) + eps
        out[k] = (P[k] - (lr / bc1) * m2[k] / denom).astype(P[k].dtype)
    return out, m2, v2


def pd_uniforms(seed, B):
    """use_pd sampling: u[b, t, j] = (fmix32((idx * 0x9E3779B1) ^ seed) >> 8) * 2^-24, idx = (b * 32 + t) * 27 + j -- the counter hash of
    the dropout masks (include/groove_hip.h, gt_predict_pd)."""
    m = np.uint64(0xFFFFFFFF)
    idx = (np.arange(B * T, dtype=np.uint64)[:, None] * np.uint64(27) + np.arange(NV, dtype=np.uint64)[None, :]) & m
    h = _fmix32(((idx * np.uint64(0x9E3779B1)) & m) ^ np.uint64(seed & 0xFFFFFFFF))
    return ((h >> np.uint64(8)).astype(np.float64) / 16777216.0).reshape(B, T, NV)


def predict(P, cfg, x, use_thres=True, thres=0.5, dtype=np.float32, pd_seed=None):
    """eval-mode forward + threshold (ref:evaluator.py:173).  Encoder-decoder: greedy decode with
    tgt row 0 = zeros and row t+1 = the step-t prediction.  Returns (h,v,o) and the per-element
    decision margin |sigmoid(logit) - thres| (for "bit-exact where the margin allows" tests).
    pd_seed: the hits are sampled instead (use_pd): 1 iff p > pd_uniforms(pd_seed)."""
    Ld = cfg.get("num_decoder_layers", 0)
    B = x.shape[0]
    U = pd_uniforms(pd_seed, B) if pd_seed is not None else None

    def thr(hl, t=None):
        pr = 1.0 / (1.0 + np.exp(-hl.astype(np.float64)))
        cut = thres if U is None else (U if t is None else U[:, t])
        return (np.where(pr > cut, 1.0, 0.0) if (use_thres or U is not None) else pr).astype(dtype), np.abs(pr - cut)

    if not Ld:
        (h, v, o), _ = forward(P, cfg, x, dtype=dtype)
        hh, margin = thr(h)
        return (hh, v, o), margin
    tgt = np.zeros((B, T, 27), dtype)
    out = np.zeros((B, T, 27), dtype)
    margin = np.zeros((B, T, NV))
    for t in range(T):
        (h, v, o), _ = forward(P, cfg, x, tgt=tgt, dtype=dtype)
        hh, mg = thr(h[:, t], t)
        margin[:, t] = mg
        out[:, t] = np.concatenate([hh, v[:, t], o[:, t]], -1)
        if t + 1 < T:
            tgt[:, t + 1] = out[:, t]
    return (out[..., :NV], out[..., NV:2 * NV], out[..., 2 * NV:]), margin


def synthetic_batch(B, S, seed=1234, dtype=np.float32):
    """SURVEY 8(d) generator: x ~ U[0,1); hits ~ Bernoulli(0.15); v = U*h; o = (U-0.5)*h."""
    r = np.random.default_rng(seed)
    x = r.random((B, T, S), dtype=np.float32)
    h = (r.random((B, T, NV), dtype=np.float32) < 0.15).astype(np.float32)
    v = r.random((B, T, NV), dtype=np.float32) * h
    o = (r.random((B, T, NV), dtype=np.float32) - 0.5) * h
    return x.astype(dtype), np.concatenate([h, v, o], -1).astype(dtype)


def init_params(cfg, seed=0, perturb=0.0, dtype=np.float32):
    """Deterministic (numpy PCG64) torch-default-style init: xavier-uniform packed in-proj with zero
    attention biases, U(+-1/sqrt(fan_in)) linears, LayerNorm gamma=1 beta=0, IO layers U(+-0.1)
    with zero bias (ckpt: InputLayer weight range +-0.0995).  ``perturb`` adds N(0, perturb) to
    every bias / LayerNorm tensor so parity tests do not run on trivial zeros and ones."""
    r = np.random.default_rng(seed)
    P = {}
    for name, shape in param_names(cfg):
        if name.endswith("in_proj_weight"):
            bnd = math.sqrt(6.0 / (shape[0] + shape[1]))
            a = r.uniform(-bnd, bnd, shape)
        elif name.startswith(("InputLayer", "OutputLayer")):
            a = r.uniform(-0.1, 0.1, shape) if name.endswith("weight") else np.zeros(shape)
        elif "norm" in name:
            a = np.ones(shape) if name.endswith("weight") else np.zeros(shape)
        elif name.endswith(("in_proj_bias", "out_proj.bias")):
            a = np.zeros(shape)
        elif name.endswith("weight"):
            a = r.uniform(-1, 1, shape) / math.sqrt(shape[1])
        else:  # linear1/linear2 bias: fan_in of the matching weight
            fan_in = cfg["d_model"] if "linear1" in name else cfg["dim_feedforward"]
            a = r.uniform(-1, 1, shape) / math.sqrt(fan_in)
        if perturb and (len(shape) == 1):
            a = a + r.normal(0, perturb, shape)
        P[name] = a.astype(dtype)
    return P
