"""State-dict names of the GrooveTransformer parameters, in the order libgroove_hip.so lays them
out in its flat parameter buffer (gt_param_layout).  Names are the reference's checkpoint keys
(ref:demo/transformer_run_171tyqit_Epoch_1.Model); the decoder names follow torch's
nn.TransformerDecoderLayer (self_attn / multihead_attn / linear1,2 / norm1..3)."""
import math

import numpy as np

PE_KEYS = ("InputLayerEncoder.PositionalEncoding.pe", "InputLayerDecoder.PositionalEncoding.pe")


def _attn(prefix, d):
    return [(prefix + "in_proj_weight", (3 * d, d)), (prefix + "in_proj_bias", (3 * d,)),
            (prefix + "out_proj.weight", (d, d)), (prefix + "out_proj.bias", (d,))]


def param_names(d_model, dim_ff, src_dim, n_enc_layers, n_dec_layers=0, tgt_dim=27):
    d, F = d_model, dim_ff
    out = [("InputLayerEncoder.Linear.weight", (d, src_dim)), ("InputLayerEncoder.Linear.bias", (d,))]
    for l in range(n_enc_layers):
        p = "Encoder.Encoder.layers.%d." % l
        out += _attn(p + "self_attn.", d)
        out += [(p + "linear1.weight", (F, d)), (p + "linear1.bias", (F,)), (p + "linear2.weight", (d, F)),
                (p + "linear2.bias", (d,)), (p + "norm1.weight", (d,)), (p + "norm1.bias", (d,)),
                (p + "norm2.weight", (d,)), (p + "norm2.bias", (d,))]
    out += [("Encoder.Encoder.norm.weight", (d,)), ("Encoder.Encoder.norm.bias", (d,))]
    if n_dec_layers:
        out += [("InputLayerDecoder.Linear.weight", (d, tgt_dim)), ("InputLayerDecoder.Linear.bias", (d,))]
        for l in range(n_dec_layers):
            p = "Decoder.Decoder.layers.%d." % l
            out += _attn(p + "self_attn.", d) + _attn(p + "multihead_attn.", d)
            out += [(p + "linear1.weight", (F, d)), (p + "linear1.bias", (F,)), (p + "linear2.weight", (d, F)),
                    (p + "linear2.bias", (d,)), (p + "norm1.weight", (d,)), (p + "norm1.bias", (d,)),
                    (p + "norm2.weight", (d,)), (p + "norm2.bias", (d,)), (p + "norm3.weight", (d,)),
                    (p + "norm3.bias", (d,))]
        out += [("Decoder.Decoder.norm.weight", (d,)), ("Decoder.Decoder.norm.bias", (d,))]
    out += [("OutputLayer.Linear.weight", (tgt_dim, d)), ("OutputLayer.Linear.bias", (tgt_dim,))]
    return out


def positional_encoding(d_model, max_len=32):
    """The registered `pe` buffer, (max_len, d_model) fp32: interleaved sin/cos with
    div = exp(arange(0,d,2) * -ln(1e4)/d) -- equals the checkpoint's buffer to 6e-8."""
    pe = np.zeros((max_len, d_model), np.float32)
    pos = np.arange(max_len, dtype=np.float32)[:, None]
    div = np.exp(np.arange(0, d_model, 2).astype(np.float32) * np.float32(-math.log(10000.0) / d_model)).astype(np.float32)
    pe[:, 0::2] = np.sin(pos * div)
    pe[:, 1::2] = np.cos(pos * div)
    return pe


def init_params(dims, seed=0):
    """{state-dict name: fp32 array}: torch-default-style initial weights from a seeded numpy generator, so every rank
    (and every run) starts from the same replica without a broadcast.  Same distributions as
    ``_GrooveBase.reset_parameters``: xavier-uniform packed in-proj, zero attention biases, U(+-1/sqrt(fan_in)) linears,
    LayerNorm 1/0, IO layers U(+-0.1) with zero bias (ckpt: InputLayer weight range +-0.0995)."""
    d, F = dims["d_model"], dims["dim_feedforward"]
    r = np.random.default_rng(seed)
    out = {}
    for name, shape in param_names(d, F, dims["embedding_size_src"], dims["num_encoder_layers"], dims.get("num_decoder_layers", 0)):
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
        else:
            a = r.uniform(-1, 1, shape) / math.sqrt(d if "linear1" in name else F)
        out[name] = a.astype(np.float32)
    return out


def synthetic_batch(B, src_dim, seed=1234):
    """SURVEY 8(d) synthetic HVO batch: x ~ U[0,1) (B,32,src_dim); hits ~ Bernoulli(0.15), velocities = U*h,
    offsets = (U-0.5)*h -> y (B,32,27) = [h | v | o] (ref:utils.py:38-47 column order)."""
    r = np.random.default_rng(seed)
    x = r.random((B, 32, src_dim), dtype=np.float32)
    h = (r.random((B, 32, 9), dtype=np.float32) < 0.15).astype(np.float32)
    v = r.random((B, 32, 9), dtype=np.float32) * h
    o = (r.random((B, 32, 9), dtype=np.float32) - 0.5) * h
    return x, np.concatenate([h, v, o], -1)
