"""Stock-torch restatement of the reference's GrooveTransformer (TEST INFRASTRUCTURE ONLY).

The reference imports its model from the un-vendored submodule ``BaseGrooveTransformers``
(ref:train.py:12, ref:.gitmodules:4-6).  What that submodule builds is pinned by the reference's
demo checkpoint (ref:demo/transformer_run_171tyqit_Epoch_1.Model): the state-dict key names are
those of ``torch.nn.TransformerEncoder`` / ``MultiheadAttention`` / ``LayerNorm`` / ``Linear``
wrapped as ``InputLayerEncoder.{Linear,ReLU,PositionalEncoding}``, ``Encoder.Encoder`` and
``OutputLayer.Linear``.  This file wires stock torch modules under exactly those names so the
checkpoint strict-loads, and is used (a) as the checker in tests, (b) to generate
``tests/golden/*.npz`` and (c) as the CPU baseline that ``bench.py`` times.

Third-party op order followed (this container's torch 2.10; same math as the pinned 1.10.2):
  encoder layer post-norm  torch:nn/modules/transformer.py:951-956,961-982
  decoder layer post-norm  torch:nn/modules/transformer.py:1143-1153
  packed in-projection     torch:nn/functional.py:5820-5850
  attention core           torch:nn/functional.py:6504-6642
"""
import math

import torch
from torch import nn


class PositionalEncoding(nn.Module):
    """x + pe[:, :T] then dropout; ``pe`` is a registered buffer of shape (1, max_len, d)
    (ckpt key ``InputLayerEncoder.PositionalEncoding.pe``)."""

    def __init__(self, d_model, max_len=32, dropout=0.1):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0))

    def forward(self, x):
        return self.dropout(x + self.pe[:, : x.size(1)])


class InputLayer(nn.Module):
    def __init__(self, embedding_size, d_model, dropout, max_len):
        super().__init__()
        self.Linear = nn.Linear(embedding_size, d_model, bias=True)
        self.ReLU = nn.ReLU()
        self.PositionalEncoding = PositionalEncoding(d_model, max_len, dropout)

    def init_weights(self, initrange=0.1):
        self.Linear.bias.data.zero_()
        self.Linear.weight.data.uniform_(-initrange, initrange)

    def forward(self, src):
        return self.PositionalEncoding(self.ReLU(self.Linear(src)))


class OutputLayer(nn.Module):
    def __init__(self, embedding_size, d_model):
        super().__init__()
        self.embedding_size = embedding_size
        self.Linear = nn.Linear(d_model, embedding_size, bias=True)

    def init_weights(self, initrange=0.1):
        self.Linear.bias.data.zero_()
        self.Linear.weight.data.uniform_(-initrange, initrange)

    def forward(self, x):
        y = self.Linear(x)
        n = self.embedding_size // 3
        h = y[:, :, 0:n]
        v = torch.sigmoid(y[:, :, n:2 * n])
        o = torch.tanh(y[:, :, 2 * n:3 * n]) * 0.5
        return h, v, o


class Encoder(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward, dropout, num_layers):
        super().__init__()
        layer = nn.TransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout)
        self.Encoder = nn.TransformerEncoder(layer, num_layers, nn.LayerNorm(d_model),
                                             enable_nested_tensor=False)

    def forward(self, src):                       # (B,T,d) -> seq-first inside torch
        return self.Encoder(src.permute(1, 0, 2)).permute(1, 0, 2)


class Decoder(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward, dropout, num_layers):
        super().__init__()
        layer = nn.TransformerDecoderLayer(d_model, nhead, dim_feedforward, dropout)
        self.Decoder = nn.TransformerDecoder(layer, num_layers, nn.LayerNorm(d_model))

    def forward(self, tgt, memory, tgt_mask):
        out = self.Decoder(tgt.permute(1, 0, 2), memory.permute(1, 0, 2), tgt_mask=tgt_mask)
        return out.permute(1, 0, 2)


def get_tgt_mask(max_len):
    """Causal mask: 0 on/below the diagonal, -inf above."""
    return torch.triu(torch.full((max_len, max_len), float("-inf")), diagonal=1)


def _threshold(h_logits, use_thres, thres):
    p = torch.sigmoid(h_logits)
    if use_thres:
        return torch.where(p > thres, torch.ones_like(p), torch.zeros_like(p))
    return p


class GrooveTransformerEncoder(nn.Module):
    """encoder_only=1 model (every shipped YAML, ref:configs/*_training.yaml:11)."""

    def __init__(self, d_model=512, nhead=8, num_encoder_layers=6, dim_feedforward=2048,
                 dropout=0.1, embedding_size_src=16, embedding_size_tgt=27, max_len=32):
        super().__init__()
        self.max_len = max_len
        self.InputLayerEncoder = InputLayer(embedding_size_src, d_model, dropout, max_len)
        self.Encoder = Encoder(d_model, nhead, dim_feedforward, dropout, num_encoder_layers)
        self.OutputLayer = OutputLayer(embedding_size_tgt, d_model)
        self.InputLayerEncoder.init_weights()
        self.OutputLayer.init_weights()

    def forward(self, src):
        return self.OutputLayer(self.Encoder(self.InputLayerEncoder(src)))

    def predict(self, src, use_thres=True, thres=0.5):
        self.eval()
        with torch.no_grad():
            h, v, o = self.forward(src)
            return _threshold(h, use_thres, thres), v, o


class GrooveTransformer(nn.Module):
    """encoder_only=0 model (ref:train.py:125-127).  PARITY UNPINNED (no checkpoint/YAML)."""

    def __init__(self, d_model=512, nhead=8, num_encoder_layers=6, num_decoder_layers=6,
                 dim_feedforward=2048, dropout=0.1, embedding_size_src=16,
                 embedding_size_tgt=27, max_len=32):
        super().__init__()
        self.max_len = max_len
        self.embedding_size_tgt = embedding_size_tgt
        self.InputLayerEncoder = InputLayer(embedding_size_src, d_model, dropout, max_len)
        self.Encoder = Encoder(d_model, nhead, dim_feedforward, dropout, num_encoder_layers)
        self.InputLayerDecoder = InputLayer(embedding_size_tgt, d_model, dropout, max_len)
        self.Decoder = Decoder(d_model, nhead, dim_feedforward, dropout, num_decoder_layers)
        self.OutputLayer = OutputLayer(embedding_size_tgt, d_model)
        self.InputLayerEncoder.init_weights()
        self.InputLayerDecoder.init_weights()
        self.OutputLayer.init_weights()

    def forward(self, src, tgt):
        mask = get_tgt_mask(self.max_len).to(src.device)
        memory = self.Encoder(self.InputLayerEncoder(src))
        out = self.Decoder(self.InputLayerDecoder(tgt), memory, mask)
        return self.OutputLayer(out)

    def predict(self, src, use_thres=True, thres=0.5):
        """Greedy 32-step decode: tgt row 0 = zeros, row t+1 = [h|v|o] predicted for step t."""
        self.eval()
        with torch.no_grad():
            B, T = src.shape[0], self.max_len
            tgt = torch.zeros(B, T, self.embedding_size_tgt, dtype=src.dtype, device=src.device)
            out = torch.zeros_like(tgt)
            for t in range(T):
                h, v, o = self.forward(src, tgt)
                step = torch.cat([_threshold(h[:, t], use_thres, thres), v[:, t], o[:, t]], dim=-1)
                out[:, t] = step
                if t + 1 < T:
                    tgt[:, t + 1] = step
            n = self.embedding_size_tgt // 3
            return out[:, :, :n], out[:, :, n:2 * n], out[:, :, 2 * n:]


def calculate_loss(prediction, y, bce_fn, mse_fn, hit_loss_penalty):
    """loss_fn handed to train_loop (ref:train.py:201-203,213); bce_fn/mse_fn are
    BCEWithLogitsLoss/MSELoss(reduction="none") (ref:train.py:176-179); hit_loss_penalty is the
    "non_hit loss multiplier" (ref:train.py:55-58).  Formula PARITY UNPINNED (submodule)."""
    n = y.shape[2] // 3
    y_h, y_v, y_o = torch.split(y, n, 2)
    pred_h, pred_v, pred_o = prediction
    pen = torch.where(y_h == 1, float(1), float(hit_loss_penalty))
    bce_hits = (bce_fn(pred_h, y_h) * pen).sum(dim=2).mean()
    mse_velocities = (mse_fn(pred_v, y_v) * pen).sum(dim=2).mean()
    mse_offsets = (mse_fn(pred_o, y_o) * pen).sum(dim=2).mean()
    total = bce_hits + mse_velocities + mse_offsets
    h = torch.where(torch.sigmoid(pred_h) > 0.5, 1, 0)
    hit_accuracy = (torch.eq(h.reshape(h.shape[0], -1), y_h.reshape(h.shape[0], -1)).sum(-1)
                    / float(h.shape[1] * h.shape[2])).mean()
    hit_perplexity = torch.exp(bce_hits)
    return (total, hit_accuracy.item(), hit_perplexity.item(), bce_hits.item(),
            mse_velocities.item(), mse_offsets.item())


def build(cfg, seed=0):
    """cfg: dict(d_model, n_heads, dim_feedforward, num_encoder_layers, num_decoder_layers,
    dropout, embedding_size_src, embedding_size_tgt, max_len)."""
    torch.manual_seed(seed)
    common = dict(d_model=cfg["d_model"], nhead=cfg["n_heads"],
                  dim_feedforward=cfg["dim_feedforward"], dropout=cfg.get("dropout", 0.0),
                  embedding_size_src=cfg.get("embedding_size_src", 16),
                  embedding_size_tgt=cfg.get("embedding_size_tgt", 27),
                  max_len=cfg.get("max_len", 32))
    if cfg.get("num_decoder_layers", 0) == 0:
        return GrooveTransformerEncoder(num_encoder_layers=cfg["num_encoder_layers"], **common)
    return GrooveTransformer(num_encoder_layers=cfg["num_encoder_layers"],
                             num_decoder_layers=cfg["num_decoder_layers"], **common)


def shift_right(y):
    """Teacher-forcing decoder input: y shifted one step right, row 0 zeros."""
    return torch.cat([torch.zeros_like(y[:, :1]), y[:, :-1]], dim=1)


def train_step(model, opt, x, y, hit_loss_penalty, encoder_only=True):
    """One reference-shaped step: zero_grad, forward, calculate_loss, backward, opt.step."""
    bce = nn.BCEWithLogitsLoss(reduction="none")
    mse = nn.MSELoss(reduction="none")
    opt.zero_grad()
    pred = model(x) if encoder_only else model(x, shift_right(y))
    out = calculate_loss(pred, y, bce, mse, hit_loss_penalty)
    out[0].backward()
    opt.step()
    return out


def forward_autocast(P, cfg, x, tgt=None):
    """Eval-mode forward of the stock modules under ``torch.autocast("cpu", dtype=torch.bfloat16)`` and, beside it, in fp32 -- the
    third-party definition of "bf16 where the bytes are" that gt_config.precision = 2 is anchored on (tests/: the device's outputs must sit
    within the bf16 effect of this run).  P: {state-dict name: array}.  -> (autocast [h | v | o], fp32 [h | v | o]) as float64 arrays."""
    import numpy as np
    m = build(dict(cfg, dropout=0.0))
    sd = m.state_dict()
    for k in sd:
        if k in P:
            sd[k] = torch.from_numpy(np.asarray(P[k], np.float32)).reshape(sd[k].shape)
    m.load_state_dict(sd)
    m.eval()
    xt = torch.from_numpy(np.asarray(x, np.float32))
    args = (xt,) if tgt is None else (xt, torch.from_numpy(np.asarray(tgt, np.float32)))
    with torch.no_grad():
        ref = torch.cat([t.float() for t in m(*args)], -1).double().numpy()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out = torch.cat([t.float() for t in m(*args)], -1).double().numpy()
    return out, ref
