"""Shared test driver: runs the C ABI (include/groove_hip.h) on either the real HIP library
(torch CUDA buffers) or the host fiber-emulator build of the same sources (numpy buffers), and
exposes results as numpy arrays for comparison with the oracle."""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transformergrooveinfilling_amd import _lib, layout  # noqa: E402

EMU_SO = os.environ.get("GT_EMU_LIB_PATH") or os.path.join(ROOT, "tests", "emu", "libgroove_emu.so")   # override: tile-rule variants
_SRC = [os.path.join(ROOT, "transformergrooveinfilling_amd", "csrc", f)
        for f in ("groove_hip.hip", "groove_seq_fwd.hip", "groove_seq_bwd.hip", "groove_seq64.hip", "gt_common.h", "gt_gemm.h", "gt_gemm32.h", "gt_gemm64.h", "gt_seq.h", "gt_seq_wg.h",
                  "gt_seq_api.h", "gt_attn.h", "gt_misc.h")] + \
       [os.path.join(ROOT, "tests", "emu", "hip_emu.h"), os.path.join(ROOT, "include", "groove_hip.h")]


def emu_lib():
    if (not os.path.exists(EMU_SO)) or any(os.path.getmtime(s) > os.path.getmtime(EMU_SO) for s in _SRC):
        subprocess.check_call([os.path.join(ROOT, "tests", "emu", "build_emu.sh")], stdout=subprocess.DEVNULL)
    return _lib.GrooveLib(EMU_SO)


class NpBuf:
    def __init__(self, arr):
        self.a = np.ascontiguousarray(arr)
        self.ptr = ctypes.c_void_p(self.a.ctypes.data)

    def numpy(self):
        return self.a


class CudaBuf:
    def __init__(self, arr):
        import torch
        self.t = torch.from_numpy(np.ascontiguousarray(arr)).cuda()
        self.ptr = ctypes.c_void_p(self.t.data_ptr())

    def numpy(self):
        import torch
        torch.cuda.synchronize()
        return self.t.cpu().numpy()


def cfg_dict(d_model, n_heads, dim_feedforward, num_encoder_layers, num_decoder_layers=0, embedding_size_src=16,
             dropout=0.0, precision=0):
    d = dict(d_model=d_model, n_heads=n_heads, dim_feedforward=dim_feedforward,
             num_encoder_layers=num_encoder_layers, num_decoder_layers=num_decoder_layers,
             embedding_size_src=embedding_size_src, dropout=dropout)
    if precision:
        d["precision"] = precision                 # 1 = bf16 GEMM operands (gt_config.precision)
    return d


class Runner:
    """One model instance behind the C ABI.  backend = 'emu' | 'hip'."""

    def __init__(self, cfg, B, backend="emu", rng=(1234, 99, 0), lr=0.094, seq=True):
        self.cfgd, self.B, self.backend = cfg, B, backend
        self.lib = emu_lib() if backend == "emu" else _lib.get_lib()
        self.lib.cdll.gt_set_seq(int(bool(seq)))  # process-global switch: sequence-resident kernels (default where supported)
        # seq = "split" / "whole": force / forbid their two-workgroups-per-sequence mode (d_model 128); True: the library's choice
        self.lib.cdll.gt_set_seq_split(1 if seq in ("split", "split-noride", "split-noquad") else 0 if seq == "whole" else -1)
        # the SPLIT forward with four workgroups per sequence (column partners + pair exchange) is the library's choice wherever
        # 4 x batch workgroups fit the chip; "split-noquad": two workgroups per sequence in the forward too
        self.lib.cdll.gt_set_seq_quad(0 if seq == "split-noquad" else -1)
        # "split": weight gradients as rider workgroups of the backward phases where the library chooses to (idle CUs); "split-noride":
        # the grouped dispatch at the end of backward
        self.lib.cdll.gt_set_seq_ride(0 if seq == "split-noride" else -1)
        self.Buf = NpBuf if backend == "emu" else CudaBuf
        self.c = _lib.make_config(B, cfg["embedding_size_src"], cfg["d_model"], cfg["n_heads"], cfg["dim_feedforward"],
                                  cfg["num_encoder_layers"], cfg.get("num_decoder_layers", 0), cfg.get("dropout", 0.0),
                                  cfg.get("precision", 0))
        self.total, self.entries = self.lib.param_layout(self.c)
        self.names = layout.param_names(cfg["d_model"], cfg["dim_feedforward"], cfg["embedding_size_src"],
                                        cfg["num_encoder_layers"], cfg.get("num_decoder_layers", 0))
        assert len(self.names) == len(self.entries)
        for (n, shp), (off, size, rows, cols) in zip(self.names, self.entries):
            assert int(np.prod(shp)) == size and shp[0] == rows, (n, shp, size, rows, cols)
        self.M = B * 32
        self.ws = self.Buf(np.full(self.lib.workspace_floats(self.c), 7.25, np.float32))     # (garbage: gt_workspace_init zeroes what must be zero)
        self.lib.call("gt_workspace_init", ctypes.byref(self.c), self.ws.ptr, ctypes.c_void_p(0))
        self.pe = self.Buf(layout.positional_encoding(cfg["d_model"]))
        self.hvo = self.Buf(np.zeros((self.M, 27), np.float32))
        self.grads = self.Buf(np.zeros(self.total, np.float32))
        self.stats = self.Buf(np.zeros(8, np.float32))
        self.d_hvo = self.Buf(np.zeros((self.M, 27), np.float32))
        self.tgt = self.Buf(np.zeros((self.M, 27), np.float32))
        st = _lib.GtStepState(rng[0], rng[1], rng[2], 0, lr, 1.0, 0.9, 0.999, 1e-8)
        self.state = self.Buf(np.frombuffer(bytes(st), dtype=np.uint8).copy())
        self.stream = ctypes.c_void_p(0)
        self.params = None

    # ---- parameters -------------------------------------------------------------------------
    def flatten(self, P):
        flat = np.zeros(self.total, np.float32)
        for (n, shp), (off, size, _, _) in zip(self.names, self.entries):
            flat[off:off + size] = np.asarray(P[n], np.float32).reshape(-1)
        return flat

    def unflatten(self, flat):
        return {n: flat[off:off + size].reshape(shp).copy() for (n, shp), (off, size, _, _) in zip(self.names, self.entries)}

    def set_params(self, P):
        self.params = self.Buf(self.flatten(P))

    # ---- calls ------------------------------------------------------------------------------
    def forward(self, x, tgt_in=None, train=False):
        self.x = self.Buf(np.asarray(x, np.float32).reshape(self.M, -1))
        self.tgt_in = self.Buf(np.asarray(tgt_in, np.float32).reshape(self.M, 27)) if tgt_in is not None else None
        self.lib.call("gt_forward", ctypes.byref(self.c), self.params.ptr, self.pe.ptr, self.x.ptr,
                      self.tgt_in.ptr if self.tgt_in else None, self.hvo.ptr, self.ws.ptr, self.state.ptr, int(train),
                      self.stream)
        return self.hvo.numpy().reshape(self.B, 32, 27).copy()

    def loss(self, y, penalty, want_grad=True):
        self.y = self.Buf(np.asarray(y, np.float32).reshape(self.M, 27))
        self.lib.call("gt_loss", ctypes.byref(self.c), self.hvo.ptr, self.y.ptr, ctypes.c_float(penalty), self.stats.ptr,
                      self.d_hvo.ptr if want_grad else None, self.stream)
        return self.stats.numpy().copy(), self.d_hvo.numpy().reshape(self.B, 32, 27).copy()

    def backward(self, d_hvo=None, train=False):
        if d_hvo is not None:
            self.d_hvo = self.Buf(np.asarray(d_hvo, np.float32).reshape(self.M, 27))
        self.lib.call("gt_backward", ctypes.byref(self.c), self.params.ptr, self.grads.ptr, self.x.ptr,
                      self.tgt_in.ptr if self.tgt_in else None, self.hvo.ptr, self.d_hvo.ptr, self.ws.ptr, self.state.ptr,
                      int(train), 0, self.stream)
        self._grads_dirty = True
        return self.unflatten(self.grads.numpy())

    def optimizer_step(self, algo=0):
        if algo == 1 and not hasattr(self, "m"):
            self.m, self.v = self.Buf(np.zeros(self.total, np.float32)), self.Buf(np.zeros(self.total, np.float32))
        self.lib.call("gt_optimizer_step", algo, self.params.ptr, self.grads.ptr, self.m.ptr if algo == 1 else None,
                      self.v.ptr if algo == 1 else None, ctypes.c_int64(self.total), self.state.ptr, 0, self.stream)
        return self.unflatten(self.params.numpy())

    def train_step(self, x, y, penalty, algo=0, skip_update=False):
        self.x = self.Buf(np.asarray(x, np.float32).reshape(self.M, -1))
        self.y = self.Buf(np.asarray(y, np.float32).reshape(self.M, 27))
        if algo == 1 and not hasattr(self, "m"):
            self.m, self.v = self.Buf(np.zeros(self.total, np.float32)), self.Buf(np.zeros(self.total, np.float32))
        if getattr(self, "_grads_dirty", False):           # gt_train_step precondition: grads are zero on entry
            self.grads = self.Buf(np.zeros(self.total, np.float32))
            self._grads_dirty = False
        self.lib.call("gt_train_step", ctypes.byref(self.c), algo, self.params.ptr, self.grads.ptr,
                      self.m.ptr if algo == 1 else None, self.v.ptr if algo == 1 else None, self.pe.ptr, self.x.ptr, self.y.ptr,
                      ctypes.c_float(penalty), self.hvo.ptr, self.stats.ptr, self.tgt.ptr, self.ws.ptr, self.state.ptr,
                      int(skip_update), self.stream)
        return self.stats.numpy().copy()

    def predict(self, x, thres=0.5, use_thres=True, pd_seed=None):
        self.x = self.Buf(np.asarray(x, np.float32).reshape(self.M, -1))
        if pd_seed is not None:
            self.lib.call("gt_predict_pd", ctypes.byref(self.c), self.params.ptr, self.pe.ptr, self.x.ptr, self.hvo.ptr,
                          ctypes.c_uint32(pd_seed), self.tgt.ptr, self.ws.ptr, self.stream)
            return self.hvo.numpy().reshape(self.B, 32, 27).copy()
        self.lib.call("gt_predict", ctypes.byref(self.c), self.params.ptr, self.pe.ptr, self.x.ptr, self.hvo.ptr,
                      ctypes.c_float(thres), int(use_thres), self.tgt.ptr, self.ws.ptr, self.stream)
        return self.hvo.numpy().reshape(self.B, 32, 27).copy()

    BF16_ONLY = ("ctx", "hact", "dhid", "dqkv", "dzAm", "dzBm")

    def precision_in_force(self):
        """0 / 1 / 2: what gt_config.precision really runs as for this shape (2 needs the level-2 operand shadows and 64- / 128-wide heads)"""
        return int(self.lib.cdll.gt_precision_in_force(ctypes.byref(self.c)))

    def bf16_only(self, name, layer=0):
        """Is this saved tensor stored in bf16 alone (gt_set_operand_shadows level 2: the encoder layers' operand-only tensors;
        precision 2: qkv as well)?"""
        if name == "qkv":
            return layer < self.cfgd["num_encoder_layers"] and self.precision_in_force() == 2
        return (name in self.BF16_ONLY and layer < self.cfgd["num_encoder_layers"]
                and self.lib.cdll.gt_operand_shadow_level(ctypes.byref(self.c)) == 2)

    def ws_get(self, name, layer=0):
        off, cnt = self.lib.ws_find(self.c, name, layer)
        if self.bf16_only(name, layer):           # the live tensor is "<name>16": widened to fp32 (exact)
            o16, _ = self.lib.ws_find(self.c, name + "16", layer)
            u = self.ws.numpy()[o16:o16 + (cnt + 1) // 2].view(np.uint16)[:cnt].astype(np.uint32) << 16
            return u.view(np.float32).copy()
        return self.ws.numpy()[off:off + cnt].copy()

    def step_state(self):
        return _lib.GtStepState.from_buffer_copy(self.state.numpy().tobytes())


def free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def run_ranks(fn, world, *args):
    """mp.start_processes(fn, (world, port) + args) on a free rendezvous port -- once more on a fresh port when the first attempt dies (the
    port is picked, released and re-bound by rank 0: two pytest-xdist workers running multi-rank tests at once can pick the same one)."""
    import torch.multiprocessing as mp
    for attempt in (0, 1):
        try:
            mp.start_processes(fn, args=(world, free_port()) + tuple(args), nprocs=world, join=True, start_method="spawn")
            return
        except Exception:
            if attempt:
                raise
