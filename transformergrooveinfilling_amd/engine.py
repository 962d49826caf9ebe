"""StepEngine: device-resident state of one GrooveTransformer replica and the per-step hot path.

Everything a train step touches lives in HBM for the life of the engine: ONE flat fp32 parameter
buffer (state-dict order, gt_param_layout), one flat gradient buffer of the same layout (a single RCCL
all-reduce and a single fused optimizer launch), optimizer moments, the activation workspace, the
positional-encoding buffer, the 48-byte device step state (dropout seed/step, lr, Adam betas) and
static input/output buffers.  A whole step (forward, loss, backward, update) is one call into
libgroove_hip.so and is captured once into a hipGraph (torch.cuda.CUDAGraph) and replayed.

Replaces, for the hot path only, the body of the reference's train_loop batch iteration
(ref:train.py:195-215: zero_grad / forward / calculate_loss / backward / opt.step) and
model.predict (ref:evaluator.py:173).  Data-parallel: one process per GPU, gradients summed with one
all-reduce over the flat buffer and averaged inside the optimizer kernel (grad_scale = 1/world).
"""
import ctypes

import numpy as np
import torch

from . import _lib, layout

ALGO = {"sgd": 0, "adam": 1}


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class StepEngine:
    def __init__(self, d_model, n_heads, dim_feedforward, num_encoder_layers, num_decoder_layers=0,
                 dropout=0.0, embedding_size_src=16, batch_size=64, optimizer="sgd", learning_rate=0.05,
                 hit_loss_penalty=1.0, seed=0, device="cuda", world_size=1, use_graph=True, lib=None):
        if not torch.cuda.is_available():
            raise RuntimeError("StepEngine needs a ROCm GPU (torch.cuda.is_available() is False); "
                               "there is no CPU fallback for the hot path")
        self.lib = lib or _lib.get_lib()
        self.device = torch.device(device)
        self.B, self.M = int(batch_size), int(batch_size) * 32
        self.S = int(embedding_size_src)
        self.encoder_only = num_decoder_layers == 0
        self.algo = ALGO[optimizer.lower()]
        self.penalty = float(hit_loss_penalty)
        self.world_size = int(world_size)
        self.use_graph = use_graph
        self.dims = dict(d_model=d_model, n_heads=n_heads, dim_feedforward=dim_feedforward,
                         num_encoder_layers=num_encoder_layers, num_decoder_layers=num_decoder_layers,
                         dropout=dropout, embedding_size_src=embedding_size_src)
        self.cfg = _lib.make_config(self.B, self.S, d_model, n_heads, dim_feedforward, num_encoder_layers,
                                    num_decoder_layers, dropout)
        self.total, self.entries = self.lib.param_layout(self.cfg)
        self.names = layout.param_names(d_model, dim_feedforward, self.S, num_encoder_layers, num_decoder_layers)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.params = torch.zeros(self.total, **f32)
        self.grads = torch.zeros(self.total, **f32)
        self.m = torch.zeros(self.total, **f32) if self.algo == 1 else None
        self.v = torch.zeros(self.total, **f32) if self.algo == 1 else None
        self.pe = torch.from_numpy(layout.positional_encoding(d_model)).to(self.device)
        self.x = torch.zeros(self.B, 32, self.S, **f32)
        self.y = torch.zeros(self.B, 32, 27, **f32)
        self.hvo = torch.zeros(self.B, 32, 27, **f32)
        self.tgt = torch.zeros(self.B, 32, 27, **f32)
        self.stats = torch.zeros(8, **f32)
        self._ws = {}            # batch -> workspace (predict may run other batch sizes)
        self.ws = self._workspace(self.cfg)
        st = _lib.GtStepState(seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, 0, 0, learning_rate,
                              1.0 / self.world_size, 0.9, 0.999, 1e-8)
        self.state = torch.from_numpy(np.frombuffer(bytes(st), dtype=np.uint8).copy()).to(self.device)
        self._graph = None
        self._graph_key = None

    # ---- buffers ---------------------------------------------------------------------------------
    def _workspace(self, cfg):
        n = self.lib.workspace_floats(cfg)
        key = cfg.batch
        if key not in self._ws or self._ws[key].numel() < n:
            self._ws[key] = torch.empty(n, dtype=torch.float32, device=self.device)
        return self._ws[key]

    def views(self, flat=None):
        """name -> view into the flat parameter (or gradient/moment) buffer, state-dict order."""
        flat = self.params if flat is None else flat
        return {n: flat[off:off + size].view(*shp) for (n, shp), (off, size, _, _) in zip(self.names, self.entries)}

    def load_named(self, tensors):
        """Copy {state-dict name: tensor/ndarray} into the flat buffer (pe buffers are ignored)."""
        v = self.views()
        for n, t in tensors.items():
            if n in layout.PE_KEYS:
                continue
            if n not in v:
                raise KeyError("unexpected parameter %r" % n)
            v[n].copy_(torch.as_tensor(np.asarray(t) if not torch.is_tensor(t) else t).to(self.device).view_as(v[n]))

    def state_struct(self):
        return _lib.GtStepState.from_buffer_copy(self.state.cpu().numpy().tobytes())

    def set_state(self, **kw):
        st = self.state_struct()
        for k, val in kw.items():
            setattr(st, k, val)
        self.state.copy_(torch.from_numpy(np.frombuffer(bytes(st), dtype=np.uint8).copy()))

    @property
    def stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- the hot path ----------------------------------------------------------------------------
    def _enqueue_step(self, skip_update):
        self.lib.call("gt_train_step", ctypes.byref(self.cfg), self.algo, _ptr(self.params), _ptr(self.grads),
                      _ptr(self.m), _ptr(self.v), _ptr(self.pe), _ptr(self.x), _ptr(self.y),
                      ctypes.c_float(self.penalty), _ptr(self.hvo), _ptr(self.stats), _ptr(self.tgt), _ptr(self.ws),
                      _ptr(self.state), int(skip_update), self.stream)

    def _enqueue_update(self):
        self.lib.call("gt_optimizer_step", self.algo, _ptr(self.params), _ptr(self.grads), _ptr(self.m), _ptr(self.v),
                      ctypes.c_int64(self.total), _ptr(self.state), self.stream)

    def _replay(self, key, fn):
        if not self.use_graph:
            fn()
            return
        if self._graph_key != key:
            # warm-up launch outside capture (module load), then capture once
            s = torch.cuda.Stream(self.device)
            s.wait_stream(torch.cuda.current_stream(self.device))
            snap = (self.params.clone(), self.state.clone(), None if self.m is None else (self.m.clone(), self.v.clone()))
            with torch.cuda.stream(s):
                fn()
            torch.cuda.current_stream(self.device).wait_stream(s)
            torch.cuda.synchronize(self.device)
            self.params.copy_(snap[0]); self.state.copy_(snap[1])
            if snap[2] is not None:
                self.m.copy_(snap[2][0]); self.v.copy_(snap[2][1])
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            self._graph, self._graph_key = g, key
        self._graph.replay()

    def train_step(self, x=None, y=None):
        """One optimisation step on (x, y) (device or host tensors; None = reuse the static buffers).
        Returns the device stats tensor [loss, hit_acc, -, bce, mse_v, mse_o, -, -] without syncing."""
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        if y is not None:
            self.y.copy_(y, non_blocking=True)
        if self.world_size == 1:
            self._replay("fused", lambda: self._enqueue_step(0))
        else:
            import torch.distributed as dist
            self._replay("fwdbwd", lambda: self._enqueue_step(1))
            dist.all_reduce(self.grads)                      # RCCL sum over xGMI; averaged by grad_scale
            self._enqueue_update()
        return self.stats

    def forward(self, x, tgt_in=None, train=False):
        """(h_logits, v, o) views of the (B,32,27) HVO buffer for a batch of the engine's size."""
        self.x.copy_(x)
        if tgt_in is not None:
            self.tgt.copy_(tgt_in)
        self.lib.call("gt_forward", ctypes.byref(self.cfg), _ptr(self.params), _ptr(self.pe), _ptr(self.x),
                      None if self.encoder_only else _ptr(self.tgt), _ptr(self.hvo), _ptr(self.ws), _ptr(self.state),
                      int(train), self.stream)
        return self.hvo

    def predict(self, x, use_thres=True, thres=0.5):
        """model.predict for ANY batch size (ref:evaluator.py:173 passes the whole evaluation set at once):
        returns a (N,32,27) HVO tensor on the device ([h | v | o], one D2H for the evaluator)."""
        x = torch.as_tensor(x, dtype=torch.float32).to(self.device).contiguous()
        n = x.shape[0]
        cfg = _lib.make_config(n, self.S, self.cfg.d_model, self.cfg.n_heads, self.cfg.dim_ff, self.cfg.n_enc_layers,
                               self.cfg.n_dec_layers, self.cfg.dropout)
        ws = self._workspace(cfg)
        out = torch.empty(n, 32, 27, dtype=torch.float32, device=self.device)
        tgt = torch.empty(n, 32, 27, dtype=torch.float32, device=self.device) if not self.encoder_only else None
        self.lib.call("gt_predict", ctypes.byref(cfg), _ptr(self.params), _ptr(self.pe), _ptr(x), _ptr(out),
                      ctypes.c_float(thres), int(use_thres), _ptr(tgt), _ptr(ws), self.stream)
        return out

    def profile(self, steps):
        """Eager (no graph) pass of `steps` train steps with HIP events around every launch.
        -> {kernel class: (launches, total_ms, total_flops, total_bytes)}.  Measurement aid for bench.py."""
        snap = (self.params.clone(), self.state.clone())
        torch.cuda.synchronize(self.device)
        self.lib.cdll.gt_profile_enable(1)
        try:
            for _ in range(steps):
                self._enqueue_step(0)
            buf = ctypes.create_string_buffer(1 << 16)
            self.lib.cdll.gt_profile_report(buf, len(buf), 256)
        finally:
            self.lib.cdll.gt_profile_enable(0)
        self.params.copy_(snap[0]); self.state.copy_(snap[1])
        out = {}
        for line in buf.value.decode().splitlines():
            lab, cnt, ms, fl, by = line.split()
            out[lab] = (int(cnt), float(ms), float(fl), float(by))
        return out
