"""StepEngine: device-resident state of one GrooveTransformer replica and the per-step hot path.

Everything a train step touches lives in HBM for the life of the engine: ONE flat fp32 parameter
buffer (state-dict order, gt_param_layout), one flat gradient buffer of the same layout (a single RCCL
all-reduce and a single fused optimizer launch), optimizer moments, the positional-encoding buffer and
the 48-byte device step state (dropout seed/step, lr, Adam betas).  Per batch size there is a "slot":
activation workspace, static input/output buffers and the captured hipGraph of the whole step
(forward, loss, backward, update = one call into libgroove_hip.so, captured once, replayed).

Replaces, for the hot path only, the body of the reference's train_loop batch iteration
(ref:train.py:195-215: zero_grad / forward / calculate_loss / backward / opt.step) and
model.predict (ref:evaluator.py:173).  Data-parallel: one process per GPU, gradients summed with one
all-reduce over the flat buffer and averaged inside the optimizer kernel (grad_scale = 1/world).
"""
import ctypes
import os
import warnings

import numpy as np
import torch

from . import _lib, layout

ALGO = {"sgd": 0, "adam": 1}
OVERLAP_MIN_BYTES = 16 << 20
EAGER_MAX_LAUNCHES = 24                # use_graph="auto": steps of at most this many launches are enqueued directly, not replayed
PREDICT_CHUNK = 512                    # floor of the sequences per gt_predict call
PREDICT_WS_BYTES = 32 << 30            # ... the chunk grows (x2) while its workspace stays under this and under half the free HBM
PREDICT_WS_KEEP = 4 << 30              # predict() keeps a workspace between calls only up to this size (a per-epoch evaluation must not pin tens of GiB)
MAX_GRAPHS = 4                         # captured step graphs kept per slot (least recently used dropped first)
MAX_SLOTS = 6                          # per-batch-size step slots kept (least recently created dropped first; the engine's own batch size stays)
XCHG_POLL_EVERY = 16                   # train steps between two asynchronous reads of the in-launch exchanges' error word (see poll_exchange; the first
                                       # read starts with the slot's first step)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class _Slot:
    """Buffers and captured graph for one batch size."""

    def __init__(self, eng, B):
        f32 = dict(dtype=torch.float32, device=eng.device)
        d = eng.dims
        self.B = B
        self.cfg = _lib.make_config(B, d["embedding_size_src"], d["d_model"], d["n_heads"], d["dim_feedforward"],
                                    d["num_encoder_layers"], d["num_decoder_layers"], d["dropout"], d["precision"], eng.cfg_flags)
        self.layout_epoch = eng.lib.cdll.gt_layout_epoch()      # (the workspace below is laid out for the switches in force NOW: StepEngine.slot)
        self.ws = torch.empty(eng.lib.workspace_floats(self.cfg), **f32)
        eng.lib.call("gt_workspace_init", ctypes.byref(self.cfg), _ptr(self.ws), eng.stream)   # (zeroes the regions whose protocol relies on it)
        self.x = torch.zeros(B, 32, d["embedding_size_src"], **f32)
        self.y = torch.zeros(B, 32, 27, **f32)
        self.hvo = torch.zeros(B, 32, 27, **f32)
        self.tgt = torch.zeros(B, 32, 27, **f32)
        self.stats = torch.zeros(8, **f32)
        self.idx = torch.zeros(B, dtype=torch.int64, device=eng.device)     # static batch indices of the indexed step
        self.graphs = {}               # step recipe -> captured hipGraph
        self.keep = {}                 # step recipe -> tensors whose raw pointers its graph holds
        self.use_graph = None          # StepEngine.graph_for's decision for this slot (use_graph="auto")
        self.fwd_id = 0                # bumped by every call that overwrites the saved activations (see StepEngine.forward)
        self.pack_epoch = -1           # StepEngine._pepoch at which this workspace's fragment-ordered weight copies were written
        self.xchg_off = None           # workspace offset of the QUAD pair-exchange region's error word (-1: none; looked up lazily)
        self.xchg_host = None          # pinned [error word, skipped updates] (data-parallel: the all-reduced guard element) copied asynchronously,
        self.xchg_event = None         # every XCHG_POLL_EVERY-th train step ... and the event behind that copy
        self.xchg_count = 0
        self.xchg_skipped_seen = 0     # value of the region's skipped-update counter at the last look (the counter restarts with the region)


class _LossSlot:
    """Buffers of one calculate_loss call size (no activation workspace: an evaluation set of thousands of sequences only
    needs its (N,32,27) predictions and targets here)."""

    def __init__(self, eng, B):
        f32 = dict(dtype=torch.float32, device=eng.device)
        d = eng.dims
        self.B = B
        self.cfg = _lib.make_config(B, d["embedding_size_src"], d["d_model"], d["n_heads"], d["dim_feedforward"],
                                    d["num_encoder_layers"], d["num_decoder_layers"], d["dropout"], d["precision"])
        self.hvo = torch.zeros(B, 32, 27, **f32)
        self.y = torch.zeros(B, 32, 27, **f32)
        self.stats = torch.zeros(8, **f32)


class StepEngine:
    def __init__(self, d_model, n_heads, dim_feedforward, num_encoder_layers, num_decoder_layers=0,
                 dropout=0.0, embedding_size_src=16, batch_size=None, optimizer="sgd", learning_rate=0.05,
                 hit_loss_penalty=1.0, seed=0, device="cuda", world_size=1, use_graph="auto", lib=None, precision="fp32"):
        self.device = torch.device(device)
        # The only way onto host memory is an EXPLICITLY passed library object (tests hand in the host-emulator build of
        # the same kernel sources to cover the multi-rank step sequence over gloo); nothing in the package does that.
        self.on_host = self.device.type == "cpu"
        if (self.on_host and lib is None) or (not self.on_host and not torch.cuda.is_available()):
            raise RuntimeError("StepEngine needs a ROCm GPU (torch.cuda.is_available() is False); "
                               "there is no CPU fallback for the hot path")
        self.lib = lib or _lib.get_lib()
        if not self.on_host and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.encoder_only = num_decoder_layers == 0
        self.algo = ALGO[optimizer.lower()]
        self.penalty = float(hit_loss_penalty)
        self.world_size = int(world_size)
        self.force_dp = False          # measurement aid: take the data-parallel step sequence even with one rank
        # True: one captured hipGraph per step recipe; False: plain launches; "auto" (default): a graph unless the step is a dozen
        # launches (sequence-resident path, gt_step_launches) -- there the graph's per-node cost (~0.4 us) exceeds what it saves
        # (headline shape: 0.246 ms replayed, 0.241 ms enqueued directly) and the one C call per step keeps the host ahead anyway
        self.use_graph = False if self.on_host else use_graph
        self.dims = dict(d_model=int(d_model), n_heads=int(n_heads), dim_feedforward=int(dim_feedforward),
                         num_encoder_layers=int(num_encoder_layers), num_decoder_layers=int(num_decoder_layers),
                         dropout=float(dropout), embedding_size_src=int(embedding_size_src),
                         precision=_lib.PRECISION[precision])     # 0 fp32 | 1 bf16 GEMM operands (BASELINE configs[4])
        probe = _lib.make_config(1, embedding_size_src, d_model, n_heads, dim_feedforward, num_encoder_layers,
                                 num_decoder_layers, dropout)
        self.total, self.entries = self.lib.param_layout(probe)
        self.names = layout.param_names(d_model, dim_feedforward, embedding_size_src, num_encoder_layers, num_decoder_layers)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.params = torch.zeros(self.total, **f32)
        self.grads = torch.zeros(self.total, **f32)
        self.m = torch.zeros(self.total, **f32) if self.algo == 1 else None
        self.v = torch.zeros(self.total, **f32) if self.algo == 1 else None
        self.pe = torch.from_numpy(layout.positional_encoding(d_model)).to(self.device)
        st = _lib.GtStepState(seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, 0, 0, learning_rate,
                              1.0 / self.world_size, 0.9, 0.999, 1e-8)
        self.state = torch.from_numpy(np.frombuffer(bytes(st), dtype=np.uint8).copy()).to(self.device)
        # Two gradient buckets, the first all-reduced under the rest of backward.  Splitting the step into two graphs costs
        # ~90 us per step on one GPU (0.434 vs 0.344 ms at the C2 shape, bench.py --force-dp), more than a 2.4 MB all-reduce
        # takes, so it is chosen by gradient bytes: from 16 MB (C3 31.7 MB, C4 38 MB) the all-reduce is several hundred
        # microseconds and worth hiding.  GT_DP_OVERLAP=0 / 1 forces it off / on.
        env = os.environ.get("GT_DP_OVERLAP")
        self.overlap_allreduce = (env == "1") if env in ("0", "1") else (4 * self.total >= OVERLAP_MIN_BYTES)
        self.reduce_stats = True       # data-parallel: the logged 8-float stats are averaged over ranks (one tiny all-reduce)
        # GT_DP_GRAPH=1: the whole data-parallel step (collectives included) as ONE captured hipGraph per step (measured with
        # bench.py --force-dp; off by default: it could not be exercised with more than one rank on the 1-GPU boxes of the build)
        self.dp_graph = os.environ.get("GT_DP_GRAPH", "0") == "1"
        self.dp_graph_failed = False   # a capture of the one-enqueue step failed once: eager sequence from then on (_dp_whole)
        self.dp_tune = None            # autotune_dp's table: {mode: ms per step (max over ranks)} and the mode kept
        self._capturing = False        # inside _replay's warm-up launch / capture: the recipe being recorded must not depend on the moment
        self._slots = {}
        # Fused whole steps on the sequence-resident path end with an update that also writes the next step's fragment-ordered
        # weights into the slot's workspace (GT_STEP_PACKS_CURRENT): valid while nothing else has written the parameters --
        # _pepoch counts the engine's own writes, params._version torch's (load_state_dict, init, in-place ops on the Parameters)
        self._pepoch, self._pver = 0, -1
        self.fold_pack = os.environ.get("GT_PACK_FOLD", "1") != "0"
        self._loss_slots = {}
        self._train_B = None           # batch size of the most recent train step (its slot is never evicted)
        self._predict_ws = {}          # chunk size -> (cfg, workspace, tgt scratch) of predict()
        self._predict_epoch = None     # gt_layout_epoch() those were sized at
        # gt_config.flags of every configuration this engine hands to the library (per ENGINE, not per process: train.py's evaluation engines and
        # bench.py's second engine keep their own).  _flags_fallback: set for good once an in-launch exchange timed out (no QUAD, no row exchange);
        # _flags_recipe: what the data-parallel recipe in force asks for (bucketed overlap: no row exchange beside a collective's workgroups)
        self._flags_fallback, self._flags_recipe = 0, 0
        self.exchange_timeouts = 0     # in-launch exchanges that timed out (each: updates skipped until noticed, then the exchange-free schedule)
        self.skipped_updates = 0       # updates the device refused because of them (its own count: word 1 of the region's header), as far as seen
        self._ar_plan, self._ar_issued = [], 0      # data-parallel: the gradient all-reduces of the step in flight, and how many were issued (autotune_dp)
        self._bwd_slot = None          # module API: the slot of the last backward() (its workspace's error word guards the optimizer step)
        self._fused_opt = False        # ... a GrooveSGD / GrooveAdam is bound (training._FusedMixin): the update kernel itself honours the word
        self.xchg_strict = os.environ.get("GT_XCHG_STRICT", "0") == "1"       # raise instead of recovering
        self.B = int(batch_size) if batch_size else None
        if self.B:
            self.slot(self.B)

    cfg_flags = property(lambda self: self._flags_fallback | self._flags_recipe)

    def _apply_flags(self):
        """Push cfg_flags into every configuration the engine keeps; schedules change with them, so captured graphs and the per-slot
        graph decision go (the workspace layout does not depend on them)."""
        f = self.cfg_flags
        for t in self._slots.values():
            if t.cfg.flags != f:
                t.cfg.flags = f
                t.graphs.clear(); t.keep.clear(); t.use_graph = None
        for cfg, _, _ in self._predict_ws.values():
            cfg.flags = f

    # ---- buffers ---------------------------------------------------------------------------------
    def slot(self, B):
        B = int(B)
        s = self._slots.get(B)
        if s is not None and s.layout_epoch != self.lib.cdll.gt_layout_epoch():
            # a layout switch (gt_set_operand_shadows) changed after this slot's workspace was sized: every offset may have moved and the
            # bf16-only regions may no longer fit -- the slot is re-made (its saved activations, graphs and weight packs go with it)
            warnings.warn("StepEngine: the workspace layout changed after the batch-%d slot was sized; re-making the slot" % B)
            del self._slots[B]
            self._predict_ws.clear()
        if B not in self._slots:
            if len(self._slots) >= MAX_SLOTS:                     # evaluation remainders etc.: bounded, like the loss slots
                # least recently USED first; never the engine's own batch size nor the slot the last train step ran on
                # (its captured graphs, kept tensors and weight packs would be re-made every epoch)
                keep = {self.B, self._train_B}
                self._slots.pop(next((k for k in self._slots if k not in keep), next(iter(self._slots))))
            self._slots[B] = _Slot(self, B)
        else:
            self._slots[B] = self._slots.pop(B)                   # (most recently used last)
        return self._slots[B]

    def loss_slot(self, B):
        """Buffers for a stand-alone calculate_loss over B sequences (kept apart from the train slots: the loss of an
        evaluation forward must not touch the activations a pending backward still needs)."""
        B = int(B)
        if B not in self._loss_slots:
            if len(self._loss_slots) >= 4:
                self._loss_slots.pop(next(iter(self._loss_slots)))
            self._loss_slots[B] = _LossSlot(self, B)
        return self._loss_slots[B]

    # convenience views of the default slot (bench / tests)
    x = property(lambda self: self.slot(self.B).x)
    y = property(lambda self: self.slot(self.B).y)
    hvo = property(lambda self: self.slot(self.B).hvo)
    stats = property(lambda self: self.slot(self.B).stats)
    cfg = property(lambda self: self.slot(self.B).cfg)

    def ensure_adam(self):
        if self.m is None:
            self.m = torch.zeros_like(self.params)
            self.v = torch.zeros_like(self.params)

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
            src = t if torch.is_tensor(t) else torch.from_numpy(np.asarray(t, np.float32))
            v[n].copy_(src.to(self.device).view_as(v[n]))

    def state_struct(self):
        return _lib.GtStepState.from_buffer_copy(self.state.cpu().numpy().tobytes())

    def set_state(self, **kw):
        st = self.state_struct()
        for k, val in kw.items():
            setattr(st, k, val)
        self.state.copy_(torch.from_numpy(np.frombuffer(bytes(st), dtype=np.uint8).copy()))   # graphs stay valid

    def set_step_async(self, step):
        """Overwrite only the dropout step counter (bytes 8..11 of the device state), stream-ordered."""
        t = torch.tensor([step & 0x7FFFFFFF], dtype=torch.int32).view(torch.uint8)
        self.state[8:12].copy_(t, non_blocking=True)

    @property
    def stream(self):
        if self.on_host:
            return ctypes.c_void_p(0)
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- the hot path ----------------------------------------------------------------------------
    def _packs_current(self, s):
        """Are the fragment-ordered weight copies in slot s's workspace those of the current parameters?  (written by the last
        fused update on THIS slot, nothing -- the engine or torch -- has written the parameters since; never under graph replay:
        a captured graph would replay the flag blindly)"""
        return (self.fold_pack and not self._capturing and not self.graph_for(s) and s.pack_epoch == self._pepoch
                and self._pver == self.params._version)

    def _note_fused_step(self, s):
        """A fused whole step is about to update the parameters and write the NEXT step's weight copies into slot s's workspace:
        every other slot's copies go stale.  Called for every such step however it is issued -- also before a graph REPLAY, which
        runs none of _enqueue_step's Python (an eager slot would otherwise keep passing GT_STEP_PACKS_CURRENT over weights that
        a graphed slot's replay has updated)."""
        self._pepoch += 1
        s.pack_epoch, self._pver = self._pepoch, self.params._version

    def _enqueue_step(self, s, skip_update):
        flags = skip_update
        if skip_update != 3 and self._packs_current(s):
            flags |= 4                         # GT_STEP_PACKS_CURRENT: no packing launch at the head of the step
        if skip_update == 0:                   # whole step: its update writes the next step's copies into this slot
            self._note_fused_step(s)
        self.lib.call("gt_train_step", ctypes.byref(s.cfg), self.algo, _ptr(self.params), _ptr(self.grads),
                      _ptr(self.m), _ptr(self.v), _ptr(self.pe), _ptr(s.x), _ptr(s.y),
                      ctypes.c_float(self.penalty), _ptr(s.hvo), _ptr(s.stats), _ptr(s.tgt), _ptr(s.ws),
                      _ptr(self.state), int(flags), self.stream)

    def enqueue_update(self, zero_grads=True, slot=None):
        """Fused update over the flat buffers.  zero_grads=True also clears the consumed gradients (fused step path);
        the torch.optim-style front keeps them until zero_grad() like torch does.  slot: the step's slot -- its workspace then
        receives the next step's weight copies (gt_optimizer_step_ws)."""
        self._pepoch += 1
        if slot is None:
            slot = self._bwd_slot          # module API (loss.backward(); opt.step()): the slot backward() ran on
        if slot is not None and slot.B in self._slots and self._slots[slot.B] is slot:
            # with the configuration and its workspace at hand the update honours the exchange region's error word and the data-parallel
            # guard element (gt_optimizer_step_ws: a timed-out exchange never reaches the parameters, Adam's t does not advance)
            if zero_grads and self.fold_pack:
                slot.pack_epoch, self._pver = self._pepoch, self.params._version
            self.lib.call("gt_optimizer_step_ws", ctypes.byref(slot.cfg), self.algo, _ptr(self.params), _ptr(self.grads), _ptr(self.m),
                          _ptr(self.v), _ptr(slot.ws), _ptr(self.state), int(zero_grads), self.stream)
            return
        self.lib.call("gt_optimizer_step", self.algo, _ptr(self.params), _ptr(self.grads), _ptr(self.m), _ptr(self.v),
                      ctypes.c_int64(self.total), _ptr(self.state), int(zero_grads), self.stream)

    def graph_for(self, s):
        """Does slot s replay captured graphs?  (use_graph True / False / "auto": by the step's launch count)"""
        if self.use_graph == "auto":
            if s.use_graph is None:
                n = self.lib.cdll.gt_step_launches(ctypes.byref(s.cfg))
                s.use_graph = not (0 < n <= EAGER_MAX_LAUNCHES)
            return s.use_graph
        return bool(self.use_graph)

    def _replay(self, s, key, fn, aux=False, force=False):
        """Replay the hipGraph captured for `key` on slot s (captured on first use).  aux=True: a graph that continues a
        step another graph began (second half of a bucketed backward) -- it keeps the slot's other graphs.  force=True: a graph
        whatever graph_for says (the one-enqueue data-parallel step)."""
        if not (force or self.graph_for(s)):
            fn()
            return
        if key not in s.graphs:
            if not aux and len(s.graphs) >= MAX_GRAPHS:      # a small LRU instead of dropping everything: alternating recipes (train_step /
                s.graphs.pop(next(iter(s.graphs)))          # train_step_indexed, a second resident dataset) must not re-capture every call
            # one launch outside capture (code-object load), every buffer it touched restored, then capture once
            side = torch.cuda.Stream(self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            snap = (self.params.clone(), self.state.clone(), None if self.m is None else (self.m.clone(), self.v.clone()),
                    self.grads.clone())
            self._capturing = True                # (no GT_STEP_PACKS_CURRENT in a recorded recipe: a replay would pass it blindly)

            def restore():
                torch.cuda.current_stream(self.device).wait_stream(side)
                torch.cuda.synchronize(self.device)
                self.params.copy_(snap[0]); self.state.copy_(snap[1])
                if snap[2] is not None:
                    self.m.copy_(snap[2][0]); self.v.copy_(snap[2][1])
                self.grads.copy_(snap[3])         # zeros for a whole-step graph (gt_train_step's precondition), else what the first half left
            try:
                try:
                    with torch.cuda.stream(side):
                        fn()
                finally:
                    restore()                     # (also when the warm-up raised half-way: a caller that falls back to the eager sequence
                                                  #  must not apply part of this step twice)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    fn()
            finally:
                self._capturing = False
            s.graphs[key] = g
        g = s.graphs.pop(key)
        s.graphs[key] = g                         # (most recently used last)
        g.replay()

    def _dp_whole(self, s, key, whole):
        """The data-parallel step as ONE captured hipGraph (collectives included) -- with a way back: a capture that fails (a collective
        that cannot be recorded on this stack, an allocator call inside the capture) is logged once and the step, like every later one,
        runs as the eager sequence in this same process.  A FIRST capture runs one real warm-up step (collectives included) before it records,
        so it must happen on the same step on every rank: it does at the first step, and again after a recovery from an exchange time-out --
        which is collective in a multi-rank run for exactly this reason (_recover_exchange).  When the capture itself fails on one rank, that
        rank's eager step stands in for the replay the others run: the same collectives in the same order."""
        if not self.dp_graph_failed:
            try:
                self._replay(s, key, whole, force=True)
                self._note_fused_step(s)          # (after the replay: the update inside it wrote the next step's weight copies)
                return
            except Exception as e:                # noqa: BLE001 -- whatever the capture raised
                import warnings
                self.dp_graph_failed = True
                s.graphs.pop(key, None)
                if not self.on_host:
                    torch.cuda.synchronize(self.device)
                warnings.warn("data-parallel step: hipGraph capture failed (%s: %s); falling back to the eager sequence" %
                              (type(e).__name__, e), RuntimeWarning)
        whole()

    def _ar(self, t, async_op=False):
        """A gradient all-reduce of the data-parallel step (counted: autotune_dp keeps a failing rank in step with the others)."""
        import torch.distributed as dist
        self._ar_issued += 1
        return dist.all_reduce(t, async_op=async_op)

    def _dp_recipe(self, overlap, graph):
        """Select a data-parallel recipe: {one all-reduce after backward | two buckets, the first under the rest of backward} x {eager
        sequence | one hipGraph per step}.  The bucketed recipe runs a collective's workgroups beside the backward's launches, whose
        LayerNorm row exchange needs its whole grid resident: this ENGINE's configurations then carry GT_CFG_NO_LN_XCHG (the norm as a row pass
        of its own); captured graphs of another recipe are dropped with the flag change."""
        self.overlap_allreduce, self.dp_graph, self.dp_graph_failed = bool(overlap), bool(graph), False
        two = bool(overlap) and self.B is not None and len(self.lib.grad_buckets(self.slot(self.B).cfg)) == 2
        self._flags_recipe = _lib.CFG_NO_LN_XCHG if two else 0
        self._apply_flags()

    def autotune_dp(self, steps=30, warmup=3, modes=None):
        """world > 1: time `steps` steps of every data-parallel recipe -- {single all-reduce, two overlapped buckets} x {eager sequence,
        one hipGraph per step} -- on the engine's static batch, keep the fastest.  Every rank takes the MAX over ranks of each time (one
        tiny all-reduce per recipe), so all ranks keep the same recipe.  Parameters, optimizer state and step counters are restored:
        the run that follows is the run that would have been.  -> {"modes": {name: ms}, "chosen": name}

        A rank on which a recipe RAISES must not leave the others inside that recipe's collectives: it completes the gradient all-reduces
        of the step it failed in (those of the step's plan not yet issued), then issues the plan of every remaining step of the recipe --
        garbage sums, thrown away with the snapshot restore -- and reports +inf, so the MAX all-reduce that closes the recipe is reached by
        every rank after the same number of collectives and the recipe is simply not chosen (tests/test_exchange_failsafe.py)."""
        import time
        import torch.distributed as dist
        s = self.slot(self.B)
        two = len(self.lib.grad_buckets(s.cfg)) == 2
        cand = [(False, False)] + ([(True, False)] if two else [])
        # the one-hipGraph-per-step recipes (collectives recorded as graph nodes) join the comparison only on request (GT_DP_TUNE_GRAPH=1):
        # they have run with a 1-rank RCCL group only, and a collective that HANGS under capture on real ranks -- unlike one that raises,
        # which _dp_whole survives -- would take the whole run with it.  The eager recipes are the safe set for a first multi-GPU run.
        if not self.on_host and os.environ.get("GT_DP_TUNE_GRAPH", "0") == "1":
            cand += [(False, True)] + ([(True, True)] if two else [])
        if modes is not None:
            cand = [c for c in cand if c in modes]
        name = lambda c: ("buckets" if c[0] else "plain") + ("_graph" if c[1] else "_eager")
        snap = (self.params.clone(), self.state.clone(), None if self.m is None else (self.m.clone(), self.v.clone()))
        sync = (lambda: None) if self.on_host else (lambda: torch.cuda.synchronize(self.device))
        table, errors = {}, {}
        for c in cand:
            self._dp_recipe(*c)                    # (flags and graphs are this candidate's own: like is compared with like)
            t, t0, failed = float("inf"), None, None
            for i in range(warmup + steps):
                if i == warmup and failed is None:
                    sync(); dist.barrier(); t0 = time.perf_counter()
                elif i == warmup:
                    dist.barrier()                 # (a failed rank still joins the barrier the others time from)
                if failed is None:
                    try:
                        self.train_step()
                        continue
                    except Exception as e:         # noqa: BLE001 -- a recipe that cannot run on this stack is simply not chosen
                        failed = e
                        errors[name(c)] = "%s: %s" % (type(e).__name__, e)
                        plan = list(self._ar_plan)
                        done = self._ar_issued % len(plan) if plan else 0
                        self.grads[self.total - 1:].zero_()       # (the guard element: this rank's garbage must not read as an exchange time-out)
                        if not (plan and self._ar_issued and done == 0):       # (all of the step's collectives were out already)
                            for o, n in plan[done:]:
                                dist.all_reduce(self.grads[o:o + n])
                        continue
                for o, n in self._ar_plan:         # a failed rank: the collectives of the steps the others still run
                    dist.all_reduce(self.grads[o:o + n])
            sync()
            if failed is None:
                t = (time.perf_counter() - t0) / steps * 1e3
                if c[1] and self.dp_graph_failed:
                    t = float("inf")               # (it ran, but as the eager sequence)
            tt = torch.tensor([t if t != float("inf") else 1e30], dtype=torch.float64, device=self.device if not self.on_host else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            table[name(c)] = float(tt)
            self.params.copy_(snap[0]); self.state.copy_(snap[1]); self.grads.zero_()
            if snap[2] is not None:
                self.m.copy_(snap[2][0]); self.v.copy_(snap[2][1])
            self._pepoch += 1                      # (the parameters were rewritten: every slot's weight copies are stale)
            for t_ in self._slots.values():        # (graphs of this candidate go: the next one -- and the run -- record their own)
                t_.graphs.clear(); t_.keep.clear()
        best = min(cand, key=lambda c: table[name(c)])
        self._dp_recipe(*best)
        self.dp_tune = {"modes": {k: (None if v >= 1e29 else round(v, 5)) for k, v in table.items()}, "chosen": name(best), "steps": steps}
        if errors:
            self.dp_tune["errors"] = errors
        return self.dp_tune

    def _watched_step(self, s, on_grads):
        """One step with the gradients VISIBLE between backward and update: forward + loss + backward (skip_update = 1), on_grads() --
        the caller reads self.grads / the Parameters' .grad views --, then the fused update.  The fused whole step consumes and
        re-zeroes the gradients inside its last launch, so nothing (wandb.watch's hooks included, ref:train.py:150) ever sees them;
        train_loop takes this path every `watch_log_freq` batches instead."""
        self._enqueue_step(s, 1)
        on_grads()
        self.enqueue_update(slot=s)

    def train_step(self, x=None, y=None, B=None, on_grads=None):
        """One optimisation step on (x, y) (device or host tensors; None = reuse the static buffers).
        Returns the device stats tensor [loss, hit_acc, -, bce, mse_v, mse_o, -, -] without syncing.
        on_grads: called between backward and update of THIS step (single process; see _watched_step)."""
        s = self.slot(x.shape[0] if x is not None else (B or self.B))
        self._train_B = s.B
        s.fwd_id += 1
        if x is not None:
            s.x.copy_(x, non_blocking=True)
        if y is not None:
            s.y.copy_(y, non_blocking=True)
        if self.world_size == 1 and not self.force_dp and on_grads is not None:
            self._watched_step(s, on_grads)
        elif self.world_size == 1 and not self.force_dp:
            if self.graph_for(s):
                self._note_fused_step(s)          # (a replay updates the parameters without running _enqueue_step)
            self._replay(s, ("fused", self.algo, self.penalty), lambda: self._enqueue_step(s, 0))
        else:
            buckets = self.lib.grad_buckets(s.cfg) if self.overlap_allreduce else []
            two = len(buckets) == 2
            if two != bool(self._flags_recipe & _lib.CFG_NO_LN_XCHG):
                # bucket 0's all-reduce runs UNDER the rest of backward: a collective's workgroups beside launches whose workgroups wait for each
                # other (the LayerNorm row exchange needs its whole grid resident) -- the norm runs as a row pass of its own in this recipe.
                # A flag of THIS engine's configurations (gt_config.flags), not a process-wide switch.
                self._flags_recipe = _lib.CFG_NO_LN_XCHG if two else 0
                self._apply_flags()
            # the step's gradient all-reduces, in issue order (autotune_dp completes them on a rank whose step raised)
            self._ar_plan = [tuple(b) for b in buckets] if two else [(0, self.total)]
            self._ar_issued = 0
            guard = self._guard_fn(s)
            if self.dp_graph and on_grads is None and not self.on_host:
                # ONE enqueue per step: forward + backward, the all-reduce(s) and the update captured in one hipGraph -- the collectives are
                # nodes of the graph (RCCL enqueues on its own stream: fork / join edges), nothing returns to Python between the halves
                def whole():
                    if two:
                        (o0, c0), (o1, c1) = buckets
                        self._enqueue_step(s, 2)
                        guard()
                        w0 = self._ar(self.grads[o0:o0 + c0], async_op=True)
                        self._enqueue_step(s, 3)
                        w1 = self._ar(self.grads[o1:o1 + c1], async_op=True)
                        w0.wait(); w1.wait()
                    else:
                        self._enqueue_step(s, 1)
                        guard()
                        self._ar(self.grads)
                    self.enqueue_update(slot=s)
                self._dp_whole(s, ("dp_whole", self.algo, self.penalty, len(buckets)), whole)
                self.poll_exchange(s)
                return s.stats
            if two:
                # bucketed overlap: graph A ends as soon as the upper bucket's gradients are final; its
                # all-reduce (RCCL stream) runs under graph B, the rest of backward.  The collectives stay OUTSIDE the
                # captured graphs.  Sums over ranks; averaged by grad_scale inside the optimizer kernel.
                (o0, c0), (o1, c1) = buckets
                self._replay(s, ("bwd_top", self.algo, self.penalty), lambda: self._enqueue_step(s, 2))
                guard()
                w0 = self._ar(self.grads[o0:o0 + c0], async_op=True)
                if self.graph_for(s) and ("bwd_rest", self.algo, self.penalty) not in s.graphs:
                    w0.wait()                     # first step only: the capture warm-up snapshots and restores the gradient buffer
                self._replay(s, ("bwd_rest", self.algo, self.penalty), lambda: self._enqueue_step(s, 3), aux=True)
                w1 = self._ar(self.grads[o1:o1 + c1], async_op=True)
                w0.wait(); w1.wait()
            else:
                self._replay(s, ("fwdbwd", self.algo, self.penalty), lambda: self._enqueue_step(s, 1))
                guard()
                self._ar(self.grads)                         # RCCL sum over xGMI; averaged by grad_scale
            if on_grads is not None:
                on_grads()                        # (data-parallel: the all-reduced sums; the update averages them by grad_scale)
            self.enqueue_update(slot=s)
        self.poll_exchange(s)
        return s.stats

    def train_step_indexed(self, xs, ys, idx, on_grads=None):
        """One optimisation step on rows `idx` (device int64, len B) of a dataset resident in HBM (xs (N,32,S), ys (N,32,27)):
        the gather into the static step inputs is the first launch of the step's hipGraph (gt_gather_batch), nothing is
        copied on the host side but the B indices into their static buffer.  Data-parallel: as train_step."""
        s = self.slot(idx.shape[0])
        self._train_B = s.B
        s.fwd_id += 1
        s.idx.copy_(idx, non_blocking=True)
        key = (xs.data_ptr(), ys.data_ptr(), xs.shape[0])

        def gather():
            self.lib.call("gt_gather_batch", _ptr(xs), _ptr(ys), _ptr(s.idx), ctypes.c_int64(xs.shape[0]), int(s.B),
                          int(self.dims["embedding_size_src"]), _ptr(s.x), _ptr(s.y), self.stream)

        if on_grads is not None:
            gather()
            return self.train_step(B=s.B, on_grads=on_grads)
        if self.world_size == 1 and not self.force_dp:
            gkey = ("fused_idx", self.algo, self.penalty) + key
            if self.graph_for(s):
                self._note_fused_step(s)
            self._replay(s, gkey, lambda: (gather(), self._enqueue_step(s, 0)))
            if gkey in s.graphs:
                s.keep[gkey] = (xs, ys)           # the captured graph holds their raw pointers: keep the tensors alive with it
                for k in [k for k in s.keep if k not in s.graphs]:
                    del s.keep[k]
            self.poll_exchange(s)
            return s.stats
        gather()
        return self.train_step(B=s.B)

    def voice_metrics(self, hvo_pred, hvo_gt):
        """Per-voice evaluation metrics of predictions against ground truth, both (N,32,27) HVO tensors on the device
        (gt_voice_metrics; ref:evaluator.py:522-525) -> 30-float device tensor (layout: include/groove_hip.h)."""
        p = torch.as_tensor(hvo_pred, dtype=torch.float32).to(self.device).contiguous()
        g = torch.as_tensor(hvo_gt, dtype=torch.float32).to(self.device).contiguous()
        assert p.shape == g.shape and p.shape[-1] == 27
        rows = p.numel() // 27
        out = torch.empty(30, dtype=torch.float32, device=self.device)
        scratch = torch.empty(int(self.lib.cdll.gt_voice_metrics_scratch_floats(ctypes.c_int64(rows))), dtype=torch.float32, device=self.device)
        self.lib.call("gt_voice_metrics", _ptr(p), _ptr(g), ctypes.c_int64(rows), _ptr(out), _ptr(scratch), self.stream)
        return out

    def forward_eval_chunked(self, x, tgt_in=None, chunk=PREDICT_CHUNK):
        """Eval forward of a large set (the reference hands the whole test / validation set to the model at once,
        ref:train.py:195-215) in chunks of `chunk` sequences through ONE reusable slot: the activation workspace of N = 4096
        sequences would be 18.7 GiB at d512 / F2048.  Returns a fresh (N,32,27) tensor; no saved activations (not differentiable)."""
        n = x.shape[0]
        out = torch.empty(n, 32, 27, dtype=torch.float32, device=self.device)
        for i in range(0, n, chunk):
            j = min(n, i + chunk)
            out[i:j].copy_(self.forward(x[i:j], None if tgt_in is None else tgt_in[i:j], train=False))
        return out

    def forward(self, x, tgt_in=None, train=False):
        """Eval/train forward for any batch size -> (B,32,27) [h logits | v | o] buffer of that slot."""
        s = self.slot(x.shape[0])
        s.fwd_id += 1
        s.x.copy_(x)
        if tgt_in is not None:
            s.tgt.copy_(tgt_in)
        def run():
            self.lib.call("gt_forward", ctypes.byref(s.cfg), _ptr(self.params), _ptr(self.pe), _ptr(s.x),
                          None if self.encoder_only else _ptr(s.tgt), _ptr(s.hvo), _ptr(s.ws), _ptr(self.state),
                          int(train), self.stream)
        run()
        # evaluation forwards are consumed on the host right away: a timed-out exchange is noticed here and the forward repeated on the
        # exchange-free schedule (a forward has no side effects).  Rank-LOCAL by construction (an evaluation may run on one rank only): in a
        # multi-rank run the repeat borrows the fallback flags for this call alone, see _local_retry.  Training forwards of the module API are
        # covered by the optimizer step: the update kernel applies nothing while the word is set (backward() starts the asynchronous poll).
        if not train and self._xchg_word(s) is not None and int(self._xchg_word(s)[0].item()) != 0:
            self._local_retry(s.ws, s.cfg, "an evaluation forward", run)
        return s.hvo

    def loss(self, s, y, penalty, want_grad=True):
        """calculate_loss on slot s's current hvo.  -> (stats, d_hvo) device tensors."""
        s.y.copy_(y)
        d_hvo = torch.empty_like(s.hvo) if want_grad else None
        self.lib.call("gt_loss", ctypes.byref(s.cfg), _ptr(s.hvo), _ptr(s.y), ctypes.c_float(penalty), _ptr(s.stats),
                      _ptr(d_hvo), self.stream)
        return s.stats, d_hvo

    def backward(self, s, d_hvo, train, accumulate=True):
        self.lib.call("gt_backward", ctypes.byref(s.cfg), _ptr(self.params), _ptr(self.grads), _ptr(s.x),
                      None if self.encoder_only else _ptr(s.tgt), _ptr(s.hvo), _ptr(d_hvo), _ptr(s.ws), _ptr(self.state),
                      int(train), int(accumulate), self.stream)
        # module API (loss.backward(); opt.step()): gradients computed through a timed-out exchange are garbage.  With the package's own
        # optimizers (GrooveSGD / GrooveAdam: initialize_model's) the UPDATE KERNEL refuses them -- enqueue_update hands it this slot's error
        # word: parameters, moments and Adam's t stay, the skip is counted -- and the asynchronous poll started here notices the word and
        # falls back.  Only for a foreign torch optimizer stepping the Parameters directly are the gradients zeroed on the device (no
        # synchronisation); that costs a pass over the gradient buffer and still lets weight decay / old moments move the weights.
        self._bwd_slot = s
        w = self._xchg_word(s)
        if w is not None and not self.on_host:
            if not self._fused_opt:
                self.grads.masked_fill_((w[:1] != 0).expand(self.total), 0.0)
            self.poll_exchange(s)

    def predict_chunk(self, n):
        """Sequences per gt_predict call for a set of n: as many as fit the workspace budget (greedy decoding is launch-bound --
        2240 launches per call whatever the batch -- so 8 calls of 512 cost 2.4x one call of 4096)."""
        d = self.dims
        budget = PREDICT_WS_BYTES
        if not self.on_host:
            budget = min(budget, torch.cuda.mem_get_info(self.device)[0] // 2)
        chunk = PREDICT_CHUNK
        while chunk < n:
            cfg = _lib.make_config(2 * chunk, d["embedding_size_src"], d["d_model"], d["n_heads"], d["dim_feedforward"],
                                   d["num_encoder_layers"], d["num_decoder_layers"], d["dropout"], d["precision"])
            if 4 * self.lib.workspace_floats(cfg) > budget:
                break
            chunk *= 2
        return chunk

    def predict(self, x, use_thres=True, thres=0.5, chunk=None, pd_seed=None):
        out = self._predict(x, use_thres, thres, chunk, pd_seed)
        # a timed-out exchange inside the call (its own workspaces): noticed when the call ends, the call repeated on the exchange-free schedule
        # (predict has no side effects; rank-local like an evaluation forward)
        bad = [(cfg, ws) for cfg, ws, _ in self._predict_ws.values()
               if self._xchg_word(ws, cfg) is not None and int(self._xchg_word(ws, cfg)[0].item()) != 0]
        if bad:
            for cfg, ws in bad:
                self.lib.call("gt_workspace_init", ctypes.byref(cfg), _ptr(ws), self.stream)

            def again():
                nonlocal out
                out = self._predict(x, use_thres, thres, chunk, pd_seed)
            self._local_retry(None, None, "predict", again)
        self._trim_predict_ws()
        return out

    def _predict(self, x, use_thres=True, thres=0.5, chunk=None, pd_seed=None):
        """model.predict for ANY batch size (ref:evaluator.py:173 passes the whole evaluation set at once):
        returns a (N,32,27) HVO tensor on the device ([h | v | o], one D2H for the evaluator).  The set is walked in chunks
        of `chunk` sequences (default: predict_chunk) over one cached workspace (sized for a chunk, not for N)."""
        x = torch.as_tensor(x, dtype=torch.float32).to(self.device).contiguous()
        n = x.shape[0]
        d = self.dims
        if chunk is None:
            chunk = self.predict_chunk(n)
        out = torch.empty(n, 32, 27, dtype=torch.float32, device=self.device)
        for i in range(0, n, chunk):
            m = min(chunk, n - i)
            if self._predict_epoch != self.lib.cdll.gt_layout_epoch():      # (a layout switch changed: see slot())
                self._predict_ws.clear()
                self._predict_epoch = self.lib.cdll.gt_layout_epoch()
            if m not in self._predict_ws:
                if len(self._predict_ws) >= 2:             # the full chunk + one remainder size at most
                    self._predict_ws.pop(next(k for k in self._predict_ws if k != chunk), None)
                cfg = _lib.make_config(m, d["embedding_size_src"], d["d_model"], d["n_heads"], d["dim_feedforward"],
                                       d["num_encoder_layers"], d["num_decoder_layers"], d["dropout"], d["precision"], self.cfg_flags)
                ws = torch.empty(self.lib.workspace_floats(cfg), dtype=torch.float32, device=self.device)
                self.lib.call("gt_workspace_init", ctypes.byref(cfg), _ptr(ws), self.stream)
                tgt = torch.empty(m, 32, 27, dtype=torch.float32, device=self.device) if not self.encoder_only else None
                self._predict_ws[m] = (cfg, ws, tgt)
            cfg, ws, tgt = self._predict_ws[m]
            if pd_seed is not None:         # use_pd: hits sampled from the probabilities, hashed from the element's index in the WHOLE set
                self.lib.call("gt_predict_pd_at", ctypes.byref(cfg), _ptr(self.params), _ptr(self.pe), _ptr(x[i:i + m]), _ptr(out[i:i + m]),
                              ctypes.c_uint32(int(pd_seed) & 0xFFFFFFFF), ctypes.c_int64(i), _ptr(tgt), _ptr(ws), self.stream)
                continue
            self.lib.call("gt_predict", ctypes.byref(cfg), _ptr(self.params), _ptr(self.pe), _ptr(x[i:i + m]), _ptr(out[i:i + m]),
                          ctypes.c_float(thres), int(use_thres), _ptr(tgt), _ptr(ws), self.stream)
        return out

    def _trim_predict_ws(self):
        for m in [k for k, (_, ws, _) in self._predict_ws.items() if 4 * ws.numel() > PREDICT_WS_KEEP]:
            del self._predict_ws[m]               # (stream-ordered free: the caching allocator keeps the block until the launches ran)

    # ---- in-launch exchanges (QUAD pair exchange, LayerNorm row exchange): what happens when partner workgroups were not co-resident ------
    # The exchanges poll with a bound (csrc/gt_seq.h seq_xchg_get, csrc/gt_gemm64.h g64_collect); a workgroup that gives its partner up raises
    # the error word at the head of the slot's exchange region and goes on with garbage.  Device side, every update that knows the workspace
    # refuses to apply anything while the word is set (parameters, moments and Adam's t untouched, gradients cleared, the skip COUNTED in word 1
    # of the region's header) -- and, data-parallel, while the all-reduced guard element is non-zero, so every rank skips together.
    # Host side, single process: the two words are read (a) asynchronously from the first train step of a slot on and then every
    # XCHG_POLL_EVERY-th (poll_exchange: an 8-byte copy into pinned memory, looked at one poll later -- no synchronisation on the step path),
    # (b) wherever the host synchronises anyway: mean_stats (the logging path), eval forwards, predict.  On a set word the engine zeroes the
    # region, falls back to the exchange-free schedules for the rest of ITS life (gt_config.flags of its own configurations: same numbers, no
    # in-launch exchange -- other engines of the process are not touched), drops its captured graphs and warns -- or raises, with
    # GT_XCHG_STRICT=1.  The steps in between were skipped, never applied; exchange_timeouts / skipped_updates say how many.
    # Multi-rank: a recovery changes the launch sequence and re-captures graphs -- the one-hipGraph recipe's capture runs a real warm-up step
    # with collectives -- so it must happen on the SAME step on every rank: it is decided only from values every rank holds identically (the
    # all-reduced guard element, read on the same step count and waited for; the flag inside mean_stats' / check_exchange's all-reduce).
    # Rank-local observations (an evaluation forward or predict on one rank) repeat their own call on borrowed flags and change nothing else.
    FALLBACK_FLAGS = _lib.CFG_NO_QUAD | _lib.CFG_NO_LN_XCHG

    def _xchg_word(self, s_or_ws, cfg=None):
        """2-element int32 view [error word, skipped updates] of a slot (or of a (cfg, workspace) pair), None when the shape has no
        exchange region."""
        if cfg is None:
            s = s_or_ws
            if s.xchg_off is None:
                try:
                    s.xchg_off = self.lib.ws_find(s.cfg, "xchg_err")[0]
                except Exception:
                    s.xchg_off = -1
            return s.ws[s.xchg_off:s.xchg_off + 2].view(torch.int32) if s.xchg_off >= 0 else None
        try:
            off = self.lib.ws_find(cfg, "xchg_err")[0]
        except Exception:
            return None
        return s_or_ws[off:off + 2].view(torch.int32) if off >= 0 else None

    def _guard_fn(self, s):
        """Data-parallel: before the gradient all-reduce every rank writes its error flag into the guard element (the last float of the
        gradient buffer: padding behind the 27-float output bias); the update kernel of EVERY rank skips when the sum is non-zero."""
        if self._xchg_word(s) is None:
            return lambda: None
        return lambda: self.lib.call("gt_dp_guard", ctypes.byref(s.cfg), _ptr(self.grads), _ptr(s.ws), self.stream)      # (one 1-thread launch)

    def _timeout_message(self, what):
        return ("an in-launch exchange (four-workgroups-per-sequence pair exchange / LayerNorm row exchange) timed out during %s (partner "
                "workgroups not co-resident: another stream / process holds CUs?)" % what)

    def _local_retry(self, ws, cfg, what, again):
        """A rank-local time-out in a call without side effects (evaluation forward, predict): zero the region, repeat the call on the
        exchange-free schedule.  Single process: that schedule stays (a full recovery).  Multi-rank: the flags are borrowed for the repeat
        alone -- the training schedule, its graphs and the other ranks are not touched."""
        self.exchange_timeouts += 1
        if self.xchg_strict:
            raise RuntimeError(self._timeout_message(what) + "; GT_XCHG_STRICT=1")
        if self.world_size == 1:
            self.exchange_timeouts -= 1            # (counted by _recover_exchange)
            self._recover_exchange(ws, cfg, what)
            again()
            return
        warnings.warn(self._timeout_message(what) + " -- repeated on the exchange-free schedule (this call only)", RuntimeWarning)
        if ws is not None:
            self.lib.call("gt_workspace_init", ctypes.byref(cfg), _ptr(ws), self.stream)
        keep = (self._flags_fallback, [(t, dict(t.graphs), dict(t.keep), t.use_graph) for t in self._slots.values()])
        self._flags_fallback = self.FALLBACK_FLAGS
        try:
            self._apply_flags()
            again()
        finally:
            self._flags_fallback = keep[0]
            self._apply_flags()
            for t, g, k, u in keep[1]:             # (the training graphs were recorded under the flags now back in force)
                t.graphs, t.keep, t.use_graph = g, k, u

    def _recover_exchange(self, ws, cfg, what):
        """Fall back to the exchange-free schedules for the rest of this engine's life.  Multi-rank: call it on EVERY rank on the same step."""
        self.exchange_timeouts += 1
        if self.xchg_strict:
            raise RuntimeError(self._timeout_message(what) + "; GT_XCHG_STRICT=1")
        for t in self._slots.values():             # the device's own count, before the regions are zeroed
            w = self._xchg_word(t)
            if w is not None:
                self.skipped_updates += int(w[1].item()) - t.xchg_skipped_seen
                t.xchg_skipped_seen = 0
        warnings.warn(self._timeout_message(what) + " -- the affected updates were skipped on the device (%d so far); falling back to the "
                      "exchange-free schedules for the rest of this engine's life" % self.skipped_updates, RuntimeWarning)
        self._flags_fallback = self.FALLBACK_FLAGS
        self._apply_flags()                        # (drops the captured graphs: they hold the exchanging launches)
        if ws is not None:
            self.lib.call("gt_workspace_init", ctypes.byref(cfg), _ptr(ws), self.stream)      # (zeroes the region: stale granules, the words)
        for t in self._slots.values():
            t.graphs.clear(); t.keep.clear(); t.use_graph = None
            t.xchg_host = None
            if t.ws is not ws and self._xchg_word(t) is not None:
                self.lib.call("gt_workspace_init", ctypes.byref(t.cfg), _ptr(t.ws), self.stream)
        self.grads[self.total - 1:].zero_()

    def poll_exchange(self, s):
        """Step path, no host-side wait in a single process: the slot's first call and then every XCHG_POLL_EVERY-th look at the PREVIOUS
        asynchronous copy of [error word, skipped updates] (complete by now, or left for the next poll) and start the next one.
        Multi-rank: the copy is of the ALL-REDUCED guard element (identical on every rank) and the look WAITS for it, so every rank decides
        on the same step from the same value (the copy is XCHG_POLL_EVERY steps old: the wait only bounds how far the host runs ahead)."""
        s.xchg_count += 1
        if s.xchg_count != 1 and s.xchg_count % XCHG_POLL_EVERY:
            return
        w = self._xchg_word(s)
        if w is None:
            return
        multi = self.world_size > 1
        if self.on_host:                            # (host emulator: no streams -- read in place; multi-rank tests over gloo)
            flag = float(self.grads[self.total - 1]) != 0.0 if multi else int(w[0]) != 0
            if flag:
                self._recover_exchange(s.ws, s.cfg, "a train step")
            return
        src = self.grads[self.total - 1:].view(torch.int32) if multi else w
        if s.xchg_host is None:
            s.xchg_host = torch.zeros(src.numel(), dtype=torch.int32).pin_memory()
            s.xchg_event = torch.cuda.Event()
        else:
            if multi:
                s.xchg_event.synchronize()
            elif not s.xchg_event.query():
                return                              # (the last copy is still in flight: look again next time)
            if int(s.xchg_host[0]) != 0:
                s.xchg_host.zero_()
                self._recover_exchange(s.ws, s.cfg, "a train step")
                return
            if not multi:
                seen = int(s.xchg_host[1])
                self.skipped_updates += seen - s.xchg_skipped_seen
                s.xchg_skipped_seen = seen
        s.xchg_host.copy_(src, non_blocking=True)
        s.xchg_event.record(torch.cuda.current_stream(self.device))

    def check_exchange(self, s, what="a train step"):
        """Synchronising read of slot s's error word (call where the host waits for the device anyway).  -> True when a time-out was found
        (and recovered from: the numbers of the launches since the last check are garbage, their updates were skipped).  Multi-rank: a
        COLLECTIVE -- every rank must call it at the same point (the flag is all-reduced, every rank recovers or none does)."""
        w = self._xchg_word(s)
        if self.world_size > 1:
            import torch.distributed as dist
            flag = torch.zeros(1, dtype=torch.float32, device=s.stats.device) if w is None else (w[:1] != 0).to(torch.float32)
            dist.all_reduce(flag)
            if float(flag) == 0.0:
                return False
            self._recover_exchange(s.ws, s.cfg, what + " (some rank)")
            return True
        if w is None or int(w[0].item()) == 0:
            return False
        self._recover_exchange(s.ws, s.cfg, what)
        return True

    def exchange_report(self, s=None):
        """{"exchange_timeouts", "skipped_updates"} up to now (synchronising: adds what the device counted since the last look)."""
        n = self.skipped_updates
        for t in ([s] if s is not None else list(self._slots.values())):
            w = self._xchg_word(t)
            if w is not None:
                n += int(w[1].item()) - t.xchg_skipped_seen
        return {"exchange_timeouts": self.exchange_timeouts, "skipped_updates": n}

    def mean_stats(self, s):
        """The slot's 8-float stats averaged over the data-parallel ranks (one tiny all-reduce; every rank must call it).
        Single process: the stats tensor itself.  Also the synchronising check of the exchanges: stats of a step whose exchange timed
        out come back as NaN (data-parallel: on every rank -- the flag travels with the stats all-reduce, so all ranks fall back together)."""
        if self.world_size == 1 or not self.reduce_stats:
            if self.world_size == 1 and self.check_exchange(s):
                return torch.full_like(s.stats, float("nan"))
            return s.stats
        import torch.distributed as dist
        w = self._xchg_word(s)
        flag = torch.zeros(1, dtype=torch.float32, device=s.stats.device) if w is None else (w[:1] != 0).to(torch.float32)
        t = torch.cat([s.stats, flag])
        dist.all_reduce(t)
        if float(t[8]) != 0.0:
            self._recover_exchange(s.ws, s.cfg, "a train step (some rank)")     # (ranks whose own word is clean fall back too: one schedule everywhere)
            return torch.full_like(s.stats, float("nan"))
        return t[:8] / self.world_size

    def profile(self, steps):
        """Eager (no graph) pass of `steps` train steps with HIP events around every launch.
        -> {kernel class: (launches, total_ms, total_flops, total_bytes)}.  Measurement aid for bench.py."""
        if self.on_host:
            return {}
        s = self.slot(self.B)
        snap = (self.params.clone(), self.state.clone(), None if self.m is None else (self.m.clone(), self.v.clone()))
        torch.cuda.synchronize(self.device)
        self.lib.cdll.gt_profile_enable(1)
        try:
            for _ in range(steps):
                self._enqueue_step(s, 0)
            buf = ctypes.create_string_buffer(1 << 16)
            self.lib.cdll.gt_profile_report(buf, len(buf), 256)
        finally:
            self.lib.cdll.gt_profile_enable(0)
        self.params.copy_(snap[0]); self.state.copy_(snap[1]); self.grads.zero_()
        if snap[2] is not None:
            self.m.copy_(snap[2][0]); self.v.copy_(snap[2][1])
        out = {}
        for line in buf.value.decode().splitlines():
            lab, cnt, ms, fl, by = line.split()
            out[lab] = (int(cnt), float(ms), float(fl), float(by))
        return out
