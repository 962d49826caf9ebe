"""initialize_model / calculate_loss / train_loop: the three names the reference imports from its
un-vendored submodule (ref:train.py:12), with the call signatures the reference uses
(ref:train.py:149,195-215; ref:tutorial.py:62-68,98-105), running on the HIP hot path.

Checkpoints keep the reference's layout: ``{epoch, model_state_dict, optimizer_state_dict, loss}`` in a file
``transformer_run_{run}_Epoch_{epoch}.Model`` (ckpt keys; pattern ref:tutorial.py:65), and
``params["load_model"]`` = ``{location: local|wandb, dir, file_pattern, epoch, run}`` resumes from one.
"""
import glob
import math
import os
import re

import torch

from .model import GrooveTransformer, GrooveTransformerEncoder, _GrooveBase, engine_of

try:                                    # Weights & Biases is optional here (ref:train.py:106-113,150,252)
    import wandb
except Exception:                       # pragma: no cover - not installed in the build container
    wandb = None


def _wandb_active():
    return wandb is not None and getattr(wandb, "run", None) is not None


# ------------------------------------------------------------------------------------------------ optimizers
class _FusedMixin:
    """torch.optim-compatible front of the fused flat update (one kernel over all tensors).  Keeps torch's
    state_dict format: SGD -> per-param {'momentum_buffer': None} (ckpt); Adam -> step/exp_avg/exp_avg_sq."""

    def _bind(self, engine, algo):
        self.engine = engine
        self._algo = algo
        engine.algo = algo
        engine._fused_opt = True           # (the update kernel honours the exchange error word: engine.backward need not zero gradients)
        self._lr_on_device = None

    def zero_grad(self, set_to_none=False):
        self.engine.grads.zero_()

    def _push_lr(self):
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_on_device:
            self.engine.set_state(lr=lr)
            self._lr_on_device = lr

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self._push_lr()
        self.engine.algo = self._algo
        self.engine.enqueue_update(zero_grads=False)
        return loss


class GrooveSGD(_FusedMixin, torch.optim.SGD):
    """torch.optim.SGD(lr, momentum=0) semantics (ckpt: optimizer param_groups) on the fused kernel."""

    def __init__(self, params, lr, engine):
        torch.optim.SGD.__init__(self, params, lr=lr)
        self._bind(engine, 0)
        for p in self.param_groups[0]["params"]:
            self.state[p]["momentum_buffer"] = None


class GrooveAdam(_FusedMixin, torch.optim.Adam):
    """torch.optim.Adam(lr) defaults; exp_avg / exp_avg_sq are views into the engine's flat moment buffers."""

    def __init__(self, params, lr, engine):
        torch.optim.Adam.__init__(self, params, lr=lr)
        self._bind(engine, 1)
        engine.ensure_adam()
        m, v = engine.views(engine.m), engine.views(engine.v)
        by_ptr = {t.data_ptr(): n for n, t in engine.views().items()}
        for p in self.param_groups[0]["params"]:
            n = by_ptr[p.data_ptr()]
            self.state[p] = {"step": torch.tensor(0.0), "exp_avg": m[n], "exp_avg_sq": v[n]}

    @torch.no_grad()
    def step(self, closure=None):
        out = _FusedMixin.step(self, closure)
        for st in self.state.values():
            st["step"] += 1
        return out

    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        self.param_groups[0]["lr"] = groups[0]["lr"]
        params = self.param_groups[0]["params"]
        steps = 0
        for i, p in enumerate(params):
            st = sd["state"].get(i)
            if st:
                self.state[p]["exp_avg"].copy_(st["exp_avg"])
                self.state[p]["exp_avg_sq"].copy_(st["exp_avg_sq"])
                self.state[p]["step"] = torch.as_tensor(float(st["step"]))
                steps = int(st["step"])
        self.engine.set_state(opt_step=steps)


# ------------------------------------------------------------------------------------------------ checkpoints
FILE_PATTERN = "transformer_run_{}_Epoch_{}.Model"


def save_checkpoint(path, epoch, model, optimizer, loss):
    """Exactly the reference's four keys (ckpt).  The position of this replica's dropout stream rides in
    ``optimizer_state_dict["param_groups"][0]["dropout_step"]`` (torch optimizers keep unknown group keys), so a resumed
    run continues with fresh masks instead of replaying the ones of step 0."""
    osd = optimizer.state_dict()
    eng = getattr(model, "engine", None)
    if eng is not None and osd.get("param_groups"):
        osd["param_groups"][0]["dropout_step"] = max(int(eng.state_struct().step), int(getattr(model, "_train_forwards", 0)))
    torch.save({"epoch": epoch, "model_state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
                "optimizer_state_dict": osd, "loss": float(loss)}, path)
    return path


def find_checkpoint(load_model):
    """load_model = {location, dir, file_pattern, [epoch], [run]} (ref:tutorial.py:62-66,98-104).
    Without 'epoch' the last stored epoch is used (ref:tutorial.py:36)."""
    loc = load_model.get("location", "local")
    if loc == "wandb":
        if wandb is None:
            raise RuntimeError("load_model.location == 'wandb' needs the wandb package")
        ep = load_model["epoch"]
        f = wandb.restore(load_model["file_pattern"].format(load_model["run"], ep), run_path=load_model["dir"])
        return f.name
    d, pat = load_model["dir"], load_model["file_pattern"]
    if load_model.get("epoch") is not None:
        cands = glob.glob(os.path.join(d, pat.format(load_model.get("run", "*"), load_model["epoch"])))
        if not cands:
            raise FileNotFoundError("no checkpoint for epoch %s in %s" % (load_model["epoch"], d))
        return sorted(cands)[-1]
    rx = re.compile(re.escape(pat).replace(r"\{\}", "(.+)", 1).replace(r"\{\}", r"(\d+)", 1) + "$")
    best = None
    for f in glob.glob(os.path.join(d, pat.format(load_model.get("run", "*"), "*"))):
        m = rx.match(os.path.basename(f))
        if m and (best is None or int(m.group(2)) > best[0]):
            best = (int(m.group(2)), f)
    if best is None:
        raise FileNotFoundError("no checkpoint matching %s in %s" % (pat, d))
    return best[1]


# ------------------------------------------------------------------------------------------------ initialize_model
def initialize_model(params):
    """params = {"model": {...}, "training": {...}, "load_model": None | {...}} (ref:train.py:115-143).
    -> (model, optimizer, initial_epoch)."""
    mp, tp = params["model"], params["training"]
    common = dict(d_model=mp["d_model"], nhead=mp["n_heads"], dim_feedforward=mp["dim_feedforward"],
                  dropout=mp["dropout"], embedding_size_src=mp["embedding_size_src"],
                  embedding_size_tgt=mp["embedding_size_tgt"], max_len=mp["max_len"], device=mp.get("device", "cuda"),
                  seed=int(params.get("seed", tp.get("seed", 0)) or 0),      # dropout stream (train.py --seed); rank mixed in by the model
                  precision=mp.get("precision", "fp32"))                     # "bf16": GEMM operands in bf16 (BASELINE configs[4]); "autocast": ... and bf16 storage of the Linear outputs (precision 2)
    if mp["encoder_only"]:
        model = GrooveTransformerEncoder(num_encoder_layers=mp["num_encoder_layers"], **common)
    else:
        model = GrooveTransformer(num_encoder_layers=mp["num_encoder_layers"],
                                  num_decoder_layers=mp["num_decoder_layers"], **common)
    lr = tp["learning_rate"]
    algo = str(mp.get("optimizer", "sgd")).lower()
    if algo == "adam":
        optimizer = GrooveAdam(model.parameters(), lr, model.engine)
    elif algo == "sgd":
        optimizer = GrooveSGD(model.parameters(), lr, model.engine)
    else:
        raise ValueError("optimizer_algorithm must be 'sgd' or 'adam' (ref:train.py:40-42), got %r" % algo)
    _bind_engine(model)
    model.engine.set_state(lr=float(lr))
    model.engine.penalty = float(tp.get("hit_loss_penalty", 1.0))
    initial_epoch = 0
    if params.get("load_model"):
        ck = torch.load(find_checkpoint(params["load_model"]), map_location="cpu", weights_only=True)
        model.load_state_dict(ck["model_state_dict"], strict=True)
        optimizer.load_state_dict(ck["optimizer_state_dict"])
        optimizer._lr_on_device = None
        initial_epoch = int(ck["epoch"]) + 1          # resume after the stored epoch (payload, not file name: SURVEY 5)
        # continue the dropout stream where the stored run stopped (a reference-written checkpoint has no position: start
        # far from the masks of the first epochs)
        groups = ck["optimizer_state_dict"].get("param_groups") or [{}]
        step = int(groups[0].get("dropout_step", initial_epoch << 20)) & 0x7FFFFFFF
        model.engine.set_state(step=step)
        model._train_forwards = step
    return model, optimizer, initial_epoch


# ------------------------------------------------------------------------------------------------ calculate_loss
class _LossFn(torch.autograd.Function):
    """gt_loss: BCE(hits)*pen + MSE(vel)*pen + MSE(off)*pen, voices summed, (B,T) averaged; grad = gt_loss's d_hvo."""

    @staticmethod
    def forward(ctx, hvo, y, penalty, engine):
        s = engine.loss_slot(hvo.shape[0])
        s.hvo.copy_(hvo)
        stats, d_hvo = engine.loss(s, y, penalty, want_grad=True)
        ctx.save_for_backward(d_hvo)
        ctx.stats = stats.clone()
        return ctx.stats[0].clone()

    @staticmethod
    def backward(ctx, g):
        return ctx.saved_tensors[0] * g, None, None, None


def calculate_loss(prediction, y, bce_fn, mse_fn, hit_loss_penalty):
    """loss_fn of train_loop (ref:train.py:201-203,213).  bce_fn / mse_fn must be the reference's
    BCEWithLogitsLoss / MSELoss(reduction='none') (ref:train.py:176-179): the fused kernel implements exactly
    those.  Returns (loss tensor, hit_accuracy, hit_perplexity, bce_hits, mse_velocities, mse_offsets)."""
    for fn, kind in ((bce_fn, torch.nn.BCEWithLogitsLoss), (mse_fn, torch.nn.MSELoss)):
        if fn is not None and (not isinstance(fn, kind) or fn.reduction != "none"):
            raise ValueError("calculate_loss expects %s(reduction='none') as in ref:train.py:176-179" % kind.__name__)
    h, v, o = prediction
    if not h.is_cuda:
        raise RuntimeError("calculate_loss runs on the GPU path only (predictions must be CUDA tensors)")
    # the engine that produced these predictions (model._run registers its output); predictions built elsewhere fall back to
    # the engine of the last initialised model
    engine = engine_of(h) or calculate_loss._engine
    if engine is None:
        raise RuntimeError("no model initialised: call initialize_model() (or bind calculate_loss._engine) first")
    hvo = torch.cat([h, v, o], dim=-1).contiguous()
    y = y.to(hvo.device, torch.float32)
    loss = _LossFn.apply(hvo, y, float(hit_loss_penalty), engine)
    st = engine.loss_slot(hvo.shape[0]).stats.tolist()       # ONE D2H of the stats struct (no 5-6 .item() syncs)
    return loss, st[1], math.exp(st[3]), st[3], st[4], st[5]


calculate_loss._engine = None


def _bind_engine(model):
    if isinstance(model, _GrooveBase):
        calculate_loss._engine = model.engine


def save_schedule(total_epochs, initial_epochs_lim=10, initial_step=1, secondary_step_partial=10, secondary_step_all=20,
                  only_final=False):
    """Which epochs store a checkpoint / full evaluation: every epoch for the first `initial_epochs_lim`, then every
    10 (partial) / 20 (all), plus the last one -- the schedule ref:train.py:182-190 builds with ref:utils.py:230-264."""
    if only_final:
        return {total_epochs - 1}, set()
    part = set(range(0, min(initial_epochs_lim, total_epochs), initial_step))
    full = set(part)
    if initial_epochs_lim < total_epochs:
        part |= set(range(initial_epochs_lim, total_epochs, secondary_step_partial)) | {total_epochs - 1}
        full |= set(range(initial_epochs_lim, total_epochs, secondary_step_all)) | {total_epochs - 1}
    return part, full


# ------------------------------------------------------------------------------------------------ train_loop
def shift_right(y):
    """decoder teacher-forcing input: y delayed by one step, first row zeros."""
    return torch.cat([torch.zeros_like(y[:, :1]), y[:, :-1]], dim=1)


def _metrics_dict(prefix, st):
    return {prefix + "loss": st[0], prefix + "hit_accuracy": st[1], prefix + "hit_perplexity": math.exp(st[3]),
            prefix + "bce_h": st[3], prefix + "mse_v": st[4], prefix + "mse_o": st[5]}


def train_loop(dataloader, groove_transformer, encoder_only, opt, epoch, loss_fn, bce_fn, mse_fn, device,
               test_inputs=None, test_gt=None, validation_inputs=None, validation_gt=None, hit_loss_penalty=1,
               save=False, save_dir=None, run_id=None, log_every=50, on_log=None):
    """One epoch (ref:train.py:195-215).  For every (x, y, idx) batch: forward, calculate_loss, backward, update.
    When model, loss_fn and optimizer are this package's, the whole batch body is ONE captured hipGraph replay
    (gt_train_step) and metrics leave the GPU as one 8-float copy every `log_every` batches; any other
    combination takes the generic autograd path.  Returns the metrics of the last logged batch."""
    model = groove_transformer
    _bind_engine(model)
    model.train()
    eng = getattr(model, "engine", None)
    fast = (isinstance(model, _GrooveBase) and loss_fn is calculate_loss and isinstance(opt, _FusedMixin)
            and opt.engine is eng)
    world = 1
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        world = torch.distributed.get_world_size()
    fused_opt = isinstance(opt, _FusedMixin) and opt.engine is eng
    if eng is not None and eng.world_size != world:
        eng.world_size = world
    if fused_opt:                              # the fused update averages by grad_scale on the device: keep it = 1/world on BOTH paths
        eng.set_state(grad_scale=1.0 / world)
    if fast:
        eng.grads.zero_()                      # gt_train_step precondition; every fused update re-zeroes them
        eng.penalty = float(hit_loss_penalty)
        eng.algo = opt._algo
        opt._push_lr()
    last, stats = None, None
    n_batches = 0
    # a dataset resident in HBM hands over INDICES: the gather is the first launch of the step's graph
    indexed = fast and hasattr(dataloader, "index_batches") and getattr(dataloader, "x", None) is not None \
        and dataloader.x.device == eng.device
    batches = ((None, None, i) for i in dataloader.index_batches()) if indexed else dataloader
    # wandb.watch(model, log_freq=1000) (ref:train.py:150) hooks the Parameters' gradients; the fused step never materialises a .grad
    # autograd could hook (it consumes and re-zeroes the flat gradient buffer inside its last launch).  Equivalent: every
    # model.watch_log_freq batches (0 = never) the step runs split -- backward, gradients visible, update -- and their histograms
    # are logged under the names wandb.watch uses ("gradients/<parameter>").
    watch = int(getattr(model, "watch_log_freq", 0) or 0) if fast else 0

    def watch_cb():
        rec = {}
        for n, g in eng.views(eng.grads).items():
            a = g.detach().float().cpu().numpy()
            rec["gradients/" + n] = wandb.Histogram(a) if (_wandb_active() and hasattr(wandb, "Histogram")) else a
        model.last_watch = rec
        if _wandb_active():
            wandb.log(rec, commit=False)

    for batch, (X, y, _idx) in enumerate(batches):
        n_batches += 1
        model._watch_step = getattr(model, "_watch_step", 0) + 1
        on_grads = watch_cb if (watch and model._watch_step % watch == 0) else None
        if indexed:
            stats = eng.train_step_indexed(dataloader.x, dataloader.y, _idx, on_grads=on_grads)
            X = _idx                           # (only its length is used below)
            if (batch + 1) % log_every == 0:
                last = _metrics_dict("train/", eng.mean_stats(eng.slot(X.shape[0])).tolist())
            if last is not None and (batch + 1) % log_every == 0:
                rec = dict(last, epoch=epoch, batch=batch)
                if _wandb_active():
                    wandb.log(rec, commit=True)
                if on_log:
                    on_log(rec)
            continue
        X = X.to(device, torch.float32, non_blocking=True)
        y = y.to(device, torch.float32, non_blocking=True)
        if fast:
            stats = eng.train_step(X, y, on_grads=on_grads)
            if (batch + 1) % log_every == 0:
                last = _metrics_dict("train/", eng.mean_stats(eng.slot(X.shape[0])).tolist())
        else:
            opt.zero_grad()
            pred = model(X) if encoder_only else model(X, shift_right(y))
            out = loss_fn(pred, y, bce_fn, mse_fn, hit_loss_penalty)
            out[0].backward()
            if world > 1:
                if eng is not None:            # every .grad is a view of ONE flat buffer: one collective, not one per tensor
                    torch.distributed.all_reduce(eng.grads)
                    if not fused_opt:          # a foreign optimizer knows nothing of grad_scale: average here
                        eng.grads /= world
                else:
                    for p in model.parameters():
                        torch.distributed.all_reduce(p.grad)
                        p.grad /= world
            opt.step()
            last = {"train/loss": float(out[0]), "train/hit_accuracy": out[1], "train/hit_perplexity": out[2],
                    "train/bce_h": out[3], "train/mse_v": out[4], "train/mse_o": out[5]}
        if last is not None and ((batch + 1) % log_every == 0 or not fast):      # reference logs per batch; fast path every log_every
            rec = dict(last, epoch=epoch, batch=batch)
            if _wandb_active():
                wandb.log(rec, commit=True)
            if on_log:
                on_log(rec)
    if fast and stats is not None:
        last = _metrics_dict("train/", eng.mean_stats(eng.slot(X.shape[0])).tolist())
    if isinstance(opt, GrooveAdam) and fast:
        for st in opt.state.values():
            st["step"] += n_batches
    if save:
        d = save_dir or (wandb.run.dir if _wandb_active() else ".")
        rid = run_id or (wandb.run.id if _wandb_active() else "local")
        save_checkpoint(os.path.join(d, FILE_PATTERN.format(rid, epoch)), epoch, model, opt,
                        last["train/loss"] if last else float("nan"))
    for name, xin, gt in (("test/", test_inputs, test_gt), ("validation/", validation_inputs, validation_gt)):
        if xin is None or gt is None:
            continue
        model.eval()
        with torch.no_grad():
            xin, gt = xin.to(device, torch.float32), gt.to(device, torch.float32)
            pred = model(xin) if encoder_only else model(xin, shift_right(gt))
            out = loss_fn(pred, gt, bce_fn, mse_fn, hit_loss_penalty)
        rec = {name + "loss": float(out[0]), name + "hit_accuracy": out[1], name + "hit_perplexity": out[2],
               name + "bce_h": out[3], name + "mse_v": out[4], name + "mse_o": out[5], "epoch": epoch}
        if _wandb_active():
            wandb.log(rec, commit=False)
        if on_log:
            on_log(rec)
        model.train()
    return last
