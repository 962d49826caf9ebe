"""The four-workgroups-per-sequence (QUAD) pair exchange must fail SAFE (ADVICE r04 / VERDICT r04 #4): a timed-out exchange raises an
error word; from then on the fused update applies nothing, the host notices (asynchronous poll on the step path, synchronising check on
the logging / evaluation paths), zeroes the region and falls back to two workgroups per sequence.  Here on host memory with the
emulator build of the kernels (the GPU suite drives a real time-out: tests/test_hip_api.py)."""
import os
import socket
import sys
import warnings

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from harness import run_ranks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIMS = dict(d_model=128, n_heads=4, dim_feedforward=32, num_encoder_layers=1, num_decoder_layers=0, dropout=0.0, embedding_size_src=16)


DIMS2 = dict(d_model=48, n_heads=4, dim_feedforward=24, num_encoder_layers=2, num_decoder_layers=0, dropout=0.0, embedding_size_src=16)   # (one kernel per op: two gradient buckets)


def _engine(B, world=1, seed=3, dims=None, **kw):
    from harness import emu_lib
    from transformergrooveinfilling_amd import layout
    from transformergrooveinfilling_amd.engine import StepEngine
    lib = emu_lib()
    for f in ("gt_set_seq_quad", "gt_set_seq_split", "gt_set_seq_ride"):      # (process-global switches other tests of this run may have left)
        getattr(lib.cdll, f)(-1)
    lib.cdll.gt_set_seq(1)
    eng = StepEngine(batch_size=B, optimizer=kw.pop("optimizer", "sgd"), learning_rate=0.05, hit_loss_penalty=0.47, seed=seed, device="cpu",
                     world_size=world, lib=lib, **(dims or DIMS))
    eng.load_named(layout.init_params(dims or DIMS, seed=5))
    return eng


def _raise_word(eng, s):
    w = eng._xchg_word(s)
    assert w is not None                       # d_model 128: the workspace has the exchange region
    w[0] = 1                                   # (word 0: the error word; word 1 is the device's count of skipped updates)


@pytest.mark.parametrize("optimizer", ["sgd", "adam"])
def test_timed_out_exchange_skips_the_update_and_recovers(optimizer):
    from transformergrooveinfilling_amd import layout
    eng = _engine(2, optimizer=optimizer)
    try:
        x, y = layout.synthetic_batch(2, 16, seed=9)
        x, y = torch.from_numpy(x), torch.from_numpy(y)
        eng.train_step(x, y)
        assert not eng.check_exchange(eng.slot(2))
        before = eng.params.clone()
        m0 = None if eng.m is None else eng.m.clone()
        _raise_word(eng, eng.slot(2))
        eng.train_step(x, y)                   # forward / backward ran, the update must not have been applied
        assert torch.equal(eng.params, before)
        assert float(eng.grads.abs().max()) == 0.0          # (consumed gradients cleared all the same)
        if m0 is not None:
            assert torch.equal(eng.m, m0)
        eng.train_step(x, y)                   # the word is sticky: still nothing
        assert torch.equal(eng.params, before)
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            st = eng.mean_stats(eng.slot(2))   # the logging path: finds it, recovers, reports NaN for the garbage step
        assert torch.isnan(st).all() and eng.exchange_timeouts == 1
        assert any("pair exchange" in str(w.message) for w in rec)
        assert int(eng._xchg_word(eng.slot(2))[0].item()) == 0
        assert eng.skipped_updates == 2 and eng.exchange_report() == {"exchange_timeouts": 1, "skipped_updates": 2}      # the device's own count
        st_ = eng.state_struct()               # the dropout stream moved on with every consumed batch, Adam's t only with the applied update
        assert st_.step == 3 and st_.opt_step == 1
        assert eng.cfg_flags == eng.FALLBACK_FLAGS and eng.slot(2).cfg.flags == eng.FALLBACK_FLAGS     # this ENGINE's configurations, not the process
        eng.train_step(x, y)                   # two workgroups per sequence from here on: trains again
        assert not torch.equal(eng.params, before)
        ref = _engine(2, optimizer=optimizer)  # (a fresh engine, QUAD schedule: the same numbers to fp32 rounding)
        ref.params.copy_(before)
        if optimizer == "sgd":             # (Adam would need step 1's moments in the fresh engine: the skip itself is checked above)
            ref.set_state(step=eng.state_struct().step - 1, opt_step=eng.state_struct().opt_step - 1)
            ref.train_step(x, y)
            assert (ref.params - eng.params).abs().max() < 1e-6
        assert torch.isfinite(eng.mean_stats(eng.slot(2))).all()
    finally:
        eng.lib.cdll.gt_set_seq_quad(-1)


def test_recovery_is_per_engine_not_per_process():
    """VERDICT r05 #5b: one engine's fall-back must not change the schedule of another engine of the process (train.py's evaluation
    engines, bench.py's second StepEngine): the switch travels in gt_config.flags of the engine's own configurations."""
    import ctypes
    from transformergrooveinfilling_amd import layout
    a, b = _engine(2), _engine(2)
    x, y = layout.synthetic_batch(2, 16, seed=9)
    x, y = torch.from_numpy(x), torch.from_numpy(y)
    a.train_step(x, y); b.train_step(x, y)
    quad_launches = b.lib.cdll.gt_step_launches(ctypes.byref(b.slot(2).cfg))
    _raise_word(a, a.slot(2))
    a.train_step(x, y)
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        assert a.check_exchange(a.slot(2))
    assert a.slot(2).cfg.flags == a.FALLBACK_FLAGS and b.slot(2).cfg.flags == 0 and b.cfg_flags == 0
    assert b.lib.cdll.gt_step_launches(ctypes.byref(b.slot(2).cfg)) == quad_launches          # b still runs four workgroups per sequence
    assert a.lib.cdll.gt_step_launches(ctypes.byref(a.slot(2).cfg)) != quad_launches          # a does not (no fused last-forward / first-backward launch)
    p = b.params.clone()
    b.train_step(x, y)
    assert not torch.equal(b.params, p) and b.exchange_timeouts == 0
    bad = a.slot(2).cfg.__class__.from_buffer_copy(bytes(a.slot(2).cfg)); bad.flags = 64
    assert a.lib.cdll.gt_workspace_bytes(ctypes.byref(bad)) == 0 and b"flags" in a.lib.cdll.gt_last_error()


def test_public_optimizer_step_is_plain_for_any_n():
    """gt_optimizer_step(..., n, ...) takes any n -- a sub-range, an unpadded buffer: every element is updated and a
    non-zero last element is a gradient like any other (the guard element belongs to gt_train_step / gt_optimizer_step_ws)."""
    import ctypes
    import numpy as np
    from harness import emu_lib
    from transformergrooveinfilling_amd import _lib
    lib = emu_lib()
    for algo in (0, 1):
        for n in (5, 7, 1024, 1027):
            rng = np.random.default_rng(n)
            p, g = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
            g[-1] = 3.0
            m, v = np.zeros(n, np.float32), np.zeros(n, np.float32)
            st = np.frombuffer(bytes(_lib.GtStepState(1, 2, 0, 0, 0.5, 1.0, 0.9, 0.999, 1e-8)), np.uint8).copy()
            p0 = p.copy()
            ptr = lambda a_: ctypes.c_void_p(a_.ctypes.data)
            lib.call("gt_optimizer_step", algo, ptr(p), ptr(g), ptr(m), ptr(v), ctypes.c_int64(n), ptr(st), 1, None)
            assert (p != p0).all() and (g == 0).all(), (algo, n)
            if algo == 0:
                assert np.allclose(p[-1], p0[-1] - 0.5 * 3.0)
            s2 = _lib.GtStepState.from_buffer_copy(st.tobytes())
            assert s2.step == 1 and s2.opt_step == 1


def test_strict_mode_raises():
    from transformergrooveinfilling_amd import layout
    eng = _engine(2)
    eng.xchg_strict = True
    x, y = layout.synthetic_batch(2, 16, seed=9)
    eng.train_step(torch.from_numpy(x), torch.from_numpy(y))
    _raise_word(eng, eng.slot(2))
    with pytest.raises(RuntimeError, match="pair exchange"):
        eng.mean_stats(eng.slot(2))
    eng.lib.call("gt_workspace_init", __import__("ctypes").byref(eng.slot(2).cfg), __import__("ctypes").c_void_p(eng.slot(2).ws.data_ptr()), None)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _dp_worker(rank, world, port, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GT_DP_OVERLAP="1")
    from transformergrooveinfilling_amd import layout, parallel
    parallel.init_distributed("gloo")
    eng = _engine(2, world=world, seed=3 | (rank << 32))
    x, y = layout.synthetic_batch(4, 16, seed=9)
    sl = slice(2 * rank, 2 * rank + 2)
    x, y = torch.from_numpy(x[sl]), torch.from_numpy(y[sl])
    eng.train_step(x, y)
    before = eng.params.clone()
    if rank == 1:
        _raise_word(eng, eng.slot(2))          # ONE rank's exchange times out ...
    eng.train_step(x, y)
    skipped = torch.equal(eng.params, before)  # ... EVERY rank must skip (the flag rides the gradient all-reduce in the guard element)
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        st = eng.mean_stats(eng.slot(2))       # ... and every rank learns of it at the next log point
    nan = bool(torch.isnan(st).all())
    eng.train_step(x, y)
    torch.save({"skipped": skipped, "nan": nan, "params": eng.params.clone(), "timeouts": eng.exchange_timeouts, "before": before}, out % rank)
    dist.barrier(); dist.destroy_process_group()


def test_data_parallel_ranks_skip_together(tmp_path):
    world, port = 2, _free_port()
    out = str(tmp_path / "r%d.pt")
    run_ranks(_dp_worker, world, out)
    a, b = torch.load(out % 0), torch.load(out % 1)
    assert a["skipped"] and b["skipped"] and a["nan"] and b["nan"]
    assert a["timeouts"] == 1 and b["timeouts"] == 1
    assert torch.equal(a["params"], b["params"]) and not torch.equal(a["params"], a["before"])      # replicas identical, training resumed


def _poll_worker(rank, world, port, out):
    """a recovery re-captures graphs / changes the launch sequence, so in a multi-rank run it must happen on the same
    step on every rank: the step-path poll decides from the ALL-REDUCED guard element, never from the rank's own word."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from transformergrooveinfilling_amd import engine as E, layout, parallel
    E.XCHG_POLL_EVERY = 3
    parallel.init_distributed("gloo")
    eng = _engine(2, world=world, seed=3 | (rank << 32))
    x, y = layout.synthetic_batch(4, 16, seed=9)
    sl = slice(2 * rank, 2 * rank + 2)
    x, y = torch.from_numpy(x[sl]), torch.from_numpy(y[sl])
    eng.train_step(x, y)                       # step 1: the first poll (clean)
    log = []
    if rank == 1:
        _raise_word(eng, eng.slot(2))
        with warnings.catch_warnings(record=True):
            warnings.simplefilter("always")
            eng.forward(x)                     # a rank-LOCAL observation (evaluation forward on one rank) must not fall back on its own ...
        log.append(("local", eng.exchange_timeouts, eng.cfg_flags))
        _raise_word(eng, eng.slot(2))          # (... it zeroed the region for its repeat; raise it again for the train steps)
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        for i in range(2, 5):
            eng.train_step(x, y)               # steps 2, 3 (poll: the all-reduced guard of step 3 is set -> both recover), 4 (trains)
            log.append((i, eng.exchange_timeouts, eng.cfg_flags))
    torch.save({"log": log, "params": eng.params.clone(), "skipped": eng.skipped_updates}, out % rank)
    dist.barrier(); dist.destroy_process_group()


def test_data_parallel_recovery_is_collective(tmp_path):
    world, port = 2, _free_port()
    out = str(tmp_path / "p%d.pt")
    run_ranks(_poll_worker, world, out)
    a, b = torch.load(out % 0), torch.load(out % 1)
    F = 3                                                               # CFG_NO_QUAD | CFG_NO_LN_XCHG
    assert b["log"][0] == ("local", 1, 0)                               # the local retry counted a time-out and kept the training schedule
    assert [e[2] for e in a["log"]] == [0, F, F]                        # rank 0 (clean itself) falls back on step 3 ...
    assert [e[2] for e in b["log"][1:]] == [0, F, F]                    # ... exactly where rank 1 does
    assert a["log"][-1][1] == 1 and b["log"][-1][1] == 2
    assert torch.equal(a["params"], b["params"])
    assert b["skipped"] == 2                                            # steps 2 and 3 on the rank whose word was raised (its device counted them)


def _tune_fail_worker(rank, world, port, out):
    """VERDICT r05 #5a: rank 1 raises inside recipe 2 (the bucketed one), between its two all-reduces: it must complete the collectives the
    other rank is in, keep issuing them for the recipe's remaining steps, and both ranks must discard the recipe."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from transformergrooveinfilling_amd import layout, parallel
    parallel.init_distributed("gloo")
    eng = _engine(2, world=world, seed=3 | (rank << 32), dims=DIMS2)
    x, y = layout.synthetic_batch(4, 16, seed=9)
    sl = slice(2 * rank, 2 * rank + 2)
    eng.x.copy_(torch.from_numpy(x[sl])); eng.y.copy_(torch.from_numpy(y[sl]))
    p0, st0 = eng.params.clone(), eng.state.clone()
    real, calls = eng._enqueue_step, {"n": 0}

    def flaky(s, skip_update):
        if rank == 1 and skip_update == 3:     # the second half of a bucketed backward: bucket 0's all-reduce is out, bucket 1's is not
            calls["n"] += 1
            if calls["n"] == 2:
                raise RuntimeError("injected failure inside the bucketed recipe")
        return real(s, skip_update)
    eng._enqueue_step = flaky
    tune = eng.autotune_dp(steps=3, warmup=1)
    eng._enqueue_step = real
    same = torch.equal(eng.params, p0) and torch.equal(eng.state, st0)
    eng.train_step()
    torch.save({"tune": tune, "same": same, "params": eng.params.clone(), "overlap": eng.overlap_allreduce, "flags": eng.cfg_flags}, out % rank)
    dist.barrier(); dist.destroy_process_group()


def test_data_parallel_autotune_survives_a_rank_that_raises(tmp_path):
    world, port = 2, _free_port()
    out = str(tmp_path / "f%d.pt")
    run_ranks(_tune_fail_worker, world, out)
    a, b = torch.load(out % 0), torch.load(out % 1)
    assert a["same"] and b["same"]
    assert a["tune"]["chosen"] == b["tune"]["chosen"] == "plain_eager"
    assert a["tune"]["modes"]["buckets_eager"] is None and b["tune"]["modes"]["buckets_eager"] is None      # +inf on rank 1 -> MAX -> dropped everywhere
    assert "errors" in b["tune"] and "injected" in b["tune"]["errors"]["buckets_eager"] and "errors" not in a["tune"]
    assert not a["overlap"] and not b["overlap"] and a["flags"] == b["flags"] == 0
    assert torch.equal(a["params"], b["params"])


def _tune_worker(rank, world, port, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from transformergrooveinfilling_amd import layout, parallel
    parallel.init_distributed("gloo")
    eng = _engine(2, world=world, seed=3 | (rank << 32), dims=DIMS2)
    x, y = layout.synthetic_batch(4, 16, seed=9)
    sl = slice(2 * rank, 2 * rank + 2)
    eng.x.copy_(torch.from_numpy(x[sl])); eng.y.copy_(torch.from_numpy(y[sl]))
    p0, st0 = eng.params.clone(), eng.state.clone()
    tune = eng.autotune_dp(steps=2, warmup=1)
    same = torch.equal(eng.params, p0) and torch.equal(eng.state, st0)            # the run that follows is the run that would have been
    eng.train_step()
    torch.save({"tune": tune, "same": same, "params": eng.params.clone(), "overlap": eng.overlap_allreduce}, out % rank)
    dist.barrier(); dist.destroy_process_group()


def test_data_parallel_autotune_agrees_across_ranks(tmp_path):
    world, port = 2, _free_port()
    out = str(tmp_path / "t%d.pt")
    run_ranks(_tune_worker, world, out)
    a, b = torch.load(out % 0), torch.load(out % 1)
    assert a["same"] and b["same"]
    assert a["tune"]["chosen"] == b["tune"]["chosen"] and a["tune"]["modes"] == b["tune"]["modes"]      # (max over ranks: one table)
    assert set(a["tune"]["modes"]) == {"plain_eager", "buckets_eager"}                                  # host memory: no hipGraph recipes
    assert a["overlap"] == b["overlap"] == (a["tune"]["chosen"] == "buckets_eager")
    assert torch.equal(a["params"], b["params"])
    from transformergrooveinfilling_amd import layout
    ref = _engine(4, dims=DIMS2)
    x, y = layout.synthetic_batch(4, 16, seed=9)
    ref.train_step(torch.from_numpy(x), torch.from_numpy(y))
    assert (ref.params - a["params"]).abs().max() < 1e-6


def test_layout_switch_after_a_slot_was_sized_remakes_the_slot():
    """the workspace layout depends on process-global switches (gt_set_operand_shadows); a cached slot sized before such a
    switch changed must not be reused -- gt_layout_epoch() tells, StepEngine.slot re-makes the slot and training goes on from the same parameters."""
    from transformergrooveinfilling_amd import layout
    eng = _engine(2, dims=DIMS2)
    ref = _engine(2, dims=DIMS2)
    lib = eng.lib
    x, y = layout.synthetic_batch(2, 16, seed=9)
    x, y = torch.from_numpy(x), torch.from_numpy(y)
    try:
        eng.train_step(x, y); ref.train_step(x, y)
        s0, e0 = eng.slot(2), lib.cdll.gt_layout_epoch()
        eng.predict(x)
        assert eng.slot(2) is s0                                     # (nothing changed: the slot is reused)
        lib.cdll.gt_set_operand_shadows(1)
        assert lib.cdll.gt_layout_epoch() == e0 + 1
        lib.cdll.gt_set_operand_shadows(1)
        assert lib.cdll.gt_layout_epoch() == e0 + 1                  # (same value again: no change)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            eng.train_step(x, y)
        assert any("workspace layout changed" in str(m.message) for m in w)
        assert eng.slot(2) is not s0 and eng.slot(2).layout_epoch == e0 + 1
        assert not eng._predict_ws or eng._predict_epoch == e0 + 1
        eng.predict(x)
        assert eng._predict_epoch == e0 + 1
        lib.cdll.gt_set_operand_shadows(-1)                         # (ref's slot is re-made too: same numbers either way)
        ref.train_step(x, y)
        assert torch.allclose(eng.params, ref.params, rtol=0, atol=0)
    finally:
        lib.cdll.gt_set_operand_shadows(-1)
