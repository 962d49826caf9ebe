"""Data-parallel path on CPU: world_size 2 over gloo.  Each rank runs forward+loss+backward of ITS shard through the
C ABI (host-emulator build of the kernels), gradients are summed bucket by bucket (gt_grad_buckets: the upper bucket's all-reduce is started
between the two halves of backward) and averaged by grad_scale in the optimizer kernel -> identical replicas that match a single process on the full batch."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from harness import run_ranks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from harness import Runner, cfg_dict
    from oracle import numpy_groove as ng
    from transformergrooveinfilling_amd import parallel
    r_, l_, w_ = parallel.init_distributed("gloo")
    assert (r_, w_) == (rank, world)
    cfg = cfg_dict(32, 4, 16, 2)
    B = 4
    P = ng.init_params(cfg, seed=5 + rank, perturb=0.05)                   # deliberately different per rank ...
    x, y = ng.synthetic_batch(B, 16, seed=9)
    run = Runner(cfg, B // world, "emu", lr=0.05, seq=False)       # two gradient buckets: the one-kernel-per-op path
    run.set_params(P)
    flat = torch.from_numpy(run.params.numpy())                            # shares memory with the runner's buffer
    parallel.broadcast_parameters(flat, src=0)                             # ... then made identical
    sl = slice(rank * (B // world), (rank + 1) * (B // world))
    buckets = run.lib.grad_buckets(run.c)
    assert len(buckets) == 2                                               # 2 encoder layers: upper layer + heads first
    run.train_step(x[sl], y[sl], 0.47, algo=0, skip_update=2)              # forward, loss, backward until bucket 0 is final
    g = torch.from_numpy(run.grads.numpy())
    w0 = parallel.allreduce_bucket_async(g, *buckets[0])                   # reduced while the rest of backward runs
    run.train_step(x[sl], y[sl], 0.47, algo=0, skip_update=3)
    w1 = parallel.allreduce_bucket_async(g, *buckets[1])
    w0.wait(); w1.wait()
    st = run.step_state()
    st.grad_scale = 1.0 / world
    run.state.numpy()[:] = np.frombuffer(bytes(st), dtype=np.uint8)
    new = run.optimizer_step(0)
    torch.save({"params": new, "rank": rank}, out % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_dp2_matches_single_process(tmp_path):
    world, port = 2, _free_port()
    out = str(tmp_path / "rank%d.pt")
    run_ranks(_worker, world, out)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from harness import Runner, cfg_dict
    from oracle import numpy_groove as ng
    a = torch.load(out % 0, weights_only=False)["params"]
    b = torch.load(out % 1, weights_only=False)["params"]
    cfg = cfg_dict(32, 4, 16, 2)
    P = ng.init_params(cfg, seed=5, perturb=0.05)
    x, y = ng.synthetic_batch(4, 16, seed=9)
    single = Runner(cfg, 4, "emu", lr=0.05, seq=False)
    single.set_params(P)
    single.train_step(x, y, 0.47, algo=0)
    ref = single.unflatten(single.params.numpy())
    for k in ref:
        assert np.array_equal(a[k], b[k]), k                                # replicas stay identical
        assert np.abs(a[k] - ref[k]).max() < 1e-6, k                         # == one process on the whole batch


ENGINE_CASES = {
    # name: (model dims, sequences in the global batch, gradient buckets the overlapped step must use)
    "seq_d32": (dict(d_model=32, n_heads=4, dim_feedforward=16, num_encoder_layers=2), 4, 1),      # sequence-resident, one launch per direction: one bucket
    "op_d48": (dict(d_model=48, n_heads=4, dim_feedforward=24, num_encoder_layers=2), 4, 2),       # outside the sequence kernels: one kernel per op, bucketed backward
    "ride_d128": (dict(d_model=128, n_heads=4, dim_feedforward=64, num_encoder_layers=3), 4, 2),   # SPLIT phases with rider weight gradients: cut after phase 2
    # four ranks, two buckets of very different sizes (a 3-layer model cut after its top layer), and a RAGGED last batch: the second step
    # has half the sequences of the first (another slot, another workspace on every rank)
    "op_d48_w4": (dict(d_model=48, n_heads=4, dim_feedforward=24, num_encoder_layers=3), (8, 4), 2),
    "ride_d128_w4": (dict(d_model=128, n_heads=4, dim_feedforward=32, num_encoder_layers=2), (8, 4), 2),
}


def _engine_dims(case):
    dims, B, nb = ENGINE_CASES[case]
    return dict(dims, num_decoder_layers=0, dropout=0.0, embedding_size_src=16), B, nb


def _step_sizes(B):
    """global batch of each of the two steps"""
    return tuple(B) if isinstance(B, tuple) else (B, B)


def _engine_worker(rank, world, port, out, overlap, case):
    """the PRODUCT's data-parallel step sequence (StepEngine.train_step: fwd+loss+bwd, all-reduce(s) of the flat gradient
    buffer, fused update averaging by grad_scale) on host memory: explicit emulator library + gloo"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      GT_DP_OVERLAP="1" if overlap else "0")
    from harness import emu_lib
    from transformergrooveinfilling_amd import layout, parallel
    from transformergrooveinfilling_amd.engine import StepEngine
    parallel.init_distributed("gloo")
    dims, B, nb = _engine_dims(case)
    sizes = _step_sizes(B)
    eng = StepEngine(batch_size=sizes[0] // world, optimizer="sgd", learning_rate=0.05, hit_loss_penalty=0.47, seed=3 | (rank << 32),
                     device="cpu", world_size=world, lib=emu_lib(), **dims)
    assert eng.overlap_allreduce == bool(overlap)
    if overlap:                                   # the bucketed branch of StepEngine.train_step must really be the one that runs
        bk = eng.lib.grad_buckets(eng.slot(sizes[0] // world).cfg)
        assert len(bk) == nb
        if isinstance(B, tuple):
            assert bk[0][1] != bk[1][1]           # (uneven buckets)
    eng.load_named(layout.init_params(dims, seed=5))
    x, y = layout.synthetic_batch(max(sizes), 16, seed=9)
    for n in sizes:
        sl = slice(rank * (n // world), (rank + 1) * (n // world))
        eng.train_step(torch.from_numpy(x[:n][sl]), torch.from_numpy(y[:n][sl]))
    mean = eng.mean_stats(eng.slot(sizes[-1] // world)).clone()
    eng.B = sizes[-1] // world                    # (eng.stats = the last step's slot)
    torch.save({"params": eng.params.clone(), "stats": eng.stats.clone(), "mean_stats": mean}, out % rank)
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("case,overlap,world", [("seq_d32", 0, 2), ("seq_d32", 1, 2), ("op_d48", 1, 2), ("op_d48", 0, 2), ("ride_d128", 1, 2),
                                                ("op_d48_w4", 1, 4), ("ride_d128_w4", 1, 4)])
def test_engine_dp_matches_single_process(tmp_path, case, overlap, world):
    port = _free_port()
    out = str(tmp_path / "eng%d.pt")
    run_ranks(_engine_worker, world, out, overlap, case)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from harness import emu_lib
    from transformergrooveinfilling_amd import layout
    from transformergrooveinfilling_amd.engine import StepEngine
    a, b = torch.load(out % 0), torch.load(out % (world - 1))
    dims, B, _ = _engine_dims(case)
    sizes = _step_sizes(B)
    eng = StepEngine(batch_size=sizes[0], optimizer="sgd", learning_rate=0.05, hit_loss_penalty=0.47, seed=3, device="cpu", lib=emu_lib(), **dims)
    eng.load_named(layout.init_params(dims, seed=5))
    x, y = layout.synthetic_batch(max(sizes), 16, seed=9)
    for n in sizes:
        eng.train_step(torch.from_numpy(x[:n]), torch.from_numpy(y[:n]))
    eng.B = sizes[-1]
    for r in range(1, world):
        assert torch.equal(a["params"], torch.load(out % r)["params"])       # replicas stay identical
    assert torch.equal(a["params"], b["params"])
    assert (a["params"] - eng.params).abs().max() < 1e-6                      # == one process on the whole batch
    assert torch.allclose(a["mean_stats"], b["mean_stats"])                   # logged stats are the all-rank mean ...
    assert abs(float(a["mean_stats"][0]) - float(eng.stats[0])) < 1e-5        # ... = the loss of the whole batch
    assert abs(float(a["stats"][0]) - float(b["stats"][0])) > 1e-6            # (rank-local values differ)


def test_engine_refuses_host_memory_without_an_explicit_library():
    from transformergrooveinfilling_amd.engine import StepEngine
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        StepEngine(32, 4, 16, 2, device="cpu")
