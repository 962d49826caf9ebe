"""Data-parallel step over RCCL with TWO real ranks (one per GPU): StepEngine's step sequence (fwd+loss+bwd graph, all-reduce of
the flat gradient buffer -- whole or in two buckets overlapped with the second half of backward --, fused update averaging by
grad_scale) must leave identical replicas that equal ONE process stepping on the whole batch.  Skips on boxes with fewer than
two GPUs (the 1-GPU test boxes): the same sequence runs there over gloo on host memory (tests/test_dp_gloo.py)."""
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIMS = dict(d_model=128, n_heads=4, dim_feedforward=512, num_encoder_layers=3, num_decoder_layers=0, dropout=0.0, embedding_size_src=16)
B = 32


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out, overlap):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      GT_DP_OVERLAP="1" if overlap else "0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from transformergrooveinfilling_amd import layout, parallel
    from transformergrooveinfilling_amd.engine import StepEngine
    r, local, w = parallel.init_distributed("nccl")
    eng = StepEngine(batch_size=B // world, optimizer="sgd", learning_rate=0.05, hit_loss_penalty=0.47, seed=3 | (rank << 32),
                     device="cuda:%d" % local, world_size=world, **DIMS)
    eng.load_named(layout.init_params(DIMS, seed=5))
    if overlap:        # sequence-resident SPLIT phases with rider weight gradients: the bucketed branch of StepEngine.train_step runs
        assert len(eng.lib.grad_buckets(eng.slot(B // world).cfg)) == 2
    x, y = layout.synthetic_batch(B, 16, seed=9)
    sl = slice(rank * (B // world), (rank + 1) * (B // world))
    for _ in range(3):
        eng.train_step(torch.from_numpy(x[sl]), torch.from_numpy(y[sl]))
    mean = eng.mean_stats(eng.slot(B // world)).cpu()
    torch.cuda.synchronize()
    torch.save({"params": eng.params.cpu(), "mean_stats": mean, "world": dist.get_world_size(), "backend": dist.get_backend()}, out % rank)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [0, 1])
def test_two_rccl_ranks_match_single_process(tmp_path, overlap):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL with 2 ranks)")
    import torch.multiprocessing as mp
    from transformergrooveinfilling_amd import layout
    from transformergrooveinfilling_amd.engine import StepEngine
    out = str(tmp_path / "r%d.pt")
    mp.start_processes(_worker, args=(2, _free_port(), out, overlap), nprocs=2, join=True, start_method="spawn")
    a, b = torch.load(out % 0), torch.load(out % 1)
    assert a["world"] == 2 and a["backend"] == "nccl"
    eng = StepEngine(batch_size=B, optimizer="sgd", learning_rate=0.05, hit_loss_penalty=0.47, seed=3, device="cuda:0", **DIMS)
    eng.load_named(layout.init_params(DIMS, seed=5))
    x, y = layout.synthetic_batch(B, 16, seed=9)
    for _ in range(3):
        eng.train_step(torch.from_numpy(x), torch.from_numpy(y))
    ref = eng.params.cpu()
    assert torch.equal(a["params"], b["params"])                                   # replicas stay identical
    assert (a["params"] - ref).abs().max() < 2e-6 * max(1.0, ref.abs().max())     # == one process on the whole batch
    assert abs(float(a["mean_stats"][0]) - float(eng.stats[0])) < 1e-5            # logged loss = mean over ranks = whole-batch loss
