"""Plain data-parallel support (the reference has no multi-GPU code; BASELINE.json asks for DP only).

One process per GPU (torchrun-style env: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), backend "nccl" (= RCCL over
xGMI on ROCm) on GPUs, "gloo" in CPU tests.  Sequences are independent, so ranks take disjoint shards of every
epoch's seeded permutation; the only exchange is ONE all-reduce(SUM) of the flat gradient buffer per step -- the
loss is a mean over (B,T), so the sum is averaged by 1/world inside the fused optimizer (gt_step_state.grad_scale).
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """-> (rank, local_rank, world).  No-op (0,0,1) when WORLD_SIZE is unset or 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        return 0, 0, 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group(backend, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    return rank, local, world


def broadcast_parameters(flat, src=0):
    """Identical replicas at start: rank `src`'s flat parameter buffer to everyone."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat, src)
    return flat


def allreduce_gradients(flat, average=False):
    """ONE collective over the flat gradient buffer.  average=False leaves the 1/world to the optimizer kernel."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat /= dist.get_world_size()
    return flat


def allreduce_bucket_async(flat, offset, count):
    """Start the SUM all-reduce of flat[offset:offset+count] (one gradient bucket of gt_grad_buckets) and return its work
    handle (None without a process group); call .wait() before the optimizer reads the buffer.  Issued between the two
    halves of a bucketed backward, the collective runs while the second half computes."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        return dist.all_reduce(flat[offset:offset + count], op=dist.ReduceOp.SUM, async_op=True)
    return None


class ShardedBatchSampler:
    """Replaces DataLoader(shuffle=True) (ref:train.py:156-158) under DP: every rank draws the SAME seeded
    permutation of the epoch and keeps indices rank, rank+world, ...; yields lists of `batch_size` indices.
    Ragged tails are dropped so that every rank runs the same number of steps (collectives stay matched)."""

    def __init__(self, n_items, batch_size, rank=0, world=1, seed=0, drop_last=True):
        self.n, self.bs, self.rank, self.world, self.seed, self.drop_last = n_items, batch_size, rank, world, seed, drop_last
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        per_rank = self.n // self.world
        return per_rank // self.bs if self.drop_last else (per_rank + self.bs - 1) // self.bs

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.seed * 1000003 + self.epoch)
        perm = torch.randperm(self.n, generator=g)
        per_rank = self.n // self.world
        mine = perm[self.rank: per_rank * self.world: self.world]
        for i in range(len(self)):
            b = mine[i * self.bs:(i + 1) * self.bs]
            if len(b):
                yield b.tolist()


class DeviceBatchLoader:
    """The input pipeline on the device.  The reference's datasets are two dense fp32 tensors of tens of MB
    (ref:dataset.py:263-264,355-356) fed through DataLoader(shuffle=True, pin_memory=True) (ref:train.py:156-158); at
    0.3 ms per train step the collate + H2D copy of a host DataLoader would be several times the step.  Here both tensors
    live in HBM; every epoch draws the SAME seeded permutation on every rank (the one ShardedBatchSampler draws), a rank
    keeps indices rank, rank+world, ....  Two ways to consume it:
      * ``index_batches()`` -> device index tensors only: train_loop hands them to StepEngine.train_step_indexed, where ONE
        gather launch (gt_gather_batch, part of the step's hipGraph) fills both static step inputs -- no batch tensor is ever
        materialised outside the step (the fast path);
      * iteration -> (x, y, idx) like the reference's dataset (two index_select gathers), for any other consumer.
    Under DP ragged tails are dropped (matched collectives); single-process keeps the last partial batch, as DataLoader's
    default does."""

    def __init__(self, x, y, batch_size, device, rank=0, world=1, seed=0):
        self.x = torch.as_tensor(x, dtype=torch.float32).to(device).contiguous()
        self.y = torch.as_tensor(y, dtype=torch.float32).to(device).contiguous()
        assert self.x.shape[0] == self.y.shape[0]
        self.n, self.bs, self.rank, self.world, self.seed = int(self.x.shape[0]), int(batch_size), rank, world, seed
        self.device, self.epoch = device, 0
        self.batch_size = self.bs

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        per_rank = self.n // self.world
        return per_rank // self.bs if self.world > 1 else (per_rank + self.bs - 1) // self.bs

    def index_batches(self):
        """The epoch's batches as int64 index tensors on the device (views of ONE permutation copied once per epoch)."""
        g = torch.Generator()
        g.manual_seed(self.seed * 1000003 + self.epoch)
        perm = torch.randperm(self.n, generator=g)                      # host generator: identical on every rank and platform
        per_rank = self.n // self.world
        mine = perm[self.rank: per_rank * self.world: self.world].to(self.device)   # ONE small H2D copy per epoch
        for i in range(len(self)):
            yield mine[i * self.bs:(i + 1) * self.bs]

    def __iter__(self):
        for idx in self.index_batches():
            yield self.x.index_select(0, idx), self.y.index_select(0, idx), idx
