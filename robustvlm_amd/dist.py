"""Data-parallel helpers: one process per GPU, images sharded across ranks, no collective inside the
attack (every quantity of pgd / apgd_train / APGDAttack is per-sample).  Replaces the
reference's single-process torch.nn.DataParallel (train/adversarial_training_clip.py:184-191), which
re-broadcasts 1.2 GB of parameters on every forward.  The only collectives are the metric
aggregation here and, in the outer trainer, one all-reduce of encoder gradients per optimizer step."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous [lo, hi) slice of a global batch for `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    lo, hi = shard_range(t.shape[0], rank, world)
    return t[lo:hi]


def max_over_ranks(value: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def allreduce_mean_(tensors, world: int | None = None):
    """In-place mean all-reduce of a list of gradient tensors as ONE flat bucket (RCCL over xGMI is
    per-link bound: few large messages, not many small ones)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    world = world or dist.get_world_size()
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(world)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n


def allreduce_sum_span(flat: torch.Tensor, lo: int, hi: int, group=None, device_collectives: bool = True):
    """Sum all-reduce of the contiguous slice flat[lo:hi] (one gradient bucket of the trainer's flat fp32 buffer).
    With device collectives (backend nccl = RCCL over xGMI) the call is asynchronous and returns a work handle whose
    wait() makes the CURRENT STREAM wait - the bucket's reduction overlaps whatever the caller launches next.  A
    backend without device collectives (gloo in the CPU / one-GPU tests) reduces through a host copy, synchronously."""
    view = flat[lo:hi]
    if device_collectives or not view.is_cuda:
        work = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=group, async_op=True)
        if not view.is_cuda:
            work.wait()
            return None
        return work
    host = view.cpu()
    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
    view.copy_(host)
    return None
