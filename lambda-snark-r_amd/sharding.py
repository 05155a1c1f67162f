"""Multi-GPU layout of the hot path: independent polynomials / commitments shard across the GPUs of one node
with NO data-path collective; per-GPU results are gathered on the host.

One process per GPU (rank r drives device LOCAL_RANK).  ``torch.distributed`` is used only as plumbing:
the gather moves host arrays (gloo or nccl-with-host-objects both work), never device buffers over xGMI.
"""
import numpy as np


def shard_bounds(batch, world_size, rank):
    """Contiguous slice [lo, hi) of a batch for `rank`: sizes differ by at most one, earlier ranks take the extra."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad world_size / rank")
    base, extra = divmod(batch, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_batches(local, dst=0, group=None):
    """Gather per-rank [local_batch, ...] host arrays into one array on `dst` (None elsewhere), in rank order."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return np.ascontiguousarray(local)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    parts = [None] * world if rank == dst else None
    dist.gather_object(np.ascontiguousarray(local), parts, dst=dst, group=group)
    if rank != dst:
        return None
    parts = [p for p in parts if p.shape[0] > 0]
    return np.concatenate(parts, axis=0) if parts else np.ascontiguousarray(local)[:0]


def sharded_transform(polys, transform, group=None, dst=0):
    """Every rank holds the full host batch `polys` ([batch][n]); each transforms its contiguous slice with
    `transform` (e.g. ``NttContext.forward_batch`` bound to this rank's GPU) and rank `dst` receives the result."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(polys.shape[0], world, rank)
    local = transform(polys[lo:hi]) if hi > lo else polys[:0].copy()
    return gather_batches(local, dst=dst, group=group)


def sharded_map(inputs, fn, group=None, dst=0):
    """Several aligned host batches in, several out: every rank holds `inputs` (arrays with one leading batch dimension),
    runs ``fn(*slices)`` on its contiguous slice (e.g. ``QuotientPlan.quotient_batch`` -> (coefficients, lengths)) and rank
    `dst` receives the tuple of gathered outputs; other ranks get None."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    batch = inputs[0].shape[0]
    if any(a.shape[0] != batch for a in inputs):
        raise ValueError("inputs must share the batch dimension")
    lo, hi = shard_bounds(batch, world, rank)
    outs = fn(*(a[lo:hi] for a in inputs))
    gathered = [gather_batches(np.asarray(o), dst=dst, group=group) for o in outs]
    return None if rank != dst else tuple(gathered)

