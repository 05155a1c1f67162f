"""Multi-GPU layout of the hot path: independent polynomials / commitments shard across the GPUs of one node
with NO data-path collective; per-GPU results are gathered on the host.

Two forms.  (1) Inside ONE process: the library's own sharder (``lsr_*_sharded`` in include/lambda_snark/batch.h; Python:
``sharded_ntt``, ``sharded_commit_words``, ``sharded_matvec``) — one host thread and stream per device, slices copied
device -> host straight into one pinned array.  (2) One process per GPU (rank r drives device LOCAL_RANK), as `bench.py` is
launched: this module.  ``torch.distributed`` is plumbing only — the gather moves host rows as raw tensors over gloo, never
device buffers over xGMI, and nothing on the data path is a collective.
"""
import numpy as np


def shard_bounds(batch, world_size, rank):
    """Contiguous slice [lo, hi) of a batch for `rank`: sizes differ by at most one, earlier ranks take the extra."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad world_size / rank")
    base, extra = divmod(batch, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


_HOST_GROUP = None


def host_group(group=None):
    """The process group the host gathers run on.  They move CPU tensors, which an NCCL/RCCL group rejects: when the caller gives no
    group and the default one is not gloo (bench.py initialises "nccl" for its barrier), a gloo group over all ranks is created once
    and reused.  Collective: every rank must make its first call together.  An explicit non-gloo group is an error (round-2 advisor)."""
    global _HOST_GROUP
    import torch.distributed as dist
    if group is not None:
        if dist.get_backend(group) != "gloo":
            raise ValueError("gather_batches moves host tensors: pass a gloo group (got backend %r)" % dist.get_backend(group))
        return group
    if dist.get_backend() == "gloo":
        return None
    if _HOST_GROUP is None:
        _HOST_GROUP = dist.new_group(backend="gloo")
    return _HOST_GROUP


def gather_batches(local, dst=0, group=None):
    """Gather per-rank [local_batch, ...] host arrays into ONE preallocated array on `dst` (None elsewhere), in rank order.
    The rows travel as raw tensor bytes (``dist.gather`` of int64 views, no pickling); the slice sizes follow from
    ``shard_bounds`` of the summed batch, so no size exchange is needed beyond one small all_gather."""
    import torch
    import torch.distributed as dist
    local = np.ascontiguousarray(local)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    group = host_group(group)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([local.shape[0]], dtype=torch.int64), group=group)
    counts = [int(c.item()) for c in counts]
    row_shape = local.shape[1:]
    row_bytes = (int(np.prod(row_shape)) if row_shape else 1) * local.dtype.itemsize
    pad = max(counts)                                    # dist.gather wants equal shapes: pad to the largest slice (<= 1 row more)
    send = torch.zeros((pad, row_bytes), dtype=torch.uint8)
    if local.shape[0]:
        send[: local.shape[0]] = torch.from_numpy(local.reshape(local.shape[0], -1).view(np.uint8).reshape(local.shape[0], row_bytes))
    parts = [torch.empty((pad, row_bytes), dtype=torch.uint8) for _ in range(world)] if rank == dst else None
    dist.gather(send, parts, dst=dst, group=group)
    if rank != dst:
        return None
    out = np.empty((sum(counts),) + tuple(row_shape), dtype=local.dtype)
    flat = out.reshape(out.shape[0], -1).view(np.uint8).reshape(out.shape[0], row_bytes) if out.shape[0] else None
    at = 0
    for c, p in zip(counts, parts):
        if c:
            flat[at:at + c] = p[:c].numpy()
            at += c
    return out


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

