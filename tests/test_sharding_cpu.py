"""CPU suite, part 3: the N>1 path (world_size 2, gloo).  The shard/gather logic is the product's; the
per-shard transform is stood in by the oracle because this container has no GPU."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_the_batch(pkg):
    import importlib
    sh = importlib.import_module("lambda_snark_r_amd.sharding")
    for batch in [0, 1, 5, 8, 1024, 4097]:
        for world in [1, 2, 3, 8]:
            spans = [sh.shard_bounds(batch, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sh.shard_bounds(4, 2, 2)
    assert sh.shard_bounds(1024, 8, 3) == (384, 512)      # config 4: 128 vectors per GPU


def _worker(rank, world, port, batch, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import __graft_entry__ as entry
    import oracle_binding
    entry.load_package()
    import importlib
    sh = importlib.import_module("lambda_snark_r_amd.sharding")
    orc = oracle_binding.load()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q, n = 12289, 256
    polys = orc.splitmix(0xABC, q, batch * n).reshape(batch, n)          # same synthetic batch on every rank
    assert sh.host_group() is None                      # the default group is gloo: the gathers run on it (an NCCL default gets a gloo twin)
    got = sh.sharded_transform(polys, lambda a: orc.ntt_forward(q, n, a))
    if rank == 0:
        np.save(out_path, got)
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("batch", [6, 5, 1])
def test_two_rank_gloo_gather_equals_unsharded(oracle, tmp_path, batch):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, port, batch, out), nprocs=2, join=True)
    q, n = 12289, 256
    polys = oracle.splitmix(0xABC, q, batch * n).reshape(batch, n)
    assert np.array_equal(np.load(out), oracle.ntt_forward(q, n, polys))


def _quotient_worker(rank, world, port, batch, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import __graft_entry__ as entry
    import oracle_binding
    entry.load_package()
    import importlib
    sh = importlib.import_module("lambda_snark_r_amd.sharding")
    orc = oracle_binding.load()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b, c = _quotient_inputs(orc, batch)

    def per_shard(sa, sb, sc):          # stands in for QuotientPlan.quotient_batch on this rank's GPU
        rows = [orc.quotient(x, y, z) for x, y, z in zip(sa, sb, sc)]
        return (np.stack([r[0] for r in rows]) if rows else np.zeros((0, a.shape[1]), np.uint64),
                np.array([r[1] for r in rows], dtype=np.uint32))

    got = sh.sharded_map((a, b, c), per_shard)
    if rank == 0:
        np.savez(out_path, quot=got[0], lens=got[1])
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def _quotient_inputs(orc, batch, m=16):
    q = orc.prover_q
    rng = np.random.default_rng(77)
    a = rng.integers(0, q, size=(batch, m), dtype=np.uint64)
    b = rng.integers(0, q, size=(batch, m), dtype=np.uint64)
    c = np.array([[int(x) * int(y) % q for x, y in zip(ra, rb)] for ra, rb in zip(a, b)], dtype=np.uint64)
    if batch > 1:
        c[1, 3] ^= np.uint64(1)          # one unsatisfied instance crosses the gather as length 0
    return a, b, c


@pytest.mark.parametrize("batch", [5, 1])
def test_two_rank_gloo_quotients(oracle, tmp_path, batch):
    """The prover path shards the same way (independent instances, host gather, no collective on the data path)."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "gathered.npz")
    mp.spawn(_quotient_worker, args=(2, port, batch, out), nprocs=2, join=True)
    a, b, c = _quotient_inputs(oracle, batch)
    got = np.load(out)
    for i in range(batch):
        want, ln = oracle.quotient(a[i], b[i], c[i])
        assert got["lens"][i] == ln and (ln == 0 or np.array_equal(got["quot"][i], want))
    if batch > 1:
        assert got["lens"][1] == 0

