"""GPU robustness: concurrent callers (the Rust wrappers are Send: a handle may be used from any thread, one call at a
time; distinct handles may run concurrently — SURVEY.md §8(b) "Threading"), no device-memory growth across
create/use/free cycles, and randomized small cases against the oracle."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

Q44 = 17592169062401


def test_threads_with_private_and_shared_contexts(pkg, oracle):
    n = 4096
    shared = pkg.NttContext(Q44, n)
    errors = []

    def worker(seed, ctx):
        try:
            own = ctx or pkg.NttContext(Q44, n)
            for it in range(20):
                a = oracle.splitmix(seed * 1000 + it, Q44, 3 * n).reshape(3, n)
                f = own.forward_batch(a)
                if not np.array_equal(f, oracle.ntt_forward(Q44, n, a)) or not np.array_equal(own.inverse_batch(f), a):
                    errors.append((seed, it))
                one = own.forward(a[0])
                if not np.array_equal(one, f[0]):
                    errors.append((seed, it, "single"))
            if ctx is None:
                own.close()
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(i, None)) for i in range(4)] + [threading.Thread(target=worker, args=(10 + i, shared)) for i in range(4)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    shared.close()
    assert not errors, errors[:5]


def test_threads_committing_on_one_context(pkg, oracle):
    lctx = pkg.LweContext(pkg.Params(q=17592186044417, n=4096, k=2, sigma=3.19), key_seed=99)
    errors = []

    def worker(i):
        for it in range(10):
            msg = [i, it, 7, 9]
            c = pkg.Commitment(lctx, msg, seed=1 + i * 100 + it)
            want = oracle.lwe_commit(17592186044417, 4096, 2, 3.19, 99, msg, 1 + i * 100 + it)
            if not np.array_equal(c.as_words(), want) or not pkg.verify_opening_with_context(lctx, c, msg):
                errors.append((i, it))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(6)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    lctx.close()
    assert not errors


def test_no_device_memory_growth(pkg):
    import torch
    torch.cuda.synchronize()

    def cycle():
        ctx = pkg.NttContext(17592182243329, 65536)
        a = np.arange(2 * 65536, dtype=np.uint64).reshape(2, 65536)
        ctx.inverse_batch(ctx.forward_batch(a))
        ctx.close()
        lctx = pkg.LweContext(pkg.Params(q=12289, n=4096, k=2, sigma=3.19))
        coms = pkg.Commitment.batch(lctx, np.ones((5, 4), dtype=np.uint64), np.arange(1, 6, dtype=np.uint64))
        comb = pkg.Commitment.linear_combine(lctx, coms, [1, 2, 3, 4, 5])
        assert pkg.verify_opening_with_context(lctx, comb, [15, 15, 15, 15])
        for c in coms + [comb]:
            c.free()
        lctx.close()

    for _ in range(3):
        cycle()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(40):
        cycle()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, f"device memory shrank by {(free0 - free1) >> 20} MiB over 40 create/use/free cycles"


def test_randomized_small_cases(pkg, oracle):
    rng = np.random.default_rng(2026)
    primes = {}
    for _ in range(60):
        logn = int(rng.integers(1, 12))
        n = 1 << logn
        bits = int(rng.integers(20, 46))
        q = primes.setdefault((n, bits), oracle.L.oracle_largest_prime_1mod(2 * n, bits))
        if q == 0:
            continue
        batch = int(rng.integers(1, 40))
        a = rng.integers(0, q, size=(batch, n), dtype=np.uint64)
        ctx = pkg.NttContext(q, n)
        f = ctx.forward_batch(a)
        assert np.array_equal(f, oracle.ntt_forward(q, n, a)), (q, n, batch)
        assert np.array_equal(ctx.inverse_batch(f), a), (q, n, batch)
        b = rng.integers(0, q, size=n, dtype=np.uint64)
        assert np.array_equal(ctx.mul_pointwise(a[0], b), oracle.mul_pointwise(q, n, a[0], b))
        ctx.close()


def test_device_api_is_graph_capturable(pkg, oracle):
    """The device-resident entry points only enqueue kernels (no allocation, no synchronisation), so a caller can
    capture them into a HIP graph and replay it (launch-bound loops, e.g. many small batches)."""
    import torch
    q, n, batch = 17592182243329, 65536, 4
    ctx = pkg.NttContext(q, n, device=0)
    host = oracle.splitmix(21, q, batch * n).reshape(batch, n)
    buf = torch.from_numpy(host.view(np.int64)).cuda()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):      # warm-up outside capture
        ctx.forward_device(buf.data_ptr(), batch, side.cuda_stream)
        ctx.inverse_device(buf.data_ptr(), batch, side.cuda_stream)
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        ctx.forward_device(buf.data_ptr(), batch, torch.cuda.current_stream().cuda_stream)
    buf.copy_(torch.from_numpy(host.view(np.int64)))
    graph.replay()
    torch.cuda.synchronize()
    want = oracle.ntt_forward(q, n, host)
    assert np.array_equal(buf.cpu().numpy().view(np.uint64), want)
    graph.replay()                     # second replay transforms the transformed data: compare with the oracle again
    torch.cuda.synchronize()
    assert np.array_equal(buf.cpu().numpy().view(np.uint64), oracle.ntt_forward(q, n, want))
    ctx.close()
