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


def test_threads_mixing_asynchronous_and_synchronous_calls_on_one_context(pkg, oracle):
    """Three kinds of caller on ONE context at the same time: threads that enqueue lsr_lwe_commit_rows_device / lsr_lwe_verify_rows_device
    on streams of their own and synchronise only those streams, threads that derive keys on the device, and threads in the
    synchronous lwe_commit / lwe_verify_opening.  The context orders them (mutex for the enqueue, last-use event for the
    workspaces): every row equals the reference rows, every opening succeeds, every single commitment equals the oracle's."""
    import torch
    q, n, k, batch, msg_len = 17592186044417, 4096, 2, 64, 7
    lctx = pkg.LweContext(pkg.Params(q=q, n=n, k=k, sigma=3.19), key_seed=41)
    rng = np.random.default_rng(41)
    msgs = rng.integers(0, lctx.plain_modulus, size=(batch, msg_len), dtype=np.uint64)
    seeds = rng.integers(1, 2**63, size=batch, dtype=np.uint64)
    keys = lctx.commit_keys(msgs, seeds)
    d_msgs = torch.from_numpy(msgs.view(np.int64)).cuda(); d_keys = torch.from_numpy(keys.view(np.int64)).cuda()
    words = lctx.commitment_words
    ref = torch.zeros((batch, words), dtype=torch.int64, device="cuda")
    lctx.commit_rows_device(d_msgs.data_ptr(), msg_len, batch, d_keys.data_ptr(), ref.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    errors = []

    def rows_worker(i):
        st = torch.cuda.Stream()
        rows = torch.zeros((batch, words), dtype=torch.int64, device="cuda")
        res = torch.zeros(batch, dtype=torch.int32, device="cuda")
        for it in range(15):
            rows.zero_(); res.zero_()
            torch.cuda.current_stream().synchronize()
            lctx.commit_rows_device(d_msgs.data_ptr(), msg_len, batch, d_keys.data_ptr(), rows.data_ptr(), st.cuda_stream)
            lctx.verify_rows_device(rows.data_ptr(), d_msgs.data_ptr(), msg_len, batch, res.data_ptr(), st.cuda_stream)
            st.synchronize()
            if not torch.equal(rows, ref) or int(res.sum().item()) != batch:
                errors.append(("rows", i, it))

    def keys_worker(i):
        st = torch.cuda.Stream()
        out = torch.zeros((batch, 4), dtype=torch.int64, device="cuda")
        for it in range(15):
            out.zero_()
            torch.cuda.current_stream().synchronize()
            lctx.commit_keys_device(d_msgs.data_ptr(), msg_len, seeds, out.data_ptr(), st.cuda_stream)
            st.synchronize()
            if not torch.equal(out, d_keys):
                errors.append(("keys", i, it))

    def single_worker(i):
        for it in range(10):
            j = (7 * i + it) % batch
            c = pkg.Commitment(lctx, [int(x) for x in msgs[j]], int(seeds[j]))
            if not np.array_equal(c.as_words().view(np.int64), ref[j].cpu().numpy()) or not pkg.verify_opening_with_context(lctx, c, [int(x) for x in msgs[j]]):
                errors.append(("single", i, it))

    threads = ([threading.Thread(target=rows_worker, args=(i,)) for i in range(2)] + [threading.Thread(target=keys_worker, args=(i,)) for i in range(1)] +
               [threading.Thread(target=single_worker, args=(i,)) for i in range(2)])
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert np.array_equal(ref[3].cpu().numpy().view(np.uint64), oracle.lwe_commit(q, n, k, 3.19, 41, [int(x) for x in msgs[3]], int(seeds[3])))
    lctx.close()
    assert not errors, errors[:5]


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


# ---- prover path (include/lambda_snark/prover.h) --------------------------------------------------------------------
GOLD = 18446744069414584321


def _gold_instances(rng, m, batch):
    a = rng.integers(0, GOLD, size=(batch, m), dtype=np.uint64)
    b = rng.integers(0, GOLD, size=(batch, m), dtype=np.uint64)
    c = np.array([[int(x) * int(y) % GOLD for x, y in zip(ra, rb)] for ra, rb in zip(a, b)], dtype=np.uint64)
    return a, b, c


def test_quotient_plans_from_threads(pkg, oracle):
    """Private plans run concurrently; one shared plan is used by one call at a time (its mutex serialises the host API)."""
    rng = np.random.default_rng(99)
    m = 256
    cases = [_gold_instances(rng, m, 6) for _ in range(4)]
    want = [[oracle.quotient(a[i], b[i], c[i]) for i in range(6)] for a, b, c in cases]
    shared = pkg.QuotientPlan(m)
    errors = []

    def worker(idx, plan):
        try:
            own = plan or pkg.QuotientPlan(m)
            for _ in range(5):
                quot, lens = own.quotient_batch(*cases[idx])
                for i in range(6):
                    assert lens[i] == want[idx][i][1] and np.array_equal(quot[i], want[idx][i][0])
            if plan is None:
                own.close()
        except Exception as e:          # noqa: BLE001 — collected and re-raised below
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(i, None if i % 2 else shared)) for i in range(4)]
    for t in threads: t.start()
    for t in threads: t.join()
    shared.close()
    assert not errors, errors


def test_quotient_device_api_is_graph_capturable(pkg, oracle):
    import torch
    m, batch = 512, 8
    rng = np.random.default_rng(5)
    a, b, c = _gold_instances(rng, m, batch)
    c[3, 7] ^= np.uint64(1)
    plan = pkg.QuotientPlan(m, device=0)
    da, db, dc = (torch.from_numpy(v.view(np.int64)).cuda() for v in (a, b, c))
    dq = torch.zeros_like(da)
    dl = torch.zeros(batch, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):      # warm-up outside capture: the first call allocates the plan's workspace
        plan.quotient_device(da.data_ptr(), db.data_ptr(), dc.data_ptr(), batch, dq.data_ptr(), dl.data_ptr(), side.cuda_stream)
    side.synchronize()
    dq.zero_(); dl.zero_()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        plan.quotient_device(da.data_ptr(), db.data_ptr(), dc.data_ptr(), batch, dq.data_ptr(), dl.data_ptr(), torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        dq.zero_(); dl.zero_()
        graph.replay()
        torch.cuda.synchronize()
        quot, lens = dq.cpu().numpy().view(np.uint64), dl.cpu().numpy().view(np.uint32)
        for i in range(batch):
            w, ln = oracle.quotient(a[i], b[i], c[i])
            assert lens[i] == ln and (ln == 0 or np.array_equal(quot[i], w))
        assert lens[3] == 0
    # calls outside the capture afterwards: the plan's ordering event was never recorded into the graph, so the host entry point
    # (which waits for it) and a plain asynchronous call both still work
    hq, hl = plan.quotient_batch(a, b, c)
    assert np.array_equal(hl, lens) and np.array_equal(hq[hl > 0], quot[hl > 0])
    dq.zero_(); dl.zero_()
    plan.quotient_device(da.data_ptr(), db.data_ptr(), dc.data_ptr(), batch, dq.data_ptr(), dl.data_ptr(), side.cuda_stream)
    side.synchronize()
    assert np.array_equal(dl.cpu().numpy().view(np.uint32), lens)
    plan.close()


def test_commit_rows_are_graph_capturable_and_leave_the_context_usable(pkg, oracle):
    """lsr_lwe_commit_rows_device + lsr_lwe_verify_rows_device of the tile pipeline captured into a HIP graph (one launch each, no
    allocation after the warm-up), replayed, and then the synchronous lwe_commit / lwe_verify_opening on the same context: they wait
    for the context's last-use event, which must not have been recorded into the capture."""
    import torch
    q, n, k, batch, msg_len = 17592169062401, 4096, 2, 12, 9
    ctx = pkg.LweContext(pkg.Params(q=q, n=n, k=k, sigma=3.19), key_seed=0x5EED)
    rng = np.random.default_rng(8)
    msgs = rng.integers(0, ctx.plain_modulus, size=(batch, msg_len), dtype=np.uint64)
    seeds = rng.integers(1, 2**63, size=batch, dtype=np.uint64)
    keys = ctx.commit_keys(msgs, seeds)
    d_msgs = torch.from_numpy(msgs.view(np.int64)).cuda(); d_keys = torch.from_numpy(keys.view(np.int64)).cuda()
    rows = torch.zeros((batch, ctx.commitment_words), dtype=torch.int64, device="cuda")
    res = torch.zeros(batch, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    both = lambda st: (ctx.commit_rows_device(d_msgs.data_ptr(), msg_len, batch, d_keys.data_ptr(), rows.data_ptr(), st),
                       ctx.verify_rows_device(rows.data_ptr(), d_msgs.data_ptr(), msg_len, batch, res.data_ptr(), st))
    with torch.cuda.stream(side):
        both(side.cuda_stream)                                   # warm-up: workspaces
    side.synchronize()
    want = rows.clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        both(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        rows.zero_(); res.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(rows, want) and int(res.sum().item()) == batch
    one = pkg.Commitment(ctx, [int(x) for x in msgs[2]], int(seeds[2]))
    assert np.array_equal(one.as_words().view(np.int64), want[2].cpu().numpy())
    assert pkg.verify_opening_with_context(ctx, one, [int(x) for x in msgs[2]])
    assert np.array_equal(one.as_words(), oracle.lwe_commit(q, n, k, 3.19, 0x5EED, [int(x) for x in msgs[2]], int(seeds[2])))
    with pytest.raises(pkg.CoreError, match="not capturable"):
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=side):
            ctx.commit_keys_device(d_msgs.data_ptr(), msg_len, seeds, d_keys.data_ptr(), torch.cuda.current_stream().cuda_stream)
    ctx.close()


def test_chunk_lanes_are_graph_capturable(pkg):
    """n = 2^16: commitments and openings run their chunks on two lanes (the caller's stream and a side stream of the context, forked
    and joined with events).  Captured into a HIP graph the lanes become branches of the graph; 300 rank-1 rows = three chunks of
    commitments and two of openings.  Two replays: rows equal the eagerly computed ones, every opening succeeds."""
    import torch
    q, n, k, batch, msg_len = 17592182243329, 65536, 1, 300, 4
    ctx = pkg.LweContext(pkg.Params(q=q, n=n, k=k, sigma=3.19), key_seed=0xFACE)
    assert ctx.pipeline == "fused"
    rng = np.random.default_rng(12)
    msgs = rng.integers(0, ctx.plain_modulus, size=(batch, msg_len), dtype=np.uint64)
    seeds = rng.integers(1, 2**63, size=batch, dtype=np.uint64)
    keys = ctx.commit_keys(msgs, seeds)
    d_msgs = torch.from_numpy(msgs.view(np.int64)).cuda(); d_keys = torch.from_numpy(keys.view(np.int64)).cuda()
    rows = torch.zeros((batch, ctx.commitment_words), dtype=torch.int64, device="cuda")
    res = torch.zeros(batch, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    both = lambda st: (ctx.commit_rows_device(d_msgs.data_ptr(), msg_len, batch, d_keys.data_ptr(), rows.data_ptr(), st),
                       ctx.verify_rows_device(rows.data_ptr(), d_msgs.data_ptr(), msg_len, batch, res.data_ptr(), st))
    with torch.cuda.stream(side):
        both(side.cuda_stream)
    side.synchronize()
    want = rows.clone()
    assert int(res.sum().item()) == batch
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        both(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        rows.zero_(); res.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(rows, want) and int(res.sum().item()) == batch
    ctx.close()


def test_prover_handles_do_not_leak(pkg):
    import torch
    rng = np.random.default_rng(8)
    a, b, c = _gold_instances(rng, 1024, 4)

    def cycle():
        plan = pkg.QuotientPlan(1024)
        plan.quotient_batch(a, b, c)
        plan.close()
        t = pkg.CyclicNtt(4096)
        t.inverse(t.forward(a.reshape(-1)))
        t.close()
        prover = pkg.R1csProver(2, 6, [(0, 1, 1), (1, 3, 1)], [(0, 2, 1), (1, 4, 1)], [(0, 3, 1), (1, 5, 1)])
        prover.quotient_batch([1, 2, 3, 6, 4, 24])
        prover.close()

    for _ in range(3):
        cycle()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(30):
        cycle()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, f"device memory shrank by {(free0 - free1) >> 20} MiB over 30 create/use/free cycles"
