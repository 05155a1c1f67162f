"""GPU parity tests for the prover-side polynomial path (include/lambda_snark/prover.h), through the C-ABI.
First rust-api/lambda-snark/src/ntt.rs's own tests restated, then bit-exact parity with the oracle over every
size and arithmetic flavour, then the NTT-path quotient of r1cs.rs:474-506."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

Q = 18446744069414584321
Q44 = 17592169062401        # has 2^13-th roots of unity (r1cs.rs:534-547)
Q60 = 1152921504606584833   # = 1 mod 2^18


def root_of_order(q, n):
    """some primitive n-th root of unity mod prime q (n a power of two)"""
    g = 2
    while True:
        w = pow(g, (q - 1) // n, q)
        if n == 1 or pow(w, n // 2, q) == q - 1:
            return w
        g += 1


# ---- rust-api/lambda-snark/src/ntt.rs tests --------------------------------------------------------------------
def test_ntt_2_point(pkg):
    t = pkg.CyclicNtt(2)
    assert t.omega == Q - 1 == pkg.compute_root_of_unity(2)       # ntt.rs:266-270
    ev = t.forward([1, 2])
    assert list(ev) == [3, Q - 1]                                  # ntt.rs:293-296
    assert list(t.inverse(ev)) == [1, 2]


def test_ntt_4_and_8_point(pkg):
    t = pkg.CyclicNtt(4)
    ev = t.forward([1, 2, 3, 4])
    assert ev[0] == 10                                             # ntt.rs:312
    assert list(t.inverse(ev)) == [1, 2, 3, 4]
    t = pkg.CyclicNtt(8)
    ev = t.forward([1, 2, 3, 4, 5, 6, 7, 8])
    assert ev[0] == 36                                             # ntt.rs:327
    assert list(t.inverse(ev)) == [1, 2, 3, 4, 5, 6, 7, 8]


def test_ntt_inverse_correctness(pkg, oracle):
    for log_n in range(1, 11):                                     # ntt.rs:341-355
        n = 1 << log_n
        t = pkg.CyclicNtt(n)
        assert t.omega == oracle.prover_omega(n)
        coeffs = np.array([(i * 123456789) % Q for i in range(n)], dtype=np.uint64)
        ev = t.forward(coeffs)
        assert np.array_equal(ev, oracle.cyclic_forward(coeffs, Q, t.omega))
        assert np.array_equal(t.inverse(ev), coeffs)
        t.close()


def test_ntt_linearity(pkg):
    t = pkg.CyclicNtt(4)                                           # ntt.rs:357-389
    f, g, a, b = [1, 2, 3, 4], [5, 6, 7, 8], 3, 7
    combo = [(a * x + b * y) % Q for x, y in zip(f, g)]
    nf, ng = t.forward(f), t.forward(g)
    assert [int(v) for v in t.forward(combo)] == [(a * int(x) + b * int(y)) % Q for x, y in zip(nf, ng)]


# ---- parity with the oracle -------------------------------------------------------------------------------------
@pytest.mark.parametrize("log_n", list(range(1, 18)))
def test_goldilocks_parity_all_sizes(pkg, oracle, log_n):
    n = 1 << log_n
    batch = 5 if log_n <= 12 else 3
    t = pkg.CyclicNtt(n)
    rng = np.random.default_rng(100 + log_n)
    x = rng.integers(0, Q, size=(batch, n), dtype=np.uint64)
    x[0, 0] = Q - 1; x[0, 1] = 0; x[-1, -1] = Q - 1; x[1 % batch, :8] = Q - 1
    ev = t.forward(x)
    want = np.stack([oracle.cyclic_forward(row, Q, t.omega) for row in x])
    assert np.array_equal(ev, want)
    assert np.array_equal(t.inverse(ev), x)
    back = np.stack([oracle.cyclic_inverse(row, Q, t.omega) for row in x])
    assert np.array_equal(t.inverse(x), back)
    t.close()


def test_goldilocks_extreme_values(pkg, oracle):
    """all-(q-1) and 2^32-structured inputs stress the carry/borrow paths of the 2^64 = 2^32 - 1 reduction"""
    for n in [8, 4096, 65536]:
        t = pkg.CyclicNtt(n)
        rng = np.random.default_rng(n)
        pool = np.array([0, 1, Q - 1, Q - 2, 2**32, 2**32 - 1, 2**32 + 1, Q - 2**32, 2**63, 2**63 + 1, Q // 2, 0xFFFFFFFF00000000], dtype=np.uint64)
        x = np.stack([np.full(n, Q - 1, dtype=np.uint64), pool[rng.integers(0, pool.size, size=n)], pool[rng.integers(0, pool.size, size=n)]])
        ev = t.forward(x)
        assert np.array_equal(ev, np.stack([oracle.cyclic_forward(r, Q, t.omega) for r in x]))
        assert np.array_equal(t.inverse(ev), x)
        t.close()


@pytest.mark.parametrize("q,n", [(Q44, 8), (Q44, 256), (Q44, 8192), (Q60, 2), (Q60, 1024), (Q60, 131072), (12289, 4096), (Q, 64)])
def test_other_fields_and_explicit_roots(pkg, oracle, lib, q, n):
    """ntt_forward(coeffs, modulus, omega) takes any NTT-friendly field: both ordinary arithmetic flavours, caller's omega."""
    w = root_of_order(q, n)
    for mode in (0, 1):
        lib.lsr_set_arith_mode(mode)
        try:
            t = pkg.CyclicNtt(n, modulus=q, omega=w)
        finally:
            lib.lsr_set_arith_mode(0)
        assert t.omega == w and lib.lsr_ntt_context_is_cyclic(t.handle) == 1
        x = np.random.default_rng(n + mode).integers(0, q, size=(2, n), dtype=np.uint64)
        ev = t.forward(x)
        assert np.array_equal(ev, np.stack([oracle.cyclic_forward(r, q, w) for r in x]))
        assert np.array_equal(t.inverse(ev), x)
        t.close()


def test_native_order_and_device_api(pkg, oracle, lib):
    """A cyclic context driven through the ntt.h/batch.h entry points runs the network in its native order:
    forward output is bit-reversed; lsr_bit_reverse_device restores natural order on the device."""
    import torch
    n, batch = 16384, 4
    t = pkg.CyclicNtt(n)
    x = np.random.default_rng(3).integers(0, Q, size=(batch, n), dtype=np.uint64)
    dev = torch.from_numpy(x.view(np.int64)).cuda()
    out = torch.empty_like(dev)
    s = torch.cuda.current_stream().cuda_stream
    assert lib.lsr_ntt_forward_batch_device(t.handle, dev.data_ptr(), batch, s) == 0
    assert lib.lsr_bit_reverse_device(out.data_ptr(), dev.data_ptr(), 14, batch, s) == 0
    torch.cuda.synchronize()
    want = np.stack([oracle.cyclic_forward(r, Q, t.omega) for r in x])
    assert np.array_equal(out.cpu().numpy().view(np.uint64), want)
    rev = np.array([int(format(i, "014b")[::-1], 2) for i in range(n)])
    assert np.array_equal(dev.cpu().numpy().view(np.uint64), want[:, rev])
    # pointwise product in the field, then back: cyclic convolution theorem
    y = np.random.default_rng(4).integers(0, Q, size=(batch, n), dtype=np.uint64)
    dy = torch.from_numpy(y.view(np.int64)).cuda()
    assert lib.lsr_ntt_forward_batch_device(t.handle, dy.data_ptr(), batch, s) == 0
    assert lib.lsr_ntt_mul_pointwise_device(t.handle, dy.data_ptr(), dy.data_ptr(), dev.data_ptr(), batch * n, s) == 0
    assert lib.lsr_ntt_inverse_batch_device(t.handle, dy.data_ptr(), batch, s) == 0
    torch.cuda.synchronize()
    got = dy.cpu().numpy().view(np.uint64)
    fx, fy = want, np.stack([oracle.cyclic_forward(r, Q, t.omega) for r in y])
    prod = np.array([[int(a) * int(b) % Q for a, b in zip(r1, r2)] for r1, r2 in zip(fx[:1], fy[:1])], dtype=np.uint64)
    assert np.array_equal(got[0], oracle.cyclic_inverse(prod[0], Q, t.omega))
    assert lib.lsr_bit_reverse_device(dev.data_ptr(), dev.data_ptr(), 14, batch, s) == -1     # in place is refused
    assert lib.lsr_cyclic_ntt_forward_batch(pkg.NttContext(12289, 256).handle, x.ctypes.data, 1) == -1   # negacyclic context
    t.close()


# ---- quotient polynomial ------------------------------------------------------------------------------------------
def instances(rng, m, batch, spoil=()):
    a = rng.integers(0, Q, size=(batch, m), dtype=np.uint64)
    b = rng.integers(0, Q, size=(batch, m), dtype=np.uint64)
    c = np.array([[int(x) * int(y) % Q for x, y in zip(ra, rb)] for ra, rb in zip(a, b)], dtype=np.uint64)
    for i in spoil:
        j = int(rng.integers(0, m))
        c[i, j] = (int(c[i, j]) + 1 + int(rng.integers(0, 1000))) % Q
    return a, b, c


@pytest.mark.parametrize("m", [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192])
def test_quotient_matches_oracle(pkg, oracle, m):
    batch = 7 if m <= 256 else (3 if m <= 2048 else 2)
    rng = np.random.default_rng(900 + m)
    a, b, c = instances(rng, m, batch, spoil=(1,))
    if batch > 4:
        a[4] = 0; c[4] = 0                        # numerator identically zero -> [0]
        a[5] = 3; b[5] = 5; c[5] = 15             # constants: degree below m -> [0]
    plan = pkg.QuotientPlan(m)
    quot, lens = plan.quotient_batch(a, b, c)
    for i in range(batch):
        want, want_len = oracle.quotient(a[i], b[i], c[i])
        assert lens[i] == want_len, (m, i)
        if want_len:
            assert np.array_equal(quot[i], want), (m, i)
    assert lens[1] == 0
    if batch > 4:
        assert lens[4] == 1 and lens[5] == 1 and not quot[4].any() and not quot[5].any()
    # the reference's return convention
    assert np.array_equal(plan.compute_quotient_poly(a[0], b[0], c[0]), oracle.quotient(a[0], b[0], c[0])[0][:lens[0]])
    with pytest.raises(pkg.CoreError):
        plan.compute_quotient_poly(a[1], b[1], c[1])
    plan.close()


def test_quotient_multiplication_gates_like_r1cs_rs(pkg, oracle):
    """r1cs.rs:1685-1721 with the NTT modulus: witness (1,7,13,91) -> a=7, b=13, c=91; two gates 2*3=6, 6*4=24."""
    plan = pkg.QuotientPlan(1)
    q1 = plan.compute_quotient_poly([7], [13], [91])
    assert len(q1) <= 1 and all(int(v) < Q for v in q1)
    plan2 = pkg.QuotientPlan(2)
    q2 = plan2.compute_quotient_poly([2, 6], [3, 4], [6, 24])
    assert 1 <= len(q2) <= 2 and all(int(v) < Q for v in q2)
    assert np.array_equal(q2, oracle.quotient([2, 6], [3, 4], [6, 24])[0][:len(q2)])
    with pytest.raises(pkg.CoreError):
        plan2.compute_quotient_poly([2, 6], [3, 4], [6, 25])       # r1cs.rs:1316-1326
    plan.close(); plan2.close()


@pytest.mark.parametrize("m", [4096, 8192, 65536, 131072])
def test_quotient_identity_at_large_m(pkg, oracle, m):
    """Sizes beyond the O(m^2) oracle: the identity the reference tests (r1cs.rs:1723-1777),
    Q(alpha) (alpha^m - 1) = A(alpha) B(alpha) - C(alpha), at random alpha, with A, B, C interpolated by the oracle."""
    rng = np.random.default_rng(m)
    a, b, c = instances(rng, m, 3, spoil=(2,))
    plan = pkg.QuotientPlan(m)
    quot, lens = plan.quotient_batch(a, b, c)
    assert lens[2] == 0 and 1 <= lens[0] <= m and 1 <= lens[1] <= m
    w = oracle.prover_omega(m)
    for i in (0, 1):
        pa, pb, pc = (oracle.cyclic_inverse(v[i], Q, w) for v in (a, b, c))
        for alpha in [12345, int(rng.integers(0, Q, dtype=np.uint64))]:
            lhs = oracle.eval_poly(quot[i, :lens[i]], alpha, Q) * ((pow(alpha, m, Q) - 1) % Q) % Q
            rhs = (oracle.eval_poly(pa, alpha, Q) * oracle.eval_poly(pb, alpha, Q) - oracle.eval_poly(pc, alpha, Q)) % Q
            assert lhs == rhs
        assert not quot[i, lens[i]:].any()
    plan.close()


def test_quotient_device_api_and_chunking(pkg, oracle, monkeypatch):
    import torch
    m, batch = 64, 3000
    monkeypatch.setenv("LAMBDA_SNARK_QUOTIENT_CHUNK_LOG2", "16")      # 1024 instances per pass -> 3 passes
    rng = np.random.default_rng(77)
    a, b, c = instances(rng, m, batch, spoil=(5, 2999))
    plan = pkg.QuotientPlan(m, device=0)
    da, db, dc = (torch.from_numpy(v.view(np.int64)).cuda() for v in (a, b, c))
    dq = torch.empty_like(da)
    dl = torch.full((batch,), -1, dtype=torch.int32, device="cuda")
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        plan.quotient_device(da.data_ptr(), db.data_ptr(), dc.data_ptr(), batch, dq.data_ptr(), dl.data_ptr(), stream.cuda_stream)
    stream.synchronize()
    quot, lens = dq.cpu().numpy().view(np.uint64), dl.cpu().numpy().view(np.uint32)
    hq, hl = plan.quotient_batch(a, b, c)
    assert np.array_equal(lens, hl) and lens[5] == 0 and lens[2999] == 0 and (np.delete(lens, [5, 2999]) >= 1).all()
    ok = lens > 0
    assert np.array_equal(quot[ok], hq[ok])
    for i in [0, 1, 6, 1500, 2998]:
        want, ln = oracle.quotient(a[i], b[i], c[i])
        assert lens[i] == ln and np.array_equal(quot[i], want)
    plan.close()


def test_calls_on_one_plan_are_ordered_whatever_the_streams(pkg, oracle):
    """One plan, one workspace: asynchronous calls given DIFFERENT streams with no synchronisation between them, a host-API call
    right behind them and the plan freed with a call pending — every result equals that of the calls made one at a time."""
    import torch
    m, batch = 1024, 2048
    rng = np.random.default_rng(5150)
    sets = [instances(rng, m, batch, spoil=(i, batch - 1 - i)) for i in range(3)]
    plan = pkg.QuotientPlan(m, device=0)
    want = [plan.quotient_batch(a, b, c) for a, b, c in sets]
    dev = [tuple(torch.from_numpy(v.view(np.int64)).cuda() for v in abc) for abc in sets]
    streams = [torch.cuda.Stream() for _ in sets]
    for it in range(5):
        outs = [(torch.full((batch, m), -1, dtype=torch.int64, device="cuda"), torch.full((batch,), -1, dtype=torch.int32, device="cuda")) for _ in sets]
        torch.cuda.synchronize()
        for (da, db, dc), (dq, dl), st in zip(dev, outs, streams):
            plan.quotient_device(da.data_ptr(), db.data_ptr(), dc.data_ptr(), batch, dq.data_ptr(), dl.data_ptr(), st.cuda_stream)
        hq, hl = plan.quotient_batch(*sets[0])                       # host entry point, nothing synchronised before it
        torch.cuda.synchronize()
        assert np.array_equal(hl, want[0][1]) and np.array_equal(hq[hl > 0], want[0][0][hl > 0])
        for (dq, dl), (wq, wl) in zip(outs, want):
            lens = dl.cpu().numpy().view(np.uint32)
            assert np.array_equal(lens, wl), it
            assert np.array_equal(dq.cpu().numpy().view(np.uint64)[lens > 0], wq[wl > 0]), it
    dq = torch.full((batch, m), -1, dtype=torch.int64, device="cuda"); dl = torch.full((batch,), -1, dtype=torch.int32, device="cuda")
    da, db, dc = dev[1]
    plan.quotient_device(da.data_ptr(), db.data_ptr(), dc.data_ptr(), batch, dq.data_ptr(), dl.data_ptr(), streams[1].cuda_stream)
    plan.close()                                                      # lsr_quotient_plan_free with the call pending
    torch.cuda.synchronize()
    lens = dl.cpu().numpy().view(np.uint32)
    assert np.array_equal(lens, want[1][1]) and np.array_equal(dq.cpu().numpy().view(np.uint64)[lens > 0], want[1][0][lens > 0])


@pytest.mark.parametrize("m", [2, 64, 4096])
def test_quotient_fused_elementwise_stages_agree(pkg, oracle, m, monkeypatch):
    """m <= 4096: the a b = c test rides in the read-in of C's interpolation and the coset product in the read-in of the last
    transform (LAMBDA_SNARK_QUOTIENT_FUSE, default on).  Device API, words >= p among the inputs, spoiled instances: lengths and
    quotients equal those of the separate kernels and the oracle's."""
    import torch
    batch = 600 if m < 4096 else 40
    rng = np.random.default_rng(4242 + m)
    spoil = (0, 7, batch - 1)
    a, b, c = instances(rng, m, batch, spoil=spoil)
    for v in (a, b, c):                           # non-canonical representatives of small residues
        small = v < np.uint64(2**32 - 1)
        v[small] = v[small] + np.uint64(Q)
    da, db, dc = (torch.from_numpy(v.view(np.int64)).cuda() for v in (a, b, c))
    s = torch.cuda.current_stream().cuda_stream
    out = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("LAMBDA_SNARK_QUOTIENT_FUSE", fuse)     # read once, when the plan is created
        plan = pkg.QuotientPlan(m, device=0)
        monkeypatch.setenv("LAMBDA_SNARK_QUOTIENT_FUSE", "1" if fuse == "0" else "0")
        dq = torch.zeros_like(da)
        dl = torch.full((batch,), -1, dtype=torch.int32, device="cuda")
        plan.quotient_device(da.data_ptr(), db.data_ptr(), dc.data_ptr(), batch, dq.data_ptr(), dl.data_ptr(), s)
        torch.cuda.synchronize()
        out[fuse] = (dq.cpu().numpy().view(np.uint64), dl.cpu().numpy().view(np.uint32))
        plan.close()
    assert np.array_equal(out["1"][1], out["0"][1])
    lens = out["1"][1]
    assert all(lens[i] == 0 for i in spoil) and (np.delete(lens, list(spoil)) >= 1).all()
    ok = lens > 0
    assert np.array_equal(out["1"][0][ok], out["0"][0][ok])
    for i in (1, batch // 2, batch - 2):
        want, ln = oracle.quotient(a[i] % np.uint64(Q), b[i] % np.uint64(Q), c[i] % np.uint64(Q))
        assert lens[i] == ln and np.array_equal(out["1"][0][i], want)


@pytest.mark.parametrize("m", [8, 64, 256, 1024, 4096])
def test_quotient_degree_bound_of_every_instance_in_a_tile(pkg, oracle, m):
    """The trimmed length comes from the write-out of the last transform (one atomic per wavefront and instance at m >= 64, per
    word below): instances whose A, B have prescribed degrees d1 + d2 >= m give deg Q = d1 + d2 - m exactly (leading coefficients
    non-zero, C = A B on the domain — r1cs.rs:505-511 trims trailing zeros), laid out so that every position of a 4096-word tile, a
    tile boundary and a ragged last tile are hit; the highest non-zero index sits in every 64-word run in turn."""
    import torch
    batch = 2 * (4096 // m) + 3 if m < 4096 else 5
    rng = np.random.default_rng(31 + m)
    w = oracle.prover_omega(m)
    a = np.zeros((batch, m), dtype=np.uint64); b = np.zeros_like(a); want_len = np.zeros(batch, dtype=np.uint32)
    for i in range(batch):
        deg_q = (i * 37 + 5 * (i // 3)) % (m - 1)                      # 0 .. m-2, spread over the runs
        d1 = int(rng.integers(deg_q + 1, m))                            # d1 + d2 = m + deg_q, both below m
        d2 = m + deg_q - d1
        pa = np.zeros(m, dtype=np.uint64); pb = np.zeros(m, dtype=np.uint64)
        pa[:d1 + 1] = rng.integers(1, Q, d1 + 1, dtype=np.uint64); pb[:d2 + 1] = rng.integers(1, Q, d2 + 1, dtype=np.uint64)
        a[i], b[i] = oracle.cyclic_forward(pa, Q, w), oracle.cyclic_forward(pb, Q, w)
        want_len[i] = deg_q + 1
    c = np.array([[int(x) * int(y) % Q for x, y in zip(ra, rb)] for ra, rb in zip(a, b)], dtype=np.uint64)
    plan = pkg.QuotientPlan(m, device=0)
    quot, lens = plan.quotient_batch(a, b, c)
    assert np.array_equal(lens, want_len)
    for i in range(batch):
        assert quot[i, lens[i] - 1] != 0 and not quot[i, lens[i]:].any()
    for i in ([0, 1, batch // 2, batch - 1] if m <= 1024 else [0, batch - 1]):
        want, ln = oracle.quotient(a[i], b[i], c[i])
        assert ln == lens[i] and np.array_equal(quot[i], want)
    # the device entry point, with stale lengths in the output array
    da, db, dc = (torch.from_numpy(v.view(np.int64)).cuda() for v in (a, b, c))
    dq = torch.full_like(da, -1)
    dl = torch.full((batch,), -1, dtype=torch.int32, device="cuda")
    plan.quotient_device(da.data_ptr(), db.data_ptr(), dc.data_ptr(), batch, dq.data_ptr(), dl.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(dl.cpu().numpy().view(np.uint32), lens) and np.array_equal(dq.cpu().numpy().view(np.uint64), quot)
    plan.close()


# ---- compute_quotient_poly(witness) in full: sparse products + pipeline -------------------------------------------------
def random_r1cs(rng, m, free_vars, fan_in=3):
    """m constraints (A_i.z)(B_i.z) = z[free_vars + i] over free_vars + m variables; A_i, B_i touch earlier variables only,
    so any assignment of the free variables extends to exactly one satisfying witness."""
    n = free_vars + m
    a, b, c = [], [], []
    for i in range(m):
        for mat in (a, b):
            for col in rng.choice(free_vars + i, size=min(fan_in, free_vars + i), replace=False):
                mat.append((i, int(col), int(rng.integers(0, 2**64, dtype=np.uint64))))     # values >= q exercise `val % modulus`
        c.append((i, free_vars + i, 1))
    return n, a, b, c


def extend_witness(free, m, a, b):
    z = [int(x) % Q for x in free] + [0] * m
    rows_a, rows_b = [[] for _ in range(m)], [[] for _ in range(m)]
    for (i, col, v) in a: rows_a[i].append((col, v % Q))
    for (i, col, v) in b: rows_b[i].append((col, v % Q))
    for i in range(m):
        az = sum(v * z[col] for col, v in rows_a[i]) % Q
        bz = sum(v * z[col] for col, v in rows_b[i]) % Q
        z[len(free) + i] = az * bz % Q
    return np.array(z, dtype=np.uint64)


@pytest.mark.parametrize("m,free_vars", [(1, 3), (2, 4), (8, 5), (64, 16), (1024, 40)])
def test_r1cs_prover_matches_oracle(pkg, oracle, m, free_vars):
    rng = np.random.default_rng(4000 + m)
    n, a, b, c = random_r1cs(rng, m, free_vars)
    a.append(a[0])                                                  # duplicate coordinate entries add up
    prover = pkg.R1csProver(m, n, a, b, c)
    batch = 5
    ws = np.stack([extend_witness(rng.integers(0, Q, size=free_vars, dtype=np.uint64), m, a, b) for _ in range(batch)])
    ws[3, n - 1] = (int(ws[3, n - 1]) + 1) % Q                      # break the last constraint of witness 3
    ws[4, 0] = np.uint64(int(ws[4, 0]) + Q if int(ws[4, 0]) + Q < 2**64 else ws[4, 0])   # a non-canonical word: `v[col] % modulus`
    ea, eb, ec = prover.compute_constraint_evals(ws)
    for i in range(batch):
        assert np.array_equal(ea[i], oracle.sparse_mul_vec(a, m, ws[i], Q))      # r1cs.rs:296-304
        assert np.array_equal(eb[i], oracle.sparse_mul_vec(b, m, ws[i], Q))
        assert np.array_equal(ec[i], oracle.sparse_mul_vec(c, m, ws[i], Q))
    quot, lens = prover.quotient_batch(ws)
    for i in range(batch):
        want, want_len = oracle.quotient(ea[i], eb[i], ec[i])
        assert lens[i] == want_len
        if want_len:
            assert np.array_equal(quot[i], want)
    assert lens[3] == 0 and all(lens[i] >= 1 for i in (0, 1, 2, 4))
    assert np.array_equal(prover.compute_quotient_poly(ws[0]), quot[0, :lens[0]])
    with pytest.raises(pkg.CoreError):
        prover.compute_quotient_poly(ws[3])
    prover.close()


def test_r1cs_prover_reference_gates_and_errors(pkg, lib):
    """create_multiplication_gate / create_two_multiplications of r1cs.rs:1071-1113 over the NTT modulus."""
    gate = pkg.R1csProver(1, 4, [(0, 1, 1)], [(0, 2, 1)], [(0, 3, 1)])
    ea, eb, ec = gate.compute_constraint_evals([1, 7, 13, 91])
    assert (int(ea[0, 0]), int(eb[0, 0]), int(ec[0, 0])) == (7, 13, 91)                        # r1cs.rs:1266-1281
    assert len(gate.compute_quotient_poly([1, 7, 13, 91])) <= 1
    with pytest.raises(pkg.CoreError):
        gate.compute_quotient_poly([1, 7, 13, 90])                                             # r1cs.rs:1316-1326
    two = pkg.R1csProver(2, 6, [(0, 1, 1), (1, 3, 1)], [(0, 2, 1), (1, 4, 1)], [(0, 3, 1), (1, 5, 1)])
    q2 = two.compute_quotient_poly([1, 2, 3, 6, 4, 24])
    assert 1 <= len(q2) <= 2
    with pytest.raises(pkg.CoreError):
        two.compute_quotient_poly([1, 2, 3, 7, 4, 24])
    with pytest.raises(ValueError):
        two.compute_quotient_poly([1, 2, 3])
    with pytest.raises(pkg.CoreError):
        pkg.R1csProver(3, 4, [], [], [])                     # m not a power of two
    with pytest.raises(pkg.CoreError):
        pkg.R1csProver(2, 4, [(2, 0, 1)], [], [])            # row outside the matrix
    empty = pkg.R1csProver(4, 3, [], [], [])                 # all-zero matrices: 0 * 0 = 0 holds, Q = [0]
    assert list(empty.compute_quotient_poly([5, 6, 7])) == [0]
    assert lib.lsr_r1cs_prover_num_constraints(empty._h) == 4 and lib.lsr_r1cs_prover_num_variables(empty._h) == 3
    gate.close(); two.close(); empty.close()


def test_batched_prove_chain_matches_the_one_by_one_replay(pkg, lib, oracle):
    """prove_r1cs (lib.rs:747-809) for a batch of witnesses of one circuit on the NTT path, chained through the batched entry
    points — lsr_r1cs_quotient_batch -> lsr_lwe_commit_batch_flat -> lsr_fs_challenge_batch_flat (alpha, then beta) ->
    lsr_lwe_verify_opening_batch_flat — against the same steps taken one proof at a time with the single-call ABI."""
    import prover_replay
    rng = np.random.default_rng(2468)
    m, free_vars, batch, n_public = 64, 10, 9, 3
    n, a, b, c = random_r1cs(rng, m, free_vars)
    ws = np.stack([extend_witness(rng.integers(0, Q, size=free_vars, dtype=np.uint64), m, a, b) for _ in range(batch)])
    params = pkg.Params(q=17592186044417, n=4096, k=2, sigma=3.19)
    ctx = pkg.LweContext(params, key_seed=0x5EED)
    seeds = np.arange(1, batch + 1, dtype=np.uint64) * np.uint64(7919)
    # --- batched chain ---
    prover = pkg.R1csProver(m, n, a, b, c)
    quot, lens = prover.quotient_batch(ws)
    assert (lens >= 1).all()
    msgs = quot % np.uint64(params.q)            # Commitment::new reduces mod ctx.modulus() (commitment.rs:33-36)
    rows = pkg.Commitment.batch_words(ctx, msgs, seeds)
    publics = np.ascontiguousarray(ws[:, :n_public])
    alphas = np.zeros(batch, dtype=np.uint64); betas = np.zeros(batch, dtype=np.uint64)
    ha = np.zeros((batch, 32), dtype=np.uint8); hb = np.zeros((batch, 32), dtype=np.uint8)
    W = rows.shape[1]
    assert lib.lsr_fs_challenge_batch_flat(publics.ctypes.data, n_public, rows.ctypes.data, W, batch, Q, alphas.ctypes.data, ha.ctypes.data, 0) == 0
    assert lib.lsr_fs_challenge_batch_flat(alphas.ctypes.data, 1, rows.ctypes.data, W, batch, Q, betas.ctypes.data, hb.ctypes.data, 0) == 0
    # Field coefficients are wider than the plaintext modulus: they are embedded mod t and — exactly as with the reference's
    # decode-and-compare (commitment.cpp:223-226) — open to their residues, never to the wide words themselves.
    t = np.uint64(ctx.plain_modulus)
    assert pkg.verify_openings_words(ctx, rows, msgs % t) == [1] * batch
    assert pkg.verify_openings_words(ctx, rows, msgs) == [int(bool((msgs[i] < t).all())) for i in range(batch)]
    # full binding of the wide coefficients: commit to their 16-bit limbs (lsr_words_to_limbs)
    limbs = pkg.words_to_limbs(msgs).reshape(batch, -1)
    assert limbs.shape[1] == 4 * msgs.shape[1] and int(limbs.max()) < 2**16
    limb_rows = pkg.Commitment.batch_words(ctx, limbs, seeds)
    assert pkg.verify_openings_words(ctx, limb_rows, limbs) == [1] * batch
    tampered = msgs.copy(); tampered[0, 0] ^= np.uint64(1 << 40)          # a change the residue mod t might miss
    assert pkg.verify_openings_words(ctx, limb_rows[:1], pkg.words_to_limbs(tampered[:1]).reshape(1, -1)) == [0]
    # --- one proof at a time ---
    for i in range(batch):
        ea, eb, ec = (oracle.sparse_mul_vec(mat, m, ws[i], Q) for mat in (a, b, c))
        q_coeffs, ln = oracle.quotient(ea, eb, ec)
        assert ln == lens[i] and np.array_equal(q_coeffs, quot[i])
        com = pkg.Commitment(ctx, q_coeffs[:ln], int(seeds[i]))                           # trimmed message: the zero tail encodes the same
        assert np.array_equal(com.as_words(), rows[i])
        alpha, h_alpha = prover_replay.challenge_derive([int(x) for x in publics[i]], com.as_words(), Q)
        beta, h_beta = prover_replay.challenge_derive([alpha], com.as_words(), Q)
        assert (int(alphas[i]), bytes(ha[i]), int(betas[i]), bytes(hb[i])) == (alpha, h_alpha, beta, h_beta)
        # verify_r1cs's check (lib.rs:1016-1095) at alpha with the interpolants of the NTT domain
        w = oracle.prover_omega(m)
        pa, pb, pc = (oracle.cyclic_inverse(v, Q, w) for v in (ea, eb, ec))
        lhs = oracle.eval_poly(q_coeffs[:ln], alpha, Q) * ((pow(alpha, m, Q) - 1) % Q) % Q
        rhs = (oracle.eval_poly(pa, alpha, Q) * oracle.eval_poly(pb, alpha, Q) - oracle.eval_poly(pc, alpha, Q)) % Q
        assert lhs == rhs
        com.free()
    prover.close(); ctx.close()
