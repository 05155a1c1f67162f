"""GPU parity tests for the sampler and the Module-LWE commitment, through the C-ABI.  First the reference's
own assertions (cpp-core/tests/test_commitment.cpp, test_utils.cpp, the Rust unit test
rust-api/lambda-snark/src/commitment.rs:163-219), then bit-exactness against the CPU oracle under the same
seeds (the reference itself is non-deterministic: commitment.cpp:142), then the BASELINE configs 1, 3 and 5."""
import ctypes
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEY = 0xABCDEF


@pytest.fixture(scope="module")
def ctx(pkg):
    # test_commitment.cpp:12-20: RING_B, 128, q = 12289, n = 4096, k = 2, sigma = 3.19
    c = pkg.LweContext(pkg.Params(q=12289, n=4096, k=2, sigma=3.19), key_seed=KEY)
    yield c
    c.close()


def raw_commit(lib, ctx, msg, seed):
    m = np.array(msg, dtype=np.uint64)
    return lib.lwe_commit(ctx.handle, m.ctypes.data, m.size, seed)


def words(p):
    return np.ctypeslib.as_array(p.contents.data, shape=(p.contents.len,)).copy()


# ---- cpp-core/tests/test_commitment.cpp ---------------------------------------------------------------
def test_create_and_free(ctx):
    assert ctx.handle
    assert ctx.commit_modulus == 17592169062401 and ctx.plain_modulus == 1032193
    assert ctx.ring_degree == 4096 and ctx.module_rank == 2


def test_commit_basic(lib, ctx):
    comm = raw_commit(lib, ctx, [1, 2, 3, 4], 0x1234)                         # test_commitment.cpp:37-47
    assert comm and comm.contents.len > 0 and comm.contents.data
    assert comm.contents.data[0] == 8 * (comm.contents.len - 1)               # commitment.cpp:44-60 framing
    lib.lwe_commitment_free(comm)


def test_commit_binding_fresh_randomness(lib, ctx):
    c1, c2 = raw_commit(lib, ctx, [1, 2, 3], 0), raw_commit(lib, ctx, [4, 5, 6], 0)   # test_commitment.cpp:49-75
    assert c1 and c2 and c1.contents.len > 0 and c2.contents.len > 0
    assert not np.array_equal(words(c1), words(c2))
    c3 = raw_commit(lib, ctx, [1, 2, 3], 0)                                   # seed 0 = random (commitment.h:52)
    assert not np.array_equal(words(c1), words(c3))
    assert c1.contents.len == c3.contents.len                                 # polynomial_commitment.rs:72-98 (<= 32 words skew)
    for c in (c1, c2, c3):
        lib.lwe_commitment_free(c)


def test_commit_different_messages_same_seed(lib, ctx):
    c1, c2 = raw_commit(lib, ctx, [1, 2, 3], 0x1234), raw_commit(lib, ctx, [4, 5, 6], 0x1234)   # test_commitment.cpp:77-100
    assert not np.array_equal(words(c1), words(c2))
    lib.lwe_commitment_free(c1); lib.lwe_commitment_free(c2)


def test_null_pointer_handling(lib, ctx):
    assert not lib.lwe_commit(None, None, 0, 0)                               # test_commitment.cpp:102-113
    assert not lib.lwe_commit(ctx.handle, None, 10, 0)
    lib.lwe_commitment_free(None)


def test_verify_opening_matches_message(lib, ctx, pkg):
    msg = np.array([7, 11, 13, 17], dtype=np.uint64)
    comm = lib.lwe_commit(ctx.handle, msg.ctypes.data, 4, 0)
    rnd = np.zeros(1, dtype=np.uint64)
    opening = pkg._abi.LweOpening(rnd.ctypes.data_as(pkg._abi.u64p), 1)
    assert lib.lwe_verify_opening(ctx.handle, comm, msg.ctypes.data, 4, ctypes.byref(opening)) == 1   # test_commitment.cpp:115-132
    wrong = msg.copy(); wrong[1] ^= 1
    assert lib.lwe_verify_opening(ctx.handle, comm, wrong.ctypes.data, 4, ctypes.byref(opening)) == 0
    assert lib.lwe_verify_opening(ctx.handle, comm, msg.ctypes.data, 4, None) == 1                    # opening ignored (commitment.cpp:205)
    lib.lwe_commitment_free(comm)


def test_linear_combination_produces_expected_commitment(lib, ctx, pkg):
    m1, m2 = np.array([1, 2, 3, 4], np.uint64), np.array([5, 6, 7, 8], np.uint64)
    c1, c2 = lib.lwe_commit(ctx.handle, m1.ctypes.data, 4, 0), lib.lwe_commit(ctx.handle, m2.ctypes.data, 4, 0)
    arr = (ctypes.POINTER(pkg._abi.LweCommitment) * 2)(c1, c2)
    coeffs = np.array([2, 3], np.uint64)
    comb = lib.lwe_linear_combine(ctx.handle, arr, coeffs.ctypes.data, 2)     # test_commitment.cpp:134-166
    assert comb
    expected = 2 * m1 + 3 * m2
    assert lib.lwe_verify_opening(ctx.handle, comb, expected.ctypes.data, 4, None) == 1
    expected[0] += 1
    assert lib.lwe_verify_opening(ctx.handle, comb, expected.ctypes.data, 4, None) == 0
    # commitment.cpp:248-250,268-270: NULL entries are skipped; nothing to combine => NULL
    arr2 = (ctypes.POINTER(pkg._abi.LweCommitment) * 2)(None, c2)
    only2 = lib.lwe_linear_combine(ctx.handle, arr2, coeffs.ctypes.data, 2)
    assert lib.lwe_verify_opening(ctx.handle, only2, (3 * m2).ctypes.data, 4, None) == 1
    none = (ctypes.POINTER(pkg._abi.LweCommitment) * 2)(None, None)
    assert not lib.lwe_linear_combine(ctx.handle, none, coeffs.ctypes.data, 2)
    assert not lib.lwe_linear_combine(ctx.handle, arr, coeffs.ctypes.data, 0)
    for c in (c1, c2, comb, only2):
        lib.lwe_commitment_free(c)


def test_clone_and_framing_errors(lib, ctx, pkg):
    comm = raw_commit(lib, ctx, [9, 8, 7], 5)
    clone = lib.lwe_commitment_clone(comm)                                    # commitment.cpp:179-198
    assert clone and np.array_equal(words(clone), words(comm)) and ctypes.addressof(clone.contents) != ctypes.addressof(comm.contents)
    msg = np.array([9, 8, 7], np.uint64)
    assert lib.lwe_verify_opening(ctx.handle, clone, msg.ctypes.data, 3, None) == 1
    clone.contents.data[0] = 0                                                # commitment.cpp:71-75: byte_len == 0
    assert lib.lwe_verify_opening(ctx.handle, clone, msg.ctypes.data, 3, None) == -1
    clone.contents.data[0] = 8 * clone.contents.len                           # byte_len > available
    assert lib.lwe_verify_opening(ctx.handle, clone, msg.ctypes.data, 3, None) == -1
    too_long = np.zeros(4097, np.uint64)
    assert lib.lwe_verify_opening(ctx.handle, comm, too_long.ctypes.data, 4097, None) == 0   # commitment.cpp:219-221
    lib.lwe_commitment_free(clone); lib.lwe_commitment_free(comm)


def test_rust_wrapper_flow(pkg, ctx):
    """rust-api/lambda-snark/src/commitment.rs:163-219 (linear_combine + verify with rand_len = 0) and
    tests/lwe_verification.rs:72-94 (wrong polynomial rejected), via the mirrored safe wrappers."""
    rctx = pkg.LweContext(pkg.Params(q=17592186044417, n=4096, k=2, sigma=3.19))      # every Rust call site
    assert rctx.modulus() == 17592186044417                                            # context.rs:90
    c1 = pkg.Commitment(rctx, [1, 2, 3, 4], 0x1111)
    c2 = pkg.Commitment(rctx, [5, 6, 7, 8], 0x2222)
    comb = pkg.Commitment.linear_combine(rctx, [c1, c2], [2, 3])
    assert pkg.verify_opening_with_context(rctx, comb, [17, 22, 27, 32], randomness=np.zeros(0, np.uint64))
    assert not pkg.verify_opening_with_context(rctx, comb, [18, 22, 27, 32])
    assert not pkg.verify_opening_with_context(rctx, c1, [1, 2, 3, 5])
    twin = c1.clone()
    assert np.array_equal(twin.as_words(), c1.as_words()) and len(twin) == len(c1)
    # tests/serialization.rs:128-153: bincode = 8-byte element count + the words; data[0] = payload byte length (types.h:30-35)
    blob = c1.serialize()
    assert len(blob) == 8 + len(c1) * 8 and int.from_bytes(blob[:8], "little") == len(c1)
    assert int.from_bytes(blob[8:16], "little") == (len(c1) - 1) * 8 and blob[8:] == c1.as_bytes()
    with pytest.raises(ValueError):
        pkg.Commitment.linear_combine(rctx, [], [])
    # ScalarA profile is sent as ring_degree = 1 and rejected, as by the reference
    with pytest.raises(pkg.CoreError):
        pkg.LweContext(pkg.Params(profile=pkg.PROFILE_SCALAR_A))
    rctx.close()


# ---- sampler: cpp-core/tests/test_utils.cpp -------------------------------------------------------------
def test_sample_gaussian_rejects_invalid_inputs(lib):
    buf = np.zeros(16, np.uint64)
    assert lib.sample_gaussian(None, 16, 3.2) == -1                           # test_utils.cpp:26-33
    assert lib.sample_gaussian(buf.ctypes.data, 0, 3.2) == -1
    assert lib.sample_gaussian(buf.ctypes.data, 16, 0.0) == -1
    assert lib.sample_gaussian(buf.ctypes.data, 16, float("inf")) == -1


def test_sample_gaussian_moments(pkg):
    v = pkg.sample_gaussian(4096, 3.2).astype(np.float64)                     # test_utils.cpp:35-70
    pos, neg = int((v > 0).sum()), int((v < 0).sum())
    assert abs(v.mean()) < 0.5 and abs(v.std(ddof=1) - 3.2) < 0.8
    assert pos > 1024 and neg > 1024 and abs(pos - neg) < 4096 // 5
    assert not np.array_equal(pkg.sample_gaussian(64, 3.2), pkg.sample_gaussian(64, 3.2))   # fresh entropy per call


@pytest.mark.parametrize("sigma,length", [(3.19, 4096), (3.2, 1001), (0.4, 77), (20.0, 4099), (100.0, 513), (3.45, 4096), (5.0, 4100), (6.9, 999), (7.1, 640)])
def test_seeded_sampler_bit_exact(pkg, oracle, sigma, length):
    got = pkg.sample_gaussian(length, sigma, seed=0xFEED, domain=5, index=3)
    assert np.array_equal(got, oracle.sample_gaussian_seeded(length, sigma, 0xFEED, 5, 3))
    assert np.array_equal(got, pkg.sample_gaussian(length, sigma, seed=0xFEED, domain=5, index=3))
    assert not np.array_equal(got, pkg.sample_gaussian(length, sigma, seed=0xFEED, domain=5, index=4))


# ---- bit-exactness against the oracle -------------------------------------------------------------------
@pytest.mark.parametrize("q,n,k", [(12289, 4096, 2), (17592186044417, 4096, 2), (12289, 256, 2), (17592186044417, 1024, 4),
                                   (1152921504606584833, 4096, 1), (12289, 8192, 3), (12289, 512, 5), (1152921504606584833, 256, 7),
                                   (12289, 2, 1), (12289, 64, 16), (12289, 131072, 1), ("wide", 65536, 2), ("wide", 1024, 3)])
def test_commit_bit_exact_vs_oracle(pkg, oracle, q, n, k):
    if q == "wide":                                            # the 60-bit prime the library names for reference-range combinations
        q = pkg.wide_modulus(n)
        assert q == oracle.L.oracle_largest_prime_1mod(2 * n, 60)
    lctx = pkg.LweContext(pkg.Params(q=q, n=n, k=k, sigma=3.19), key_seed=KEY)
    oq, a_hat = oracle.lwe_public_matrix(q, n, k, 3.19, KEY)
    assert lctx.commit_modulus == oq
    assert np.array_equal(lctx.public_matrix(), a_hat)
    for seed, msg in [(0x5678, [1, 314, 628, 471, 471]), (1, list(range(n))), (2**64 - 1, [2**63, 12345678901234567, 0]), (7, list(range(n + 5)))]:
        want = oracle.lwe_commit(q, n, k, 3.19, KEY, msg, seed)
        m = np.array(msg, dtype=np.uint64)
        p = lctx._lib.lwe_commit(lctx.handle, m.ctypes.data, m.size, seed)
        got = words(p)
        assert np.array_equal(got, want)
        shown = msg[:n]
        t = lctx.plain_modulus
        opens = 1 if all(x < t for x in shown) else 0          # words >= t are embedded mod t and never open as given (commitment.cpp:223-226)
        assert lctx._lib.lwe_verify_opening(lctx.handle, p, np.array(shown, np.uint64).ctypes.data, len(shown), None) == opens
        assert oracle.lwe_verify(q, n, k, 3.19, KEY, got, shown) == opens
        canonical = [x % t for x in shown]
        assert lctx._lib.lwe_verify_opening(lctx.handle, p, np.array(canonical, np.uint64).ctypes.data, len(canonical), None) == 1
        assert oracle.lwe_verify(q, n, k, 3.19, KEY, got, canonical) == 1
        lctx._lib.lwe_commitment_free(p)
    lctx.close()


def test_linear_combine_bit_exact_vs_oracle(pkg, oracle):
    q, n, k = 17592186044417, 4096, 2
    lctx = pkg.LweContext(pkg.Params(q=q, n=n, k=k, sigma=3.19), key_seed=KEY)
    msgs = [[1, 2, 3, 4], [5, 6, 7, 8], [9, 10, 11, 12]]
    coms = [pkg.Commitment(lctx, m, seed=100 + i) for i, m in enumerate(msgs)]
    coeffs = [2, 3, 1032193 + 5]                                             # reduced mod t (commitment.cpp:90)
    comb = pkg.Commitment.linear_combine(lctx, coms, coeffs)
    rc, want = oracle.lwe_linear_combine(q, n, k, 3.19, KEY, [c.as_words() for c in coms], coeffs)
    assert rc == 0 and np.array_equal(comb.as_words(), want)
    assert pkg.verify_opening_with_context(lctx, comb, [2 * a + 3 * b + 5 * c for a, b, c in zip(*msgs)])
    lctx.close()


def test_combinations_of_large_messages_and_negative_coefficients(pkg, oracle):
    """Round 3: (1) messages near t with sum c_i in the hundreds — the message rides as round(q m / t), so the integer overflow of
    sum c_i m_i past t leaves no (q mod t)-sized noise (with floor(q/t) m such combinations passed the noise budget and failed to
    open); (2) coefficients in (t/2, t) act as small negative numbers: a subtraction (t - 1) is inside the budget.  Words equal the
    oracle's, the combination opens to sum c_i m_i mod t, and the budget still refuses what cannot open."""
    q, n, k = 17592169062401, 4096, 2
    lctx = pkg.LweContext(pkg.Params(q=q, n=n, k=k, sigma=3.19), key_seed=KEY)
    t = lctx.plain_modulus
    msgs = [[t - 1] * n, [t - 2] * n, list(range(t - n, t))]
    coms = [pkg.Commitment(lctx, m, seed=500 + i) for i, m in enumerate(msgs)]
    for j, m in enumerate(msgs):
        assert np.array_equal(coms[j].as_words(), oracle.lwe_commit(q, n, k, 3.19, KEY, m, 500 + j))
        assert pkg.verify_opening_with_context(lctx, coms[j], m)
    for coeffs in ([100, 100, 0], [300, 200, 250], [t - 1, 1, 0], [t - 3, t - 400, 7], [1, t - 1, t - 1]):
        comb = pkg.Commitment.linear_combine(lctx, coms, coeffs)
        rc, want = oracle.lwe_linear_combine(q, n, k, 3.19, KEY, [c.as_words() for c in coms], coeffs)
        assert rc == 0 and np.array_equal(comb.as_words(), want), coeffs
        expect = [sum(c * m[i] for c, m in zip(coeffs, msgs)) % t for i in range(n)]
        assert pkg.verify_opening_with_context(lctx, comb, expect), coeffs
        assert oracle.lwe_verify(q, n, k, 3.19, KEY, comb.as_words(), expect) == 1
    with pytest.raises(pkg.CoreError):
        pkg.Commitment.linear_combine(lctx, coms, [500, 400, 0])          # sum |c_i| = 900 > ~800: refused, not returned
    assert oracle.lwe_linear_combine(q, n, k, 3.19, KEY, [c.as_words() for c in coms], [500, 400, 0])[0] == -1
    lctx.close()


def test_commit_batch_matches_single_calls(pkg, oracle):
    q, n, k = 17592186044417, 4096, 2
    lctx = pkg.LweContext(pkg.Params(q=q, n=n, k=k, sigma=3.19), key_seed=KEY)
    batch, msg_len = 9, 6
    msgs = (np.arange(batch * msg_len, dtype=np.uint64).reshape(batch, msg_len) * 7919) % 1000003
    seeds = np.array([(j * 0x9E3779B97F4A7C15) % 2**64 for j in range(1, batch + 1)], dtype=np.uint64)
    coms = pkg.Commitment.batch(lctx, msgs, seeds)
    for j in range(batch):
        assert np.array_equal(coms[j].as_words(), oracle.lwe_commit(q, n, k, 3.19, KEY, [int(x) for x in msgs[j]], int(seeds[j])))
    lctx.close()


# ---- BASELINE configs -----------------------------------------------------------------------------------
def test_commit_batch_flat_equals_per_commitment_form(pkg, oracle, lib, monkeypatch):
    """lsr_lwe_commit_batch_flat: row i = the words of lwe_commit_batch's out[i], also across several device passes."""
    q, n, k, key = 17592169062401, 1024, 3, 0xABCDE
    ctx = pkg.LweContext(pkg.Params(q=q, n=n, k=k, sigma=3.19), key_seed=key)
    rng = np.random.default_rng(31)
    batch = 37
    msgs = rng.integers(0, ctx.plain_modulus, size=(batch, 9), dtype=np.uint64)
    seeds = rng.integers(1, 2**62, size=batch, dtype=np.uint64)
    flat = pkg.Commitment.batch_words(ctx, msgs, seeds)
    assert flat.shape == (batch, 5 + (k + 1) * n)
    coms = pkg.Commitment.batch(ctx, msgs, seeds)
    for i in range(batch):
        assert np.array_equal(flat[i], coms[i].as_words())
        assert np.array_equal(flat[i], oracle.lwe_commit(q, n, k, 3.19, key, msgs[i], int(seeds[i])))
        assert flat[i, 0] == 8 * (flat.shape[1] - 1)
    # a row is a valid commitment for the rest of the ABI
    row = np.ascontiguousarray(flat[5])
    as_struct = pkg._abi.LweCommitment(row.ctypes.data_as(pkg._abi.u64p), row.size)
    assert lib.lwe_verify_opening(ctx.handle, ctypes.byref(as_struct), msgs[5].ctypes.data, 9, None) == 1
    # batched verification straight from the flat rows: a wrong message, a corrupted header and a non-canonical residue
    res = np.full(batch, 7, dtype=np.int32)
    wrong = msgs.copy(); wrong[3, 0] += 1
    rows = flat.copy(); rows[4, 1] ^= 1; rows[6, 10] = q
    assert lib.lsr_lwe_verify_opening_batch_flat(ctx.handle, rows.ctypes.data, wrong.ctypes.data, 9, batch, res.ctypes.data) == 0
    want = np.ones(batch, dtype=np.int32); want[3] = 0; want[4] = -1; want[6] = -1
    assert np.array_equal(res, want)
    assert pkg.verify_openings_batch(ctx, coms, wrong) == [1, 1, 1, 0] + [1] * (batch - 4)
    assert pkg.verify_openings_words(ctx, rows, wrong) == [int(x) for x in want]
    # message lengths decided by the screening alone (commitment.cpp:207-214): empty message -> 1, longer than n -> 0, bad rows -> -1
    assert lib.lsr_lwe_verify_opening_batch_flat(ctx.handle, rows.ctypes.data, wrong.ctypes.data, 0, batch, res.ctypes.data) == 0
    want0 = np.ones(batch, dtype=np.int32); want0[4] = -1; want0[6] = -1
    assert np.array_equal(res, want0)
    longer = np.zeros((batch, n + 1), dtype=np.uint64)
    assert lib.lsr_lwe_verify_opening_batch_flat(ctx.handle, rows.ctypes.data, longer.ctypes.data, n + 1, batch, res.ctypes.data) == 0
    wantl = np.zeros(batch, dtype=np.int32); wantl[4] = -1; wantl[6] = -1
    assert np.array_equal(res, wantl)
    assert lib.lsr_lwe_verify_opening_batch_flat(ctx.handle, None, wrong.ctypes.data, 9, batch, res.ctypes.data) == -1
    assert lib.lsr_lwe_commit_batch_flat(ctx.handle, msgs.ctypes.data, 9, 0, seeds.ctypes.data, flat.ctypes.data) == 0
    assert lib.lsr_lwe_commit_batch_flat(None, msgs.ctypes.data, 9, 1, seeds.ctypes.data, flat.ctypes.data) == -1
    assert lib.lsr_lwe_commit_batch_flat(ctx.handle, msgs.ctypes.data, 9, 1, seeds.ctypes.data, None) == -1
    for c in coms: c.free()
    ctx.close()


def test_config1_tv0_linear_system_plumbing(pkg, oracle, golden_dir):
    """Config 1: lwe_context_create{RING_B,128,q,4096,2,3.19} -> statement/witness of
    TV-0 -> one lwe_commit -> lwe_verify_opening == 1.  (The published TV-0 data is itself inconsistent:
    A z != b for z = [1..5]; the reference's conformance test never evaluates it — test_conformance.cpp
    only scrapes strings.)"""
    params = json.load(open(os.path.join(golden_dir, "tv0_params.json")))
    witness = json.load(open(os.path.join(golden_dir, "tv0_witness.json")))
    expected = json.load(open(os.path.join(golden_dir, "tv0_expected.json")))
    prof = params["profile"]
    assert (prof["n"], prof["k"], prof["q"], prof["sigma"]) == (4096, 2, 17592186044417, 3.19)
    a = np.array(params["statement"]["matrix_A"], dtype=object)
    z = witness["z"]
    az = [int(sum(a[i][j] * z[j] for j in range(5)) % prof["q"]) for i in range(5)]
    assert az == [4, 10, 18, 28, 34] and az != params["statement"]["vector_b"]   # the fixture's own claim does not hold
    lctx = pkg.LweContext(pkg.Params(security_level=params["security_level"], q=prof["q"], n=prof["n"], k=prof["k"], sigma=prof["sigma"]),
                          key_seed=int(params["random_seed"], 16))
    com = pkg.Commitment(lctx, z, seed=int(params["random_seed"], 16))
    assert pkg.verify_opening_with_context(lctx, com, z) is expected["valid"]
    assert np.array_equal(com.as_words(), oracle.lwe_commit(prof["q"], 4096, 2, 3.19, 0xDEADBEEF, z, 0xDEADBEEF))
    lctx.close()


def test_config5_tv2_plaquette_commitments(pkg, oracle, golden_dir):
    """Config 5: the two commitment shapes the Rust prover makes on TV-2 — prove_simple commits to the witness
    (tests/prover.rs:78-119: [1,314,628,471,471], seed 0x5678), prove_r1cs to the constant quotient Q(X)
    (m = 1) — GPU words bit-exact vs the CPU oracle under the same seeds."""
    cons = json.load(open(os.path.join(golden_dir, "tv2_constraints.json")))
    q = cons["modular_arithmetic"]["q"]
    z = cons["verification"]["witness"]
    # R1CS check on the host harness (the C++ R1CS shim is out of scope: SURVEY.md §2)
    row = lambda ents: sum((e["value"] % q) * z[e["col"]] for e in ents) % q
    c0 = cons["constraints"][0]
    assert (row(c0["A"]) * row(c0["B"]) - row(c0["C"])) % q == 0
    lctx = pkg.LweContext(pkg.Params(q=q, n=4096, k=2, sigma=3.19), key_seed=0x1234)
    com_w = pkg.Commitment(lctx, z, seed=0x5678)
    assert np.array_equal(com_w.as_words(), oracle.lwe_commit(q, 4096, 2, 3.19, 0x1234, z, 0x5678))
    assert pkg.verify_opening_with_context(lctx, com_w, z)
    quotient = [0]                                                           # (A z)(B z) - C z = 0 on the single point => Q = 0
    com_q = pkg.Commitment(lctx, quotient, seed=0x5678)
    assert np.array_equal(com_q.as_words(), oracle.lwe_commit(q, 4096, 2, 3.19, 0x1234, quotient, 0x5678))
    assert pkg.verify_opening_with_context(lctx, com_q, quotient)
    lctx.close()


def test_config3_matvec_workload(pkg, oracle):
    """Config 3 shape (rank 4, n = 2^16) on a handful of witness vectors: u = INTT(A^T NTT(r)) + e1 bit-exact
    vs the oracle; e1 from the seeded sampler; plus the device-sampled-e1 variant."""
    import torch
    q, n, k, batch = 17592182243329, 65536, 4, 3
    lctx = pkg.LweContext(pkg.Params(q=q, n=n, k=k, sigma=3.19), key_seed=0xC0DE)
    assert lctx.commit_modulus == q
    a_hat = lctx.public_matrix()
    r = np.stack([oracle.splitmix(0xC0FFEE + j, q, k * n).reshape(k, n) for j in range(batch)])
    seeds = np.array([11, 22, 33], dtype=np.uint64)
    e1 = np.stack([np.stack([oracle.sample_gaussian_seeded(n, 3.19, int(seeds[j]), 5, i) for i in range(k)]) for j in range(batch)])
    e1 = np.where(e1 < 0, e1 + q, e1).astype(np.uint64)
    want = np.stack([oracle.mlwe_matvec(q, n, k, a_hat, r[j], e1[j]) for j in range(batch)])
    s = torch.cuda.current_stream().cuda_stream
    d_r = torch.from_numpy(r.view(np.int64)).cuda()
    d_e1 = torch.from_numpy(e1.view(np.int64)).cuda()
    d_u = torch.empty_like(d_r)
    assert lctx._lib.lsr_mlwe_matvec_batch_device(lctx.handle, d_r.data_ptr(), d_e1.data_ptr(), d_u.data_ptr(), batch, None, s) == 0
    torch.cuda.synchronize()
    assert np.array_equal(d_u.cpu().numpy().view(np.uint64), want)
    d_r = torch.from_numpy(r.view(np.int64)).cuda()
    d_u.zero_()
    assert lctx._lib.lsr_mlwe_matvec_batch_device(lctx.handle, d_r.data_ptr(), None, d_u.data_ptr(), batch, seeds.ctypes.data, s) == 0
    torch.cuda.synchronize()
    assert np.array_equal(d_u.cpu().numpy().view(np.uint64), want)
    lctx.close()


@pytest.mark.parametrize("n,k,batch", [(65536, 1, 3), (65536, 2, 5), (65536, 3, 2), (65536, 4, 67), (131072, 1, 2), (131072, 3, 3), (131072, 4, 2)])
def test_fused_pipeline_every_rank_and_degree(pkg, oracle, n, k, batch):
    """The fused NTT -> A^T product -> INTT pipeline serves n = 2^16 and 2^17 at ranks 1..4 (other shapes use the unfused
    kernels): u = INTT(A^T NTT(r)) + e1 bit-exact against the oracle, with e1 given and with e1 sampled per chunk on the
    device; a batch that is not a multiple of the chunk (64 vectors at k = 4) crosses a chunk and a stream boundary; r is left
    untouched by the fused path."""
    import torch
    q = oracle.L.oracle_lwe_select_modulus(0, n)
    lctx = pkg.LweContext(pkg.Params(q=q, n=n, k=k, sigma=3.19), key_seed=0xF00D + k)
    assert lctx.commit_modulus == q
    a_hat = lctx.public_matrix()
    r = np.stack([oracle.splitmix(0xC0FFEE + j, q, k * n).reshape(k, n) for j in range(batch)])
    seeds = (np.arange(batch, dtype=np.uint64) + np.uint64(7)) * np.uint64(0x9E3779B9)
    picks = sorted({0, batch // 2, batch - 1, min(63, batch - 1), min(64, batch - 1)})
    e1 = np.zeros((batch, k, n), dtype=np.uint64)
    s = torch.cuda.current_stream().cuda_stream
    d_e1 = torch.empty((batch, k, n), dtype=torch.int64, device="cuda")
    assert lctx._lib.lsr_lwe_sample_blinding_device(lctx.handle, d_e1.data_ptr(), batch, seeds.ctypes.data, s) == 0
    for j in picks:
        ej = np.stack([oracle.sample_gaussian_seeded(n, 3.19, int(seeds[j]), 5, i) for i in range(k)])
        e1[j] = np.where(ej < 0, ej + q, ej).astype(np.uint64)
        assert np.array_equal(d_e1[j].cpu().numpy().view(np.uint64), e1[j])
    d_r = torch.from_numpy(r.view(np.int64)).cuda()
    d_u = torch.empty_like(d_r)
    assert lctx._lib.lsr_mlwe_matvec_batch_device(lctx.handle, d_r.data_ptr(), d_e1.data_ptr(), d_u.data_ptr(), batch, None, s) == 0
    torch.cuda.synchronize()
    assert np.array_equal(d_r.cpu().numpy().view(np.uint64), r)                   # the fused path only reads r
    given = d_u.clone()
    for j in picks:
        assert np.array_equal(given[j].cpu().numpy().view(np.uint64), oracle.mlwe_matvec(q, n, k, a_hat, r[j], e1[j])), (n, k, j)
    d_u.zero_()
    assert lctx._lib.lsr_mlwe_matvec_batch_device(lctx.handle, d_r.data_ptr(), None, d_u.data_ptr(), batch, seeds.ctypes.data, s) == 0
    torch.cuda.synchronize()
    assert torch.equal(d_u, given)                                                  # every vector, not only the picks
    lctx.close()


@pytest.mark.parametrize("env", [{"LAMBDA_SNARK_COMMIT_FUSED": "0"}, {"LAMBDA_SNARK_COMMIT_MIXED": "0"}])
def test_alternative_commit_pipelines_agree(pkg, oracle, env, monkeypatch):
    """The two pipeline switches the library still reads (once, when a context is created; round 3 removed the round-2 experiment
    knobs): the unfused round-1 kernels and the three-launch schedule give the oracle's words, with the blinding residues sampled on the
    device and given.  The unfused sampled call runs on a FRESH context with a batch that grows the workspace: the stream keys are
    staged after the workspace is sized (round-2 advisor: use of a re-allocated key buffer)."""
    import torch
    for key, value in env.items():
        monkeypatch.setenv(key, value)
    q, n, batch = 17592182243329, 65536, 70
    s = torch.cuda.current_stream().cuda_stream
    for k in (4, 2):
        lctx = pkg.LweContext(pkg.Params(q=q, n=n, k=k, sigma=3.19), key_seed=0xBEEF + k)
        a_hat = lctx.public_matrix()
        r = np.stack([oracle.splitmix(0xC0FFEE + j, q, k * n).reshape(k, n) for j in range(batch)])
        seeds = (np.arange(batch, dtype=np.uint64) + np.uint64(3)) * np.uint64(0x9E3779B9)
        d_r = torch.from_numpy(r.view(np.int64)).cuda()
        d_u = torch.empty_like(d_r)
        assert lctx._lib.lsr_mlwe_matvec_batch_device(lctx.handle, d_r.data_ptr(), None, d_u.data_ptr(), batch, seeds.ctypes.data, s) == 0
        torch.cuda.synchronize()
        d_r.copy_(torch.from_numpy(r.view(np.int64)))                                  # the unfused kernels transform r in place
        d_e1 = torch.empty_like(d_r)
        assert lctx._lib.lsr_lwe_sample_blinding_device(lctx.handle, d_e1.data_ptr(), batch, seeds.ctypes.data, s) == 0
        d_u2 = torch.empty_like(d_r)
        assert lctx._lib.lsr_mlwe_matvec_batch_device(lctx.handle, d_r.data_ptr(), d_e1.data_ptr(), d_u2.data_ptr(), batch, None, s) == 0
        torch.cuda.synchronize()
        assert torch.equal(d_u, d_u2), (env, k)
        for j in (0, 63, 64, batch - 1):
            e1 = np.stack([oracle.sample_gaussian_seeded(n, 3.19, int(seeds[j]), 5, i) for i in range(k)])
            e1 = np.where(e1 < 0, e1 + q, e1).astype(np.uint64)
            assert np.array_equal(d_u[j].cpu().numpy().view(np.uint64), oracle.mlwe_matvec(q, n, k, a_hat, r[j], e1)), (env, k, j)
        lctx.close()


def test_mixed_launch_schedule_agrees(pkg, oracle, monkeypatch):
    """The default schedule for n = 2^16 with the blinding residues given: the middle stage of chunk t, the forward strided round of
    chunk t + 1 and the inverse strided round (+ e1) of chunk t - 1 as roles of one launch (mlwe_mixed).  Ragged batches over one, two
    and several chunks, ranks 1-4: every word of u equals the three-launch pipeline's (a second context of the same keys created with
    LAMBDA_SNARK_COMMIT_MIXED=0), sampled vectors equal the oracle's."""
    import torch
    q, n = 17592182243329, 65536
    s = torch.cuda.current_stream().cuda_stream
    for k, batch in ((4, 150), (4, 64), (4, 5), (2, 70), (1, 299), (3, 65), (4, 33)):
        monkeypatch.setenv("LAMBDA_SNARK_COMMIT_MIXED", "0")
        three = pkg.LweContext(pkg.Params(q=q, n=n, k=k, sigma=3.19), key_seed=0xFACE + k)
        monkeypatch.setenv("LAMBDA_SNARK_COMMIT_MIXED", "1")
        lctx = pkg.LweContext(pkg.Params(q=q, n=n, k=k, sigma=3.19), key_seed=0xFACE + k)
        monkeypatch.setenv("LAMBDA_SNARK_COMMIT_MIXED", "0")       # read at creation only: changing it now must not matter
        a_hat = lctx.public_matrix()
        d_r = torch.empty((batch, k, n), dtype=torch.int64, device="cuda")
        assert lctx._lib.lsr_fill_splitmix_device(d_r.data_ptr(), batch, k * n, 0xC0FFEE + 11 * k, q, s) == 0
        seeds = (np.arange(batch, dtype=np.uint64) + np.uint64(5)) * np.uint64(0x9E3779B9)
        d_e1 = torch.empty_like(d_r)
        assert lctx._lib.lsr_lwe_sample_blinding_device(lctx.handle, d_e1.data_ptr(), batch, seeds.ctypes.data, s) == 0
        d_u0, d_u1 = torch.empty_like(d_r), torch.empty_like(d_r)
        keep = d_r.clone()
        assert three._lib.lsr_mlwe_matvec_batch_device(three.handle, d_r.data_ptr(), d_e1.data_ptr(), d_u0.data_ptr(), batch, None, s) == 0
        assert lctx._lib.lsr_mlwe_matvec_batch_device(lctx.handle, d_r.data_ptr(), d_e1.data_ptr(), d_u1.data_ptr(), batch, None, s) == 0
        torch.cuda.synchronize()
        assert torch.equal(d_r, keep), "the fused pipelines only read r"
        assert torch.equal(d_u0, d_u1), (k, batch)
        for j in sorted({0, min(63, batch - 1), min(64, batch - 1), batch - 1}):
            r_j = d_r[j].cpu().numpy().view(np.uint64)
            e1_j = d_e1[j].cpu().numpy().view(np.uint64)
            assert np.array_equal(d_u1[j].cpu().numpy().view(np.uint64), oracle.mlwe_matvec(q, n, k, a_hat, r_j, e1_j)), (k, batch, j)
        lctx.close()
        three.close()


def test_config3_full_size_device_resident(pkg, oracle):
    """BASELINE config 3 at FULL size: rank 4, n = 2^16, 1024 witness vectors (2 GiB of r), device-resident, through
    lsr_mlwe_matvec_batch_device with e1 drawn by the seeded CDT sampler (sigma = 3.19).  r_j uniform from splitmix64
    seed 0xC0FFEE + j.  Checks: (1) sampled witness vectors — first / last of a register-blocked group
    of the matrix stage, of a transform chunk (128 vectors = 512 polynomials) and of the batch — bit-exact against
    oracle.mlwe_matvec with the oracle's own e1; (2) linearity in r over the whole batch, with e1 = 0:
    L(r_a + r_b) = L(r_a) + L(r_b) for all 512 pairs (a, a + 512)."""
    import torch
    q, n, k, batch = 17592182243329, 65536, 4, 1024
    lctx = pkg.LweContext(pkg.Params(q=q, n=n, k=k, sigma=3.19), key_seed=0xC0DE)
    a_hat = lctx.public_matrix()
    r = np.empty((batch, k, n), dtype=np.uint64)
    for j in range(batch):
        r[j] = oracle.splitmix(0xC0FFEE + j, q, k * n).reshape(k, n)
    seeds = np.arange(1, batch + 1, dtype=np.uint64) * np.uint64(0x9E3779B9)
    s = torch.cuda.current_stream().cuda_stream
    d_r = torch.from_numpy(r.view(np.int64)).cuda()
    d_u = torch.empty_like(d_r)
    assert lctx._lib.lsr_mlwe_matvec_batch_device(lctx.handle, d_r.data_ptr(), None, d_u.data_ptr(), batch, seeds.ctypes.data, s) == 0
    torch.cuda.synchronize()
    picks = [0, 3, 4, 127, 128, 511, 512, 1020, 1023]
    for j in picks:
        e1 = np.stack([oracle.sample_gaussian_seeded(n, 3.19, int(seeds[j]), 5, i) for i in range(k)])
        e1 = np.where(e1 < 0, e1 + q, e1).astype(np.uint64)
        want = oracle.mlwe_matvec(q, n, k, a_hat, r[j], e1)
        assert np.array_equal(d_u[j].cpu().numpy().view(np.uint64), want), f"witness vector {j}"
    # every output is a canonical residue
    assert int(d_u.max().item()) < q and int(d_u.min().item()) >= 0
    sampled_u = d_u.clone()
    # linearity with e1 = 0 over the whole batch
    half = batch // 2
    d_r.copy_(torch.from_numpy(r.view(np.int64)))
    d_sum = (d_r[:half] + d_r[half:]) % q
    zeros = torch.zeros_like(d_r)
    assert lctx._lib.lsr_mlwe_matvec_batch_device(lctx.handle, d_r.data_ptr(), zeros.data_ptr(), d_u.data_ptr(), batch, None, s) == 0
    d_us = torch.empty_like(d_sum)
    assert lctx._lib.lsr_mlwe_matvec_batch_device(lctx.handle, d_sum.data_ptr(), zeros.data_ptr(), d_us.data_ptr(), half, None, s) == 0
    torch.cuda.synchronize()
    assert torch.equal(d_us, (d_u[:half] + d_u[half:]) % q)
    # the DEFAULT schedule (mixed launches: e1 given as an array) at the same full size: every word equals the e1-sampled pipeline's
    # output above, and the picks equal the oracle's (round-2 verdict: the mixed default was only linearity-checked at 1024 vectors)
    del d_us, d_sum, zeros
    d_e1 = torch.empty_like(d_r)
    assert lctx._lib.lsr_lwe_sample_blinding_device(lctx.handle, d_e1.data_ptr(), batch, seeds.ctypes.data, s) == 0
    d_um = torch.empty_like(d_r)
    assert lctx._lib.lsr_mlwe_matvec_batch_device(lctx.handle, d_r.data_ptr(), d_e1.data_ptr(), d_um.data_ptr(), batch, None, s) == 0
    torch.cuda.synchronize()
    for j in picks:
        e1 = np.stack([oracle.sample_gaussian_seeded(n, 3.19, int(seeds[j]), 5, i) for i in range(k)])
        e1 = np.where(e1 < 0, e1 + q, e1).astype(np.uint64)
        assert np.array_equal(d_um[j].cpu().numpy().view(np.uint64), oracle.mlwe_matvec(q, n, k, a_hat, r[j], e1)), f"mixed schedule, witness vector {j}"
    assert torch.equal(d_um, sampled_u), "mixed launches (e1 given) and the three-launch schedule (e1 sampled in the pass) disagree"
    lctx.close()


def test_verify_opening_batch_matches_single_calls(pkg, oracle):
    """lwe_verify_opening_batch: same 1 / 0 / -1 as one lwe_verify_opening per entry, and as the oracle."""
    q, n, k = 17592186044417, 4096, 2
    lctx = pkg.LweContext(pkg.Params(q=q, n=n, k=k, sigma=3.19), key_seed=KEY)
    count, msg_len = 37, 5
    rng = np.random.default_rng(3)
    msgs = rng.integers(0, 1000, size=(count, msg_len)).astype(np.uint64)
    coms = pkg.Commitment.batch(lctx, msgs, np.arange(1, count + 1, dtype=np.uint64))
    claimed = msgs.copy()
    wrong = [3, 11, 36]
    for w in wrong:
        claimed[w, w % msg_len] += 1
    entries = list(coms)
    entries[5] = None                                      # NULL entry => -1
    got = pkg.verify_openings_batch(lctx, entries, claimed)
    want = [(-1 if i == 5 else (0 if i in wrong else 1)) for i in range(count)]
    assert got == want
    for i in (0, 3, 11, 20):
        assert int(pkg.verify_opening_with_context(lctx, coms[i], claimed[i])) == want[i]
        assert oracle.lwe_verify(q, n, k, 3.19, KEY, coms[i].as_words(), [int(x) for x in claimed[i]]) == want[i]
    # linear combinations verify in the same pass
    comb = pkg.Commitment.linear_combine(lctx, [coms[0], coms[1]], [2, 3])
    assert pkg.verify_openings_batch(lctx, [comb, comb], [2 * msgs[0] + 3 * msgs[1], 2 * msgs[0] + 3 * msgs[1] + 1]) == [1, 0]
    lctx.close()


def test_empty_message_and_zero_length_openings(pkg, lib, oracle):
    """msg_len = 0 is legal at the ABI (commitment.cpp:146-149 copies min(msg_len, slots) = 0 values): a commitment to the
    all-zero message; an opening of zero slots is vacuously valid."""
    q, n, k = 17592186044417, 4096, 2
    lctx = pkg.LweContext(pkg.Params(q=q, n=n, k=k, sigma=3.19), key_seed=KEY)
    one = np.array([5], dtype=np.uint64)
    p = lib.lwe_commit(lctx.handle, one.ctypes.data, 0, 77)
    assert p
    assert np.array_equal(words(p), oracle.lwe_commit(q, n, k, 3.19, KEY, [], 77))
    assert np.array_equal(words(p), oracle.lwe_commit(q, n, k, 3.19, KEY, [0, 0, 0], 77))
    assert lib.lwe_verify_opening(lctx.handle, p, one.ctypes.data, 0, None) == 1
    zeros = np.zeros(n, dtype=np.uint64)
    assert lib.lwe_verify_opening(lctx.handle, p, zeros.ctypes.data, n, None) == 1
    assert lib.lwe_verify_opening(lctx.handle, p, one.ctypes.data, 1, None) == 0
    lib.lwe_commitment_free(p)
    lctx.close()


def test_large_combination_coefficients_need_a_wide_modulus(pkg, oracle):
    """with the 44-bit internal modulus the noise budget admits sum(c_i) up to ~2^10; a caller that
    combines with coefficients as large as the plaintext modulus (the reference's 72-bit SEAL modulus allows that,
    commitment.cpp:88-96) passes a 60-bit NTT prime as params->modulus, which the library honours."""
    big = 1152921504606584833                       # 60-bit prime = 1 (mod 2^18)
    t = 1032193
    coeffs = [t - 1, t - 2, 777777]
    msgs = [[1, 2, 3, 4], [5, 6, 7, 8], [9, 10, 11, 12]]
    expect = [sum(c * m[i] for c, m in zip(coeffs, msgs)) % t for i in range(4)]
    wide = pkg.LweContext(pkg.Params(q=big, n=4096, k=2, sigma=3.19), key_seed=5)
    assert wide.commit_modulus == big and wide.plain_modulus == t
    coms = [pkg.Commitment(wide, m, seed=10 + i) for i, m in enumerate(msgs)]
    comb = pkg.Commitment.linear_combine(wide, coms, coeffs)
    assert pkg.verify_opening_with_context(wide, comb, expect)
    assert not pkg.verify_opening_with_context(wide, comb, [expect[0] + 1] + expect[1:])
    wide.close()
    # the library names such a prime itself (lsr_lwe_wide_modulus): sixteen terms with coefficients all over [0, t) — the reference's
    # own range (commitment.cpp:247-266) — combine and open, word for word the oracle's combination
    q60 = pkg.wide_modulus(4096)
    assert 2**59 < q60 < 2**60 and (q60 - 1) % 8192 == 0 and oracle.L.oracle_is_prime(q60) == 1 and q60 == oracle.L.oracle_largest_prime_1mod(8192, 60) and pkg.wide_modulus(4097) == 0
    rng = np.random.default_rng(60)
    many = [[int(x) for x in rng.integers(0, t, 6)] for _ in range(16)]
    cs = [int(x) for x in rng.integers(0, t, 16)]
    ctx60 = pkg.LweContext(pkg.Params(q=q60, n=4096, k=2, sigma=3.19), key_seed=6)
    assert ctx60.commit_modulus == q60
    coms = [pkg.Commitment(ctx60, m, seed=100 + i) for i, m in enumerate(many)]
    comb = pkg.Commitment.linear_combine(ctx60, coms, cs)
    assert pkg.verify_opening_with_context(ctx60, comb, [sum(c * m[i] for c, m in zip(cs, many)) % t for i in range(6)])
    ctx60.close()
    narrow = pkg.LweContext(pkg.Params(q=17592186044417, n=4096, k=2, sigma=3.19), key_seed=5)
    coms = [pkg.Commitment(narrow, m, seed=10 + i) for i, m in enumerate(msgs)]
    small = pkg.Commitment.linear_combine(narrow, coms, [200, 300, 200])
    assert pkg.verify_opening_with_context(narrow, small, [sum(c * m[i] for c, m in zip([200, 300, 200], msgs)) % t for i in range(4)])
    # beyond the budget of the 44-bit modulus the call fails loudly (NULL + message) instead of returning a commitment that
    # cannot be opened; the oracle draws the same line
    with pytest.raises(pkg.CoreError):
        pkg.Commitment.linear_combine(narrow, coms, coeffs)
    assert "noise budget" in pkg._abi.last_error()
    rc, _ = oracle.lwe_linear_combine(17592186044417, 4096, 2, 3.19, 5, [c.as_words() for c in coms], coeffs)
    assert rc == -1
    narrow.close()
    # a context whose fresh commitments could not be opened is refused at creation
    with pytest.raises(pkg.CoreError):
        pkg.LweContext(pkg.Params(q=17592186044417, n=4096, k=2, sigma=200.0), key_seed=5)


def test_blinding_never_repeats_across_messages_or_contexts(pkg, oracle):
    """Same seed, different messages / contexts: fresh blinding (round-1 advisor finding, high).  Bit-exact with the oracle's
    key schedule; same (seed, message, context) still reproduces the commitment."""
    q, n, k = 17592186044417, 4096, 2
    a = pkg.LweContext(pkg.Params(q=q, n=n, k=k, sigma=3.19), key_seed=0xABC)
    b = pkg.LweContext(pkg.Params(q=q, n=n, k=k, sigma=3.19), key_seed=0xABD)
    m1, m2 = [1, 2, 3, 4], [5, 9, 3, 1000]
    c1, c2 = pkg.Commitment(a, m1, seed=42).as_words(), pkg.Commitment(a, m2, seed=42).as_words()
    assert np.array_equal(c1, oracle.lwe_commit(q, n, k, 3.19, 0xABC, m1, 42))
    assert np.array_equal(c2, oracle.lwe_commit(q, n, k, 3.19, 0xABC, m2, 42))
    u1, u2 = c1[5:5 + k * n], c2[5:5 + k * n]
    assert np.count_nonzero(u1 == u2) < 8
    t, qi = a.plain_modulus, a.commit_modulus
    v1, v2 = c1[5 + k * n:], c2[5 + k * n:]
    leak = [(int(v1[i]) - int(v2[i]) - (qi // t) * (m1[i] - m2[i])) % qi for i in range(4)]
    assert all(min(x, qi - x) > 10**6 for x in leak)
    assert np.array_equal(c1, pkg.Commitment(a, m1, seed=42).as_words())
    assert np.array_equal(pkg.Commitment(b, m1, seed=42).as_words(), oracle.lwe_commit(q, n, k, 3.19, 0xABD, m1, 42))
    # batch path: every row gets its own message-bound key
    msgs = np.array([m1, m2, m1], dtype=np.uint64)
    rows = pkg.Commitment.batch_words(a, msgs, np.array([42, 42, 42], dtype=np.uint64))
    assert np.array_equal(rows[0], c1) and np.array_equal(rows[1], c2) and np.array_equal(rows[2], c1)
    a.close(); b.close()


def test_verify_compares_raw_message_words(pkg, lib, ctx):
    """commitment.cpp:223-226: decoded[i] ^ message[i] on the words as given — a claimed word >= t never opens."""
    t = ctx.plain_modulus
    msg = np.array([1, 2, 3, 4], dtype=np.uint64)
    comm = lib.lwe_commit(ctx.handle, msg.ctypes.data, 4, 77)
    assert lib.lwe_verify_opening(ctx.handle, comm, msg.ctypes.data, 4, None) == 1
    shifted = msg.copy(); shifted[0] += np.uint64(t)
    assert lib.lwe_verify_opening(ctx.handle, comm, shifted.ctypes.data, 4, None) == 0
    res = np.zeros(2, dtype=np.int32)
    both = np.stack([msg, shifted])
    arr = (ctypes.POINTER(pkg._abi.LweCommitment) * 2)(comm, comm)
    assert lib.lwe_verify_opening_batch(ctx.handle, arr, both.ctypes.data, 4, 2, res.ctypes.data) == 0
    assert list(res) == [1, 0]
    # the commit side embeds m mod t: the same commitment under the same seed
    again = lib.lwe_commit(ctx.handle, shifted.ctypes.data, 4, 77)
    assert np.array_equal(words(comm), words(again))
    lib.lwe_commitment_free(comm); lib.lwe_commitment_free(again)


@pytest.mark.parametrize("row_words,n_inputs", [(1, 0), (5, 2), (13, 0), (14, 1), (15, 3), (16, 0), (30, 17), (31, 40), (300, 3), (12293, 2)])
def test_device_transcripts_match_hashlib(lib, row_words, n_inputs):
    """lsr_fs_challenge_batch_device against hashlib's SHA3-256 over the transcript of challenge.rs:102-134 — row lengths
    around the 136-byte block boundaries, public inputs that spill over several blocks, and the reference-size row."""
    import torch
    import prover_replay
    count = 70 if row_words < 1000 else 5
    rng = np.random.default_rng(row_words * 131 + n_inputs)
    rows = rng.integers(0, 2**64, size=(count, row_words), dtype=np.uint64)
    ins = rng.integers(0, 2**64, size=(count, max(n_inputs, 1)), dtype=np.uint64)[:, :n_inputs]
    d_rows = torch.from_numpy(rows.view(np.int64)).cuda()
    d_ins = torch.from_numpy(np.ascontiguousarray(ins).view(np.int64)).cuda() if n_inputs else None
    d_alpha = torch.zeros(count, dtype=torch.int64, device="cuda")
    d_hash = torch.zeros((count, 32), dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for modulus in (18446744069414584321, 17592186044417, 12289):
        assert lib.lsr_fs_challenge_batch_device(d_ins.data_ptr() if n_inputs else None, n_inputs, d_rows.data_ptr(), row_words, count, modulus,
                                                 d_alpha.data_ptr(), d_hash.data_ptr(), s) == 0
        torch.cuda.synchronize()
        alphas = d_alpha.cpu().numpy().view(np.uint64); hashes = d_hash.cpu().numpy()
        for i in range(count):
            want_alpha, want_hash = prover_replay.challenge_derive([int(x) for x in ins[i]], rows[i], modulus)
            assert int(alphas[i]) == want_alpha and bytes(hashes[i]) == want_hash, (row_words, n_inputs, i)
    assert lib.lsr_fs_challenge_batch_device(None, 1, d_rows.data_ptr(), row_words, count, 12289, d_alpha.data_ptr(), None, s) == -1
    assert lib.lsr_fs_challenge_batch_device(None, 0, d_rows.data_ptr(), row_words, 0, 12289, d_alpha.data_ptr(), None, s) == 0
    assert lib.lsr_fs_challenge_batch_device(None, 0, d_rows.data_ptr(), row_words, count, 12289, d_alpha.data_ptr(), None, s) == 0   # hashes optional


def test_commit_rows_and_both_challenges_stay_on_the_device(pkg, lib, oracle):
    """lsr_lwe_commit_batch_flat_device -> alpha -> beta (public input = alpha, lib.rs:768) without the rows leaving the GPU."""
    import torch
    import prover_replay
    q, n, k, key = 17592186044417, 4096, 2, 0xFEED
    ctx = pkg.LweContext(pkg.Params(q=q, n=n, k=k, sigma=3.19), key_seed=key)
    batch, W = 6, lib.lsr_lwe_commitment_words(ctx.handle)
    rng = np.random.default_rng(12)
    msgs = rng.integers(0, 2**20, size=(batch, 7), dtype=np.uint64)
    seeds = rng.integers(1, 2**62, size=batch, dtype=np.uint64)
    publics = rng.integers(0, 2**44, size=(batch, 2), dtype=np.uint64)
    d_rows = torch.zeros((batch, W), dtype=torch.int64, device="cuda")
    assert lib.lsr_lwe_commit_batch_flat_device(ctx.handle, msgs.ctypes.data, 7, batch, seeds.ctypes.data, d_rows.data_ptr()) == 0
    d_pub = torch.from_numpy(publics.view(np.int64)).cuda()
    d_alpha = torch.zeros(batch, dtype=torch.int64, device="cuda"); d_beta = torch.zeros_like(d_alpha)
    s = torch.cuda.current_stream().cuda_stream
    assert lib.lsr_fs_challenge_batch_device(d_pub.data_ptr(), 2, d_rows.data_ptr(), W, batch, q, d_alpha.data_ptr(), None, s) == 0
    assert lib.lsr_fs_challenge_batch_device(d_alpha.data_ptr(), 1, d_rows.data_ptr(), W, batch, q, d_beta.data_ptr(), None, s) == 0
    torch.cuda.synchronize()
    rows = d_rows.cpu().numpy().view(np.uint64)
    host_rows = pkg.Commitment.batch_words(ctx, msgs, seeds)
    assert np.array_equal(rows, host_rows)
    for i in range(batch):
        assert np.array_equal(rows[i], oracle.lwe_commit(q, n, k, 3.19, key, msgs[i], int(seeds[i])))
        alpha, _ = prover_replay.challenge_derive([int(x) for x in publics[i]], rows[i], q)
        beta, _ = prover_replay.challenge_derive([alpha], rows[i], q)
        assert (int(d_alpha[i].item()) & (2**64 - 1), int(d_beta[i].item()) & (2**64 - 1)) == (alpha, beta)
    assert lib.lsr_lwe_commit_batch_flat_device(ctx.handle, msgs.ctypes.data, 7, batch, seeds.ctypes.data, None) == -1
    ctx.close()
