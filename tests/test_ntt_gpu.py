"""GPU parity tests for the NTT path, all through the C-ABI.  Reads like cpp-core/tests/test_ntt.cpp, then
widens: every ring degree the reference accepts, both arithmetic flavours, ragged batches, the
device-resident API, and size-independent properties at the BASELINE config-2 size."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

Q12, Q44, Q16, Q60 = 12289, 17592169062401, 17592182243329, 1152921504606584833


@pytest.fixture(scope="module")
def ctx256(pkg):
    c = pkg.NttContext(Q12, 256)      # test_ntt.cpp:13-17
    yield c
    c.close()


# ---- cpp-core/tests/test_ntt.cpp, one test per TEST_F ------------------------------------------------
def test_create_and_free(ctx256):
    assert ctx256.handle


def test_forward_ntt_basic(ctx256, lib):
    coeffs = np.ones(256, dtype=np.uint64)
    assert lib.ntt_forward(ctx256.handle, coeffs.ctypes.data, 256) == 0      # test_ntt.cpp:33-38


def test_inverse_ntt_basic(ctx256, lib):
    evals = np.ones(256, dtype=np.uint64)
    assert lib.ntt_inverse(ctx256.handle, evals.ctypes.data, 256) == 0       # test_ntt.cpp:40-45


def test_forward_inverse_identity(ctx256, lib):
    original = np.zeros(256, dtype=np.uint64)
    original[:8] = np.arange(1, 9)
    x = original.copy()
    assert lib.ntt_forward(ctx256.handle, x.ctypes.data, 256) == 0
    assert not np.array_equal(x, original)
    assert lib.ntt_inverse(ctx256.handle, x.ctypes.data, 256) == 0
    assert np.array_equal(x, original)                                        # test_ntt.cpp:47-68


def test_pointwise_multiplication(ctx256, lib):
    a, b, r = np.full(256, 2, np.uint64), np.full(256, 3, np.uint64), np.zeros(256, np.uint64)
    lib.ntt_mul_pointwise(ctx256.handle, r.ctypes.data, a.ctypes.data, b.ctypes.data, 256)
    assert (r == 6).all()                                                     # test_ntt.cpp:70-81


def test_null_pointer_handling(ctx256, lib):
    dummy = np.zeros(256, dtype=np.uint64)
    assert lib.ntt_forward(None, dummy.ctypes.data, 256) == -1                # test_ntt.cpp:86
    assert lib.ntt_forward(ctx256.handle, None, 256) == -1                    # test_ntt.cpp:87
    assert lib.ntt_forward(ctx256.handle, dummy.ctypes.data, 128) == -1       # ntt.cpp:81 n != degree
    assert lib.ntt_inverse(ctx256.handle, dummy.ctypes.data, 512) == -1
    lib.ntt_context_free(None)                                                # test_ntt.cpp:89
    before = dummy.copy()
    lib.ntt_mul_pointwise(ctx256.handle, dummy.ctypes.data, None, dummy.ctypes.data, 256)   # ntt.cpp:113 silent
    assert np.array_equal(dummy, before)


def test_sys_crate_smoke(lib):
    """rust-api/lambda-snark-sys/src/lib.rs:35-42."""
    c = lib.ntt_context_create(12289, 256)
    assert c
    lib.ntt_context_free(c)


# ---- parity with the oracle ---------------------------------------------------------------------------
CASES = [(Q12, 2, 9), (Q12, 4, 5), (Q12, 8, 3), (Q12, 16, 3), (Q12, 32, 3), (Q12, 64, 65), (Q12, 128, 3), (Q12, 256, 17), (Q12, 512, 3),
         (Q12, 1024, 5), (Q12, 2048, 3), (Q44, 2048, 2), (Q44, 4096, 9), (Q16, 8192, 3), (Q16, 16384, 2), (Q16, 32768, 2), (Q16, 65536, 5),
         (Q60, 4096, 2), (Q60, 65536, 2), (Q60, 131072, 1)]


@pytest.mark.parametrize("mode", [0, 1], ids=["auto", "u64"])
@pytest.mark.parametrize("q,n,batch", CASES)
def test_forward_inverse_bit_exact(pkg, lib, oracle, q, n, batch, mode):
    lib.lsr_set_arith_mode(mode)
    try:
        ctx = pkg.NttContext(q, n)
        assert ctx.uses_f64 == (mode == 0 and q < 2**45)
        assert ctx.root == oracle.root(q, n)
        a = oracle.splitmix(0xDEADBEEF + n + batch, q, batch * n).reshape(batch, n)
        a[0, 0] = q - 1; a[0, 1] = 0; a[-1, -1] = q - 1                     # extremes
        want = oracle.ntt_forward(q, n, a)
        got = ctx.forward_batch(a)
        assert np.array_equal(got, want)
        assert np.array_equal(ctx.inverse_batch(got), a)
        assert np.array_equal(ctx.inverse_batch(a), oracle.ntt_inverse(q, n, a))
        assert np.array_equal(ctx.forward(a[0]), want[0])                     # single-poly ABI
        ctx.close()
    finally:
        lib.lsr_set_arith_mode(0)


@pytest.mark.parametrize("q,n", [(Q12, 256), (Q44, 4096), (Q16, 65536), (Q60, 4096)])
def test_worst_case_operands(pkg, oracle, q, n):
    """all q-1, all zero, alternating 0 / q-1, single spikes: stresses the lazy ranges of both flavours."""
    ctx = pkg.NttContext(q, n)
    rows = [np.full(n, q - 1, np.uint64), np.zeros(n, np.uint64), np.where(np.arange(n) % 2, q - 1, 0).astype(np.uint64)]
    spike = np.zeros(n, np.uint64); spike[n - 1] = q - 1; rows.append(spike)
    spike0 = np.zeros(n, np.uint64); spike0[0] = 1; rows.append(spike0)
    a = np.stack(rows)
    f = ctx.forward_batch(a)
    assert np.array_equal(f, oracle.ntt_forward(q, n, a))
    assert (f[4] == 1).all()                                                  # NTT(1) = all ones
    assert np.array_equal(ctx.inverse_batch(f), a)
    assert np.array_equal(ctx.inverse_batch(a), oracle.ntt_inverse(q, n, a))
    ctx.close()


def test_inputs_up_to_4q_are_tolerated(pkg, oracle):
    """the reference's lazy butterflies accept inputs < 4q."""
    for q, n in [(Q44, 4096), (Q16, 65536)]:
        ctx = pkg.NttContext(q, n)
        a = oracle.splitmix(77, q, n)
        assert np.array_equal(ctx.forward(a + np.uint64(3 * q)), oracle.ntt_forward(q, n, a))
        ctx.close()


def test_pointwise_matches_oracle_any_64bit_inputs(pkg, oracle):
    """ntt.cpp:116-118 is correct for any 64-bit operands; n is not validated; result may alias."""
    for q in [Q12, Q44, Q60]:
        ctx = pkg.NttContext(q, 256)
        rng = np.random.default_rng(5)
        a = rng.integers(0, 2**64, size=1000, dtype=np.uint64)
        b = rng.integers(0, 2**64, size=1000, dtype=np.uint64)
        a[:4] = [0, 2**64 - 1, q - 1, q]; b[:4] = [5, 2**64 - 1, q - 1, q]
        got = ctx.mul_pointwise(a, b)                                        # n = 1000 != degree
        assert np.array_equal(got, oracle.mul_pointwise(q, 256, a, b))
        alias = a.copy()
        ctx._lib.ntt_mul_pointwise(ctx.handle, alias.ctypes.data, alias.ctypes.data, b.ctypes.data, alias.size)
        assert np.array_equal(alias, got)
        ctx.close()


def test_convolution_through_the_abi(pkg, oracle):
    q, n = Q44, 4096
    ctx = pkg.NttContext(q, n)
    a, b = oracle.splitmix(1, q, n), oracle.splitmix(2, q, n)
    got = ctx.inverse(ctx.mul_pointwise(ctx.forward(a), ctx.forward(b)))
    want = oracle.ntt_inverse(q, n, oracle.mul_pointwise(q, n, oracle.ntt_forward(q, n, a), oracle.ntt_forward(q, n, b)))
    assert np.array_equal(got, want)
    ctx.close()


def test_batch_edge_cases(pkg, lib, oracle):
    ctx = pkg.NttContext(Q44, 4096)
    assert lib.ntt_forward_batch(ctx.handle, None, 4) == -1
    assert lib.ntt_forward_batch(None, None, 4) == -1
    dummy = np.zeros(4096, np.uint64)
    assert lib.ntt_forward_batch(ctx.handle, dummy.ctypes.data, 0) == 0       # empty batch
    ctx.close()
    # ragged: batch * n not a multiple of the 4096-element tile
    ctx = pkg.NttContext(Q12, 64)
    for batch in [1, 63, 64, 65, 129]:
        a = oracle.splitmix(batch, Q12, batch * 64).reshape(batch, 64)
        assert np.array_equal(ctx.forward_batch(a), oracle.ntt_forward(Q12, 64, a))
        assert np.array_equal(ctx.inverse_batch(a), oracle.ntt_inverse(Q12, 64, a))
    ctx.close()


def test_device_resident_api(pkg, oracle):
    import torch
    q, n, batch = Q16, 65536, 6
    ctx = pkg.NttContext(q, n, device=0)
    host = oracle.splitmix(3, q, batch * n).reshape(batch, n)
    dev = torch.from_numpy(host.view(np.int64)).cuda()
    other = torch.cuda.Stream()
    other.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(other):
        ctx.forward_device(dev.data_ptr(), batch, other.cuda_stream)
    other.synchronize()
    assert np.array_equal(dev.cpu().numpy().view(np.uint64), oracle.ntt_forward(q, n, host))
    b = torch.from_numpy(oracle.splitmix(4, q, batch * n).view(np.int64)).cuda()
    out = torch.empty_like(b)
    ctx.mul_pointwise_device(out.data_ptr(), dev.data_ptr(), b.data_ptr(), batch * n, torch.cuda.current_stream().cuda_stream)
    ctx.inverse_device(dev.data_ptr(), batch, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(dev.cpu().numpy().view(np.uint64), host)
    want = oracle.mul_pointwise(q, n, oracle.ntt_forward(q, n, host).ravel(), b.cpu().numpy().view(np.uint64))
    assert np.array_equal(out.cpu().numpy().view(np.uint64), want)
    ctx.close()


def test_config2_full_size_properties(pkg, oracle):
    """BASELINE config 2: 4096 polys x n = 2^16 (2 GiB), device-resident.  Size-independent checks:
    round trip, linearity NTT(a+b) = NTT(a)+NTT(b), and a sample of polynomials against the oracle."""
    import torch
    q, n, batch = Q16, 65536, 4096
    ctx = pkg.NttContext(q, n, device=0)
    gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
    a = torch.randint(0, q, (batch, n), dtype=torch.int64, device="cuda", generator=gen)
    orig = a.clone()
    s = torch.cuda.current_stream().cuda_stream
    ctx.forward_device(a.data_ptr(), batch, s)
    torch.cuda.synchronize()
    picks = [0, 1, 2047, 2048, 4095]
    want = oracle.ntt_forward(q, n, orig[picks].cpu().numpy().view(np.uint64))
    assert np.array_equal(a[picks].cpu().numpy().view(np.uint64), want)
    # linearity on the first half vs second half
    half = batch // 2
    summed = (orig[:half] + orig[half:]) % q
    ctx.forward_device(summed.data_ptr(), half, s)
    torch.cuda.synchronize()
    assert torch.equal(summed, (a[:half] + a[half:]) % q)
    del summed
    ctx.inverse_device(a.data_ptr(), batch, s)
    torch.cuda.synchronize()
    assert torch.equal(a, orig)
    ctx.close()


def test_gpu_reproduces_the_independent_check_values(pkg, lib, golden_dir):
    """tests/golden/ntt_check_values.json (written by the pure-Python generator next to it: minimal-psi search, even/odd
    evaluation of the defining formula — no oracle, no library) against the GPU directly: psi, the first outputs and the
    SHA-256 of the whole forward transform of a[i] = splitmix64_i(0xDEADBEEF) mod q, for every case in the file (n = 256 ... 2^17,
    44- and 60-bit moduli, both arithmetic flavours where the modulus allows), through the legacy host-pointer ABI."""
    import hashlib, json, os
    with open(os.path.join(golden_dir, "ntt_check_values.json")) as f:
        golden = json.load(f)
    mask = (1 << 64) - 1

    def splitmix(seed, count):
        x, out = seed, []
        for _ in range(count):
            x = (x + 0x9E3779B97F4A7C15) & mask
            z = x
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
            out.append(z ^ (z >> 31))
        return out

    for case in golden["cases"]:
        q, n = case["q"], case["n"]
        a = np.array([z % q for z in splitmix(golden["splitmix_seed"], n)], dtype=np.uint64)
        assert [int(x) for x in a[:3]] == case["a"]
        for mode in ((0, 1) if q < 2**45 else (1,)):          # FP64-FMA flavour and SEAL's u64 Harvey/Shoup butterflies
            lib.lsr_set_arith_mode(mode)
            ctx = pkg.NttContext(q, n)
            lib.lsr_set_arith_mode(0)
            x = a.copy()
            assert lib.ntt_forward(ctx.handle, x.ctypes.data, n) == 0
            assert [int(v) for v in x[:3]] == case["ntt"], (q, n, mode)
            assert hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest() == case["sha256"], (q, n, mode)
            assert lib.ntt_inverse(ctx.handle, x.ctypes.data, n) == 0
            assert np.array_equal(x, a)
            ctx.close()


@pytest.mark.parametrize("n", [4096, 65536, 131072])
def test_f64_flavour_at_its_modulus_bound(pkg, oracle, n):
    """the FP64-FMA Barrett arithmetic is exact for q < 2^45.  Largest 45-bit primes = 1 (mod 2n),
    operands that maximise every lazy range (all q-1; alternating; inputs up to 4q-1), forward and inverse."""
    q = oracle.L.oracle_largest_prime_1mod(2 * n, 45)
    assert 2**44 < q < 2**45
    ctx = pkg.NttContext(q, n)
    assert ctx.uses_f64
    rows = [np.full(n, q - 1, np.uint64), np.where(np.arange(n) % 2, q - 1, 0).astype(np.uint64), np.where(np.arange(n) % 3, 0, q - 1).astype(np.uint64),
            oracle.splitmix(11, q, n), oracle.splitmix(12, q, n)]
    a = np.stack(rows)
    want = oracle.ntt_forward(q, n, a)
    assert np.array_equal(ctx.forward_batch(a), want)
    assert np.array_equal(ctx.forward_batch(a + np.uint64(3 * q)), want)          # lazy inputs < 4q
    assert np.array_equal(ctx.inverse_batch(a), oracle.ntt_inverse(q, n, a))
    assert np.array_equal(ctx.inverse_batch(want), a)
    ctx.close()
    # one bit above the bound the library must switch to the u64 flavour by itself
    q46 = oracle.L.oracle_largest_prime_1mod(2 * n, 46)
    ctx = pkg.NttContext(q46, n)
    assert not ctx.uses_f64
    assert np.array_equal(ctx.forward_batch(a % q46), oracle.ntt_forward(q46, n, a % q46))
    ctx.close()


def test_many_random_polynomials_small_prime(pkg, oracle):
    """q = 12289 (the reference's test prime, test_ntt.cpp:13) across every ring degree it supports, 2048 polynomials."""
    for n in [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048]:
        ctx = pkg.NttContext(12289, n)
        batch = max(3, 8192 // n)
        a = oracle.splitmix(n, 12289, batch * n).reshape(batch, n)
        f = ctx.forward_batch(a)
        assert np.array_equal(f, oracle.ntt_forward(12289, n, a))
        assert np.array_equal(ctx.inverse_batch(f), a)
        ctx.close()


@pytest.mark.parametrize("q,n", [(Q12, 256), (Q44, 4096), (Q16, 65536), (0, 4096), (0, 131072)])
def test_outputs_on_the_reduction_boundaries(pkg, oracle, q, n):
    """The final store maps a lazy value K*q + rho to rho with a biased floor (lsr_arith.hpp canonical_f64); the
    dangerous residues are rho = 0 and rho = q-1.  Pin forward OUTPUTS to {0, 1, q-2, q-1, random} by feeding the
    oracle's inverse of that pattern, and likewise inverse outputs.  q = 0 stands for the largest 45-bit prime."""
    if q == 0:
        q = oracle.L.oracle_largest_prime_1mod(2 * n, 45)
    ctx = pkg.NttContext(q, n)
    rng = np.random.default_rng(9)
    choices = np.array([0, 1, q - 2, q - 1], dtype=np.uint64)
    pats = [choices[rng.integers(0, 4, size=n)] for _ in range(3)]
    pats += [np.where(rng.integers(0, 2, size=n) == 0, choices[rng.integers(0, 4, size=n)], oracle.splitmix(3, q, n)).astype(np.uint64)]
    pats += [np.zeros(n, np.uint64), np.full(n, q - 1, np.uint64)]
    want = np.stack(pats)
    assert np.array_equal(ctx.forward_batch(oracle.ntt_inverse(q, n, want)), want)
    assert np.array_equal(ctx.inverse_batch(oracle.ntt_forward(q, n, want)), want)
    # largest lazy intermediates: the all-sum path of every butterfly round carries 2^R times its inputs (the inverse rounds
    # re-centre only the registers that can exceed the next round's input bound, lsr_arith.hpp needs_recentre) — INPUTS at the top
    # of the range in every position / every other position / blocks of 16 and 256
    idx = np.arange(n)
    worst = [np.full(n, q - 1, np.uint64), np.where(idx % 2 == 0, q - 1, 0).astype(np.uint64), np.where(idx % 2 == 1, q - 1, 0).astype(np.uint64),
             np.where((idx // 16) % 2 == 0, q - 1, 1).astype(np.uint64), np.where((idx // 256) % 2 == 0, q - 1, q - 2).astype(np.uint64)]
    worst = np.stack(worst)
    assert np.array_equal(ctx.inverse_batch(worst.copy()), oracle.ntt_inverse(q, n, worst))
    assert np.array_equal(ctx.forward_batch(worst.copy()), oracle.ntt_forward(q, n, worst))
    ctx.close()
