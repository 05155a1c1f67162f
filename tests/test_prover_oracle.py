"""CPU suite: the prover-path oracle (oracle/lsr_prover_oracle.c) replays the reference's own tests for
rust-api/lambda-snark/src/ntt.rs and the constants of rust-api/lambda-snark-core/src/lib.rs, and the product's
host-side number theory for that path agrees with it.  No GPU work here."""
import numpy as np
import pytest

Q = 18446744069414584321          # NTT_MODULUS, lambda-snark-core/src/lib.rs:58
ROOT_2_32 = 1753635133440165772   # NTT_PRIMITIVE_ROOT, lib.rs:78


def test_modulus_and_root_constants(oracle):
    """lib.rs:313-375: q = 2^64 - 2^32 + 1, q - 1 = 2^32 (2^32 - 1), root^(2^31) = -1, root hierarchy."""
    assert oracle.prover_q == Q == 2**64 - 2**32 + 1
    assert (Q - 1) % 2**32 == 0 and (Q - 1) // 2**32 == 2**32 - 1
    assert oracle.L.oracle_prover_root_2_32() == ROOT_2_32 < Q
    assert oracle.L.oracle_powmod(ROOT_2_32, 2**31, Q) == Q - 1 == pow(ROOT_2_32, 2**31, Q)
    for k in range(1, 10):
        w = oracle.prover_omega(1 << k)
        assert w == pow(ROOT_2_32, 2**32 >> k, Q)
        assert pow(w, 1 << k, Q) == 1 and pow(w, 1 << (k - 1), Q) == Q - 1


def test_root_of_unity_like_ntt_rs(oracle):
    """ntt.rs:264-283"""
    assert oracle.prover_omega(2) == Q - 1
    w4 = oracle.prover_omega(4)
    assert pow(w4, 4, Q) == 1 and pow(w4, 2, Q) == Q - 1
    w8 = oracle.prover_omega(8)
    assert pow(w8, 8, Q) == 1 and pow(w8, 4, Q) == Q - 1
    assert oracle.L.oracle_root_of_unity(3, Q, ROOT_2_32) == 0
    assert oracle.L.oracle_root_of_unity(0, Q, ROOT_2_32) == 0


def test_known_answers_of_ntt_rs(oracle):
    """ntt.rs:285-339: the 2-, 4- and 8-point cases."""
    ev = oracle.cyclic_forward([1, 2], Q, oracle.prover_omega(2))
    assert list(ev) == [3, Q - 1]
    assert list(oracle.cyclic_inverse(ev, Q, oracle.prover_omega(2))) == [1, 2]
    ev = oracle.cyclic_forward([1, 2, 3, 4], Q, oracle.prover_omega(4))
    assert ev[0] == 10
    assert list(oracle.cyclic_inverse(ev, Q, oracle.prover_omega(4))) == [1, 2, 3, 4]
    ev = oracle.cyclic_forward([1, 2, 3, 4, 5, 6, 7, 8], Q, oracle.prover_omega(8))
    assert ev[0] == 36
    assert list(oracle.cyclic_inverse(ev, Q, oracle.prover_omega(8))) == [1, 2, 3, 4, 5, 6, 7, 8]


def test_round_trips_of_ntt_rs(oracle):
    """ntt.rs:341-355: n = 2 .. 1024 with coefficients i * 123456789 mod q."""
    for log_n in range(1, 11):
        n = 1 << log_n
        w = oracle.prover_omega(n)
        coeffs = np.array([(i * 123456789) % Q for i in range(n)], dtype=np.uint64)
        ev = oracle.cyclic_forward(coeffs, Q, w)
        assert np.array_equal(oracle.cyclic_inverse(ev, Q, w), coeffs)


def test_linearity_of_ntt_rs(oracle):
    """ntt.rs:357-389"""
    f, g, a, b = [1, 2, 3, 4], [5, 6, 7, 8], 3, 7
    w = oracle.prover_omega(4)
    combo = [(a * x + b * y) % Q for x, y in zip(f, g)]
    lhs = oracle.cyclic_forward(combo, Q, w)
    nf, ng = oracle.cyclic_forward(f, Q, w), oracle.cyclic_forward(g, Q, w)
    assert [int(v) for v in lhs] == [(a * int(x) + b * int(y)) % Q for x, y in zip(nf, ng)]


def test_transform_is_evaluation_at_powers_of_omega(oracle):
    """The definition in ntt.rs:96-110: out[k] = f(omega^k) — checked with Python integers, independent of the C code."""
    rng = np.random.default_rng(11)
    for n in [2, 4, 16, 64, 256]:
        w = oracle.prover_omega(n)
        f = rng.integers(0, Q, size=n, dtype=np.uint64)
        f[0] = Q - 1
        ev = oracle.cyclic_forward(f, Q, w)
        want = [sum(int(c) * pow(w, i * k, Q) for i, c in enumerate(f)) % Q for k in range(n)]
        assert [int(v) for v in ev] == want
        assert np.array_equal(oracle.cyclic_naive(f, Q, w), ev)
    # another field: the commitment's 44-bit prime has 2^13-th roots (r1cs.rs:534-547 lists them)
    q2, n, w = 17592169062401, 256, 5221410271124
    assert pow(w, n // 2, q2) == q2 - 1
    f = rng.integers(0, q2, size=n, dtype=np.uint64)
    assert np.array_equal(oracle.cyclic_naive(f, q2, w), oracle.cyclic_forward(f, q2, w))


def satisfied_instance(rng, m):
    a = rng.integers(0, Q, size=m, dtype=np.uint64)
    b = rng.integers(0, Q, size=m, dtype=np.uint64)
    c = np.array([(int(x) * int(y)) % Q for x, y in zip(a, b)], dtype=np.uint64)
    return a, b, c


def test_quotient_identity_of_r1cs_rs(oracle):
    """r1cs.rs:1723-1777: Q(alpha) Z_H(alpha) = A(alpha) B(alpha) - C(alpha), with Z_H = X^m - 1 on the NTT path."""
    rng = np.random.default_rng(5)
    for m in [1, 2, 4, 8, 32, 128]:
        a, b, c = satisfied_instance(rng, m)
        quot, ln = oracle.quotient(a, b, c)
        assert 1 <= ln <= m                      # r1cs.rs:1690-1692: q.len() <= num_constraints
        assert not quot[ln:].any()
        w = oracle.prover_omega(m)
        pa, pb, pc = (oracle.cyclic_inverse(v, Q, w) for v in (a, b, c))
        for alpha in [12345, int(rng.integers(0, Q, dtype=np.uint64))]:
            zh = (pow(alpha, m, Q) - 1) % Q
            lhs = oracle.eval_poly(quot[:ln], alpha, Q) * zh % Q
            rhs = (oracle.eval_poly(pa, alpha, Q) * oracle.eval_poly(pb, alpha, Q) - oracle.eval_poly(pc, alpha, Q)) % Q
            assert lhs == rhs


def test_quotient_rejects_unsatisfied_instances(oracle):
    """poly_div_vanishing's Err (r1cs.rs:1050-1054) -> length 0"""
    rng = np.random.default_rng(6)
    for m in [1, 2, 8, 64]:
        a, b, c = satisfied_instance(rng, m)
        c[m // 2] = (int(c[m // 2]) + 1) % Q
        assert oracle.quotient(a, b, c)[1] == 0
    # all-zero numerator: Ok([0])
    z = np.zeros(8, dtype=np.uint64)
    quot, ln = oracle.quotient(z, z, z)
    assert ln == 1 and not quot.any()
    # degree of A*B below m: A, B constants => numerator a0*b0 - c0 = 0 => [0]
    ones = np.full(8, 3, dtype=np.uint64)
    quot, ln = oracle.quotient(ones, ones, np.full(8, 9, dtype=np.uint64))
    assert ln == 1 and not quot.any()


def test_sparse_mul_vec_like_sparse_matrix_rs(oracle):
    """sparse_matrix.rs:259-289 incl. the `% modulus` of both operands; r1cs.rs:1266-1301 evaluations."""
    gate = ([(0, 1, 1)], [(0, 2, 1)], [(0, 3, 1)])
    w = [1, 7, 13, 91]
    assert [int(oracle.sparse_mul_vec(m, 1, w, Q)[0]) for m in gate] == [7, 13, 91]
    two_a, two_b, two_c = [(0, 1, 1), (1, 3, 1)], [(0, 2, 1), (1, 4, 1)], [(0, 3, 1), (1, 5, 1)]
    w = [1, 2, 3, 6, 4, 24]
    assert list(oracle.sparse_mul_vec(two_a, 2, w, Q)) == [2, 6] and list(oracle.sparse_mul_vec(two_b, 2, w, Q)) == [3, 4]
    assert list(oracle.sparse_mul_vec(two_c, 2, w, Q)) == [6, 24]
    big = [(0, 0, 2**64 - 1), (0, 1, Q + 5)]
    v = [2**64 - 2, 3]
    assert int(oracle.sparse_mul_vec(big, 1, v, Q)[0]) == (((2**64 - 1) % Q) * ((2**64 - 2) % Q) + 5 * 3) % Q


def test_product_host_math_for_the_prover_path(lib):
    assert lib.lsr_prover_modulus() == Q and lib.lsr_prover_root_2_32() == ROOT_2_32
    for k in range(0, 33):
        assert lib.lsr_prover_root_of_unity(1 << k) == pow(ROOT_2_32, 2**32 >> k, Q)
    assert lib.lsr_prover_root_of_unity(0) == 0 and lib.lsr_prover_root_of_unity(12) == 0 and lib.lsr_prover_root_of_unity(1 << 33) == 0
    # argument contract without touching a device
    assert not lib.lsr_cyclic_ntt_context_create(Q, 3, 0, -1)
    assert not lib.lsr_cyclic_ntt_context_create(Q, 1 << 18, 0, -1)
    assert not lib.lsr_cyclic_ntt_context_create(Q, 8, 5, -1)                 # 5 is not an 8th root of unity
    assert not lib.lsr_cyclic_ntt_context_create(17592169062401, 8, 0, -1)    # no default root outside NTT_MODULUS
    assert not lib.lsr_cyclic_ntt_context_create(Q - 2, 8, 0, -1)
    assert not lib.lsr_quotient_plan_create(0, -1) and not lib.lsr_quotient_plan_create(3, -1) and not lib.lsr_quotient_plan_create(1 << 18, -1)
    assert lib.lsr_cyclic_ntt_forward_batch(None, None, 1) == -1
    assert lib.lsr_quotient_batch(None, None, None, None, 1, None, None) == -1
    assert lib.lsr_quotient_plan_size(None) == 0
    assert not lib.lsr_r1cs_prover_create(None, None, None, -1)
    assert lib.lsr_r1cs_prover_num_constraints(None) == 0 and lib.lsr_r1cs_prover_num_variables(None) == 0
    assert lib.lsr_r1cs_quotient_batch(None, None, 1, None, None) == -1
    assert lib.lsr_r1cs_constraint_evals_batch(None, None, 1, None, None, None) == -1
    lib.lsr_r1cs_prover_free(None)
    lib.lsr_quotient_plan_free(None)
    if lib.lsr_device_count() == 0:
        assert not lib.lsr_quotient_plan_create(8, -1)                         # fails loudly: no CPU fallback
        assert b"no HIP device" in lib.lsr_last_error()
