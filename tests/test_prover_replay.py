"""BASELINE config 5 / config 1 plumbing: the Rust prover's call sequence around the commitment kernel, replayed by
tests/prover_replay.py.  CPU part: oracle backend (runs here).  GPU part: the same proofs through the C-ABI must be
bit-identical to the oracle's under the same seeds ("proof bit-exact vs CPU")."""
import json
import os

import numpy as np
import pytest

import prover_replay as pr

Q_TV = 17592186044417      # 2^44 + 1: every test vector and Rust call site (test-vectors/*/params.json:10)
KEY_SEED = 0x1234


def tv2(golden_dir):
    cons = json.load(open(os.path.join(golden_dir, "tv2_constraints.json")))
    c0 = cons["constraints"][0]
    r1cs = pr.R1CS(cons["m"], cons["n"], 1, c0["A"], c0["B"], c0["C"], cons["modular_arithmetic"]["q"])
    return r1cs, cons["verification"]["witness"]


def multiplication_gate():
    """lib.rs:735-742 doc example / r1cs.rs tests: z = [1, 7, 13, 91], a * b = c, l = 2."""
    e = lambda col: [{"row": 0, "col": col, "value": 1}]
    return pr.R1CS(1, 4, 2, e(1), e(2), e(3), Q_TV), [1, 7, 13, 91]


def three_constraints():
    """x*x = y, y*x = z, (z + x + 5)*1 = out with x = 3: a multi-row system so that Q(X) is not constant."""
    a = [{"row": 0, "col": 1, "value": 1}, {"row": 1, "col": 2, "value": 1}, {"row": 2, "col": 3, "value": 1}, {"row": 2, "col": 1, "value": 1},
         {"row": 2, "col": 0, "value": 5}]
    b = [{"row": 0, "col": 1, "value": 1}, {"row": 1, "col": 1, "value": 1}, {"row": 2, "col": 0, "value": 1}]
    c = [{"row": 0, "col": 2, "value": 1}, {"row": 1, "col": 3, "value": 1}, {"row": 2, "col": 4, "value": 1}]
    return pr.R1CS(3, 5, 2, a, b, c, Q_TV), [1, 3, 9, 27, 35]


def oracle_commit(oracle, n=4096, k=2):
    return lambda coeffs, seed: oracle.lwe_commit(Q_TV, n, k, 3.19, KEY_SEED, coeffs, seed)


def test_replay_pieces():
    q = 97
    assert pr.eval_poly([2, 3, 1], 2, q) == 12                       # r1cs.rs:352-360 doc example
    assert pr.lagrange_interpolate([5], q) == [5]
    pts = [3, 10, 40, 7]
    poly = pr.lagrange_interpolate(pts, q)
    assert [pr.eval_poly(poly, i, q) for i in range(4)] == pts
    assert pr.vanishing_poly(3, q) == [0, 2, (-3) % q, 1]            # X(X-1)(X-2)
    assert pr.poly_div_vanishing(pr._poly_mul(pr.vanishing_poly(3, q), [4, 0, 9], q), 3, q) == [4, 0, 9]
    with pytest.raises(ValueError):
        pr.poly_div_vanishing([1, 0, 0, 1], 3, q)


@pytest.mark.parametrize("system", ["tv2", "gate", "three"])
def test_prove_verify_roundtrip_on_the_oracle(oracle, golden_dir, system):
    r1cs, witness = {"tv2": lambda: tv2(golden_dir), "gate": multiplication_gate, "three": three_constraints}[system]()
    assert r1cs.is_satisfied(witness)
    proof = pr.prove_r1cs(r1cs, witness, oracle_commit(oracle), seed=0x5678)
    assert pr.verify_r1cs(proof, witness[: r1cs.l], r1cs)
    assert 0 <= proof["alpha"] < Q_TV and 0 <= proof["beta"] < Q_TV and proof["alpha"] != proof["beta"]
    again = pr.prove_r1cs(r1cs, witness, oracle_commit(oracle), seed=0x5678)
    assert again["alpha"] == proof["alpha"] and np.array_equal(again["commitment_q"], proof["commitment_q"])   # seeded => reproducible
    tampered = dict(proof); tampered["q_alpha"] = (proof["q_alpha"] + 1) % Q_TV
    assert not pr.verify_r1cs(tampered, witness[: r1cs.l], r1cs)
    assert not pr.verify_r1cs(proof, [w + 1 for w in witness[: r1cs.l]], r1cs)      # other public input => other alpha
    bad = list(witness); bad[-1] += 1
    with pytest.raises(ValueError):
        pr.prove_r1cs(r1cs, bad, oracle_commit(oracle), seed=1)


@pytest.mark.gpu
@pytest.mark.parametrize("system", ["tv2", "gate", "three"])
def test_config5_proof_bit_exact_gpu_vs_oracle(pkg, lib, oracle, golden_dir, system):
    import ctypes
    r1cs, witness = {"tv2": lambda: tv2(golden_dir), "gate": multiplication_gate, "three": three_constraints}[system]()
    lctx = pkg.LweContext(pkg.Params(q=Q_TV, n=4096, k=2, sigma=3.19), key_seed=KEY_SEED)
    held = []

    def gpu_commit(coeffs, seed):
        c = pkg.Commitment(lctx, coeffs, seed)       # commitment.rs:31-45 through the mirrored wrapper
        held.append(c)
        return c.as_words()

    gpu = pr.prove_r1cs(r1cs, witness, gpu_commit, seed=0x5678)
    cpu = pr.prove_r1cs(r1cs, witness, oracle_commit(oracle), seed=0x5678)
    assert set(gpu) == set(cpu)
    for key in gpu:
        assert np.array_equal(gpu[key], cpu[key]), key
    assert pr.verify_r1cs(gpu, witness[: r1cs.l], r1cs)
    # the library's own transcript (lsr_fs_challenge) agrees with the replayed Challenge::derive
    public = np.array(witness[: r1cs.l], dtype=np.uint64)
    alpha = ctypes.c_uint64(0); digest = (ctypes.c_uint8 * 32)()
    assert lib.lsr_fs_challenge(public.ctypes.data, public.size, held[0]._p, Q_TV, ctypes.byref(alpha), digest) == 0
    assert alpha.value == gpu["alpha"] and bytes(digest) == gpu["alpha_hash"]
    # and the commitment opens to the quotient coefficients (prove_simple-style check, tests/prover.rs:78-119)
    # (coefficients are 44-bit field elements: the commitment embeds them mod t and opens to those residues only —
    # the reference's decode-and-compare does the same, commitment.cpp:223-226)
    quotient = r1cs.quotient(witness)
    t = lctx.plain_modulus
    assert pkg.verify_opening_with_context(lctx, held[0], [c % t for c in quotient])
    assert pkg.verify_opening_with_context(lctx, held[0], quotient) == all(c < t for c in quotient)
    lctx.close()
