"""Test harness: replays, over the C-ABI, the call sequence the Rust prover makes around the commitment kernel.

The Rust toolchain is absent, so BASELINE configs 1 and 5 ("via lambda-snark-cli", "through Rust FFI") are exercised by
restating the host-side steps that sit either side of the FFI calls:

* R1CS evaluation                      rust-api/lambda-snark/src/r1cs.rs:296-305  (compute_constraint_evals)
* Lagrange interpolation on {0..m-1}   r1cs.rs:808-828 (baseline path; the NTT path needs modulus == NTT_MODULUS, r1cs.rs:386-389)
* quotient Q = (A_z B_z - C_z) / Z_H   r1cs.rs:474-503, :846-863, :995-1065
* Fiat–Shamir transcript               challenge.rs:102-134
* Horner evaluation                    r1cs.rs:362-373
* prove_r1cs / verify_r1cs             lib.rs:747-809, :1016-1095

`commit` is injected: the GPU library through the mirrored `Commitment` wrapper, or the CPU oracle.  Test
infrastructure only — nothing here ships.
"""
import hashlib

import numpy as np

FS_TAG = b"LAMBDA-SNARK-R-FS-v1"


def challenge_derive(public_inputs, commitment_words, modulus):
    """challenge.rs:102-134: SHA3-256(tag || len || inputs || len || words), alpha = LE64(h[0..8]) mod q."""
    h = hashlib.sha3_256()
    h.update(FS_TAG)
    h.update(len(public_inputs).to_bytes(8, "little"))
    for v in public_inputs:
        h.update(int(v).to_bytes(8, "little"))
    h.update(len(commitment_words).to_bytes(8, "little"))
    h.update(np.asarray(commitment_words, dtype="<u8").tobytes())
    digest = h.digest()
    return int.from_bytes(digest[:8], "little") % modulus, digest


def _mat_vec(entries, m, z, q):
    out = [0] * m
    for e in entries:
        out[e["row"]] = (out[e["row"]] + (e["value"] % q) * z[e["col"]]) % q
    return out


def _poly_mul(a, b, q):
    if not a or not b:
        return [0]
    r = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            r[i + j] = (r[i + j] + x * y) % q
    return r


def _poly_sub(a, b, q):
    n = max(len(a), len(b))
    a = a + [0] * (n - len(a))
    b = b + [0] * (n - len(b))
    return [(x - y) % q for x, y in zip(a, b)]


def lagrange_interpolate(evals, q):
    """r1cs.rs:808-828 — domain H = {0, 1, ..., m-1}."""
    m = len(evals)
    result = [0] * m
    for i in range(m):
        basis, denom = [1], 1
        for j in range(m):
            if j == i:
                continue
            basis = _poly_mul(basis, [(-j) % q, 1], q)
            denom = denom * ((i - j) % q) % q
        inv = pow(denom, -1, q)
        for k in range(m):
            result[k] = (result[k] + evals[i] * basis[k] % q * inv) % q
    return result


def vanishing_poly(m, q):
    poly = [1]
    for i in range(m):
        poly = _poly_mul(poly, [(-i) % q, 1], q)
    return poly


def poly_div_vanishing(num, m, q):
    """r1cs.rs:995-1065 (baseline Z_H = prod (X - i))."""
    if not num:
        return [0]
    div = vanishing_poly(m, q)
    rem = list(num)
    if len(rem) - 1 < m:
        if any(rem):
            raise ValueError("remainder non-zero (witness invalid)")
        return [0]
    quot = [0] * (len(rem) - m)
    for i in range(len(quot) - 1, -1, -1):
        c = rem[i + m] % q          # divisor is monic
        quot[i] = c
        for j, d in enumerate(div):
            rem[i + j] = (rem[i + j] - c * d) % q
    if any(rem):
        raise ValueError("remainder non-zero (witness invalid)")
    while len(quot) > 1 and quot[-1] == 0:
        quot.pop()
    return quot


def eval_poly(poly, x, q):
    """r1cs.rs:362-373."""
    result, power = 0, 1
    for c in poly:
        result = (result + c * power) % q
        power = power * x % q
    return result


def eval_vanishing(m, x, q):
    r = 1
    for i in range(m):
        r = r * ((x - i) % q) % q
    return r


class R1CS:
    def __init__(self, m, n, l, a, b, c, modulus):
        self.m, self.n, self.l, self.a, self.b, self.c, self.q = m, n, l, a, b, c, modulus

    def evals(self, z):
        return _mat_vec(self.a, self.m, z, self.q), _mat_vec(self.b, self.m, z, self.q), _mat_vec(self.c, self.m, z, self.q)

    def is_satisfied(self, z):
        a, b, c = self.evals(z)
        return all((x * y - w) % self.q == 0 for x, y, w in zip(a, b, c))

    def quotient(self, z):
        if not self.is_satisfied(z):
            raise ValueError("Witness does not satisfy R1CS constraints")
        a, b, c = (lagrange_interpolate(e, self.q) for e in self.evals(z))
        return poly_div_vanishing(_poly_sub(_poly_mul(a, b, self.q), c, self.q), self.m, self.q)


def prove_r1cs(r1cs, witness, commit, seed):
    """lib.rs:747-809.  `commit(coeffs, seed)` returns the commitment's uint64 words."""
    q = r1cs.q
    q_coeffs = r1cs.quotient(witness)
    words = commit([c % q for c in q_coeffs], seed)          # commitment.rs:33-36 reduces mod ctx.modulus()
    public = list(witness[: r1cs.l])
    alpha, h_alpha = challenge_derive(public, words, q)
    beta, h_beta = challenge_derive([alpha], words, q)
    a_poly, b_poly, c_poly = (lagrange_interpolate(e, q) for e in r1cs.evals(witness))
    ev = lambda poly, x: eval_poly(poly, x, q)
    return {
        "commitment_q": words, "alpha": alpha, "alpha_hash": h_alpha, "beta": beta, "beta_hash": h_beta,
        "q_alpha": ev(q_coeffs, alpha), "q_beta": ev(q_coeffs, beta),
        "a_z_alpha": ev(a_poly, alpha), "b_z_alpha": ev(b_poly, alpha), "c_z_alpha": ev(c_poly, alpha),
        "a_z_beta": ev(a_poly, beta), "b_z_beta": ev(b_poly, beta), "c_z_beta": ev(c_poly, beta),
        "opening_alpha": ev(q_coeffs, alpha), "opening_beta": ev(q_coeffs, beta),
    }


def verify_r1cs(proof, public_inputs, r1cs):
    """lib.rs:1016-1095 (never calls the FFI)."""
    q = r1cs.q
    alpha, _ = challenge_derive(public_inputs, proof["commitment_q"], q)
    if alpha != proof["alpha"]:
        return False
    beta, _ = challenge_derive([alpha], proof["commitment_q"], q)
    if beta != proof["beta"]:
        return False
    for x, tag in ((alpha, "alpha"), (beta, "beta")):
        lhs = proof[f"q_{tag}"] * eval_vanishing(r1cs.m, x, q) % q
        rhs = (proof[f"a_z_{tag}"] * proof[f"b_z_{tag}"] - proof[f"c_z_{tag}"]) % q
        if lhs != rhs or proof[f"opening_{tag}"] != proof[f"q_{tag}"]:
            return False
    return True
