#!/usr/bin/env python3
"""Independent generator of ntt_check_values.json — pure Python integers, no oracle, no library.

PARITY UNPINNED: these numbers are NOT outputs of the reference.  cpp-core's own NTT assertions are a round trip and
2*3 = 6 (/root/reference/cpp-core/tests/test_ntt.cpp:47-81); it ships no forward known-answer vector and Microsoft SEAL
(where the arithmetic lives, cpp-core/CMakeLists.txt:75) is not in this image.  What this script restates is the DEFINITION
the survey derives from SEAL 4.1's published semantics:

    psi    = the numerically smallest primitive 2n-th root of unity mod q
    out[i] = sum_j a[j] * psi^((2*bitrev(i, log2 n) + 1) * j)  mod q          (bit-reversed evaluation order)
    a[j]   = splitmix64_j(seed = 0xDEADBEEF) mod q

computed two ways that share nothing with oracle/lsr_oracle.c: a recursive even/odd split of the evaluation (exact
Python ints) for the whole vector, and direct Horner evaluation on sampled indices.  The ROOTS_OF_UNITY block restates
rust-api/lambda-snark/src/r1cs.rs:527-547 (w_m = 3^((q-1)/m) mod q) and checks each against pow().

    python tests/golden/make_ntt_check_values.py            # verify the committed file
    python tests/golden/make_ntt_check_values.py --write    # regenerate it
"""
import hashlib
import json
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
MASK = (1 << 64) - 1
SEED = 0xDEADBEEF
CASES = [(12289, 256), (17592169062401, 4096), (17592182243329, 65536),
         # further sizes (round 2): the first two-pass degree, the 5-bit strided round, a 60-bit modulus (u64 flavour)
         (17592186028033, 8192), (17592180539393, 131072), (1152921504606830593, 1024)]
# moduli that are "the largest `bits`-bit prime = 1 (mod 2n)" (what SEAL's prime generator returns): re-derived below
LARGEST_PRIME_CASES = {17592182243329: (44, 65536), 17592186028033: (44, 8192), 17592180539393: (44, 131072), 1152921504606830593: (60, 1024)}
# rust-api/lambda-snark/src/r1cs.rs:534-547 (data of the reference, restated)
ROOTS_Q = 17592169062401
ROOTS_TABLE = [[4, 981206394875], [8, 4268641988953], [16, 9400386778549], [32, 15690227524213], [64, 8332322609789],
               [128, 9249819209096], [256, 5221410271124], [512, 9594533594163], [1024, 11016271016603],
               [2048, 14373677444369], [4096, 11176258803537], [8192, 9037003627149]]


def splitmix_stream(seed, count):
    x = seed
    for _ in range(count):
        x = (x + 0x9E3779B97F4A7C15) & MASK
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
        yield z ^ (z >> 31)


def prime_factors(m):
    out, d = [], 2
    while d * d <= m:
        if m % d == 0:
            out.append(d)
            while m % d == 0:
                m //= d
        d += 1 if d == 2 else 2
    if m > 1:
        out.append(m)
    return out


def is_prime(m):
    """Deterministic Miller-Rabin for m < 3.3 * 10^24 (the first 13 primes as bases)."""
    if m < 2:
        return False
    small = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41)
    for p in small:
        if m % p == 0:
            return m == p
    d, r = m - 1, 0
    while d % 2 == 0:
        d //= 2
        r += 1
    for a in small:
        x = pow(a, d, m)
        if x in (1, m - 1):
            continue
        for _ in range(r - 1):
            x = x * x % m
            if x == m - 1:
                break
        else:
            return False
    return True


def largest_prime_1_mod(bits, modulus):
    """Largest prime below 2^bits that is 1 modulo `modulus`."""
    c = (1 << bits) - 1
    c -= (c - 1) % modulus
    while not is_prime(c):
        c -= modulus
    return c


def minimal_primitive_2n_root(q, n):
    """Smallest element of order exactly 2n in (Z/q)^*: all of them are g^((q-1)/2n * odd) for a generator g."""
    assert (q - 1) % (2 * n) == 0
    factors = prime_factors(q - 1)
    g = next(c for c in range(2, q) if all(pow(c, (q - 1) // f, q) != 1 for f in factors))
    base = pow(g, (q - 1) // (2 * n), q)            # one primitive 2n-th root
    step = base * base % q
    best, cur = base, base
    for _ in range(n - 1):                           # the other odd powers
        cur = cur * step % q
        if cur < best:
            best = cur
    assert pow(best, n, q) == q - 1                  # order exactly 2n
    return best


def bitrev(i, bits):
    return int(format(i, "0%db" % bits)[::-1], 2) if bits else 0


def evaluate_all(a, psi, q):
    """[a(psi^(2r+1)) for r in range(n)] in NATURAL order of r, by the even/odd split
    a(x) = E(x^2) + x O(x^2): the points psi^(2r+1) square to the odd powers of psi^2, a half-size problem."""
    n = len(a)
    if n == 1:
        return [a[0] % q]
    even = evaluate_all(a[0::2], psi * psi % q, q)   # E at (psi^2)^(2s+1), s < n/2
    odd = evaluate_all(a[1::2], psi * psi % q, q)
    out = [0] * n
    x = psi
    step = psi * psi % q
    for r in range(n):
        s = r % (n // 2)                             # (psi^(2r+1))^2 = (psi^2)^(2r+1), exponent mod n => s = r mod n/2
        out[r] = (even[s] + x * odd[s]) % q
        x = x * step % q
    return out


def build():
    cases = []
    for q, (pbits, pn) in LARGEST_PRIME_CASES.items():
        assert largest_prime_1_mod(pbits, 2 * pn) == q, (q, pbits, pn)
    for q, n in CASES:
        assert is_prime(q) and (q - 1) % (2 * n) == 0, (q, n)
        bits = n.bit_length() - 1
        a = [z % q for z in splitmix_stream(SEED, n)]
        psi = minimal_primitive_2n_root(q, n)
        natural = evaluate_all(a, psi, q)
        out = [natural[bitrev(i, bits)] for i in range(n)]
        for i in (0, 1, 2, n // 3, n - 1):           # direct Horner evaluation of the defining formula
            x = pow(psi, 2 * bitrev(i, bits) + 1, q)
            acc = 0
            for c in reversed(a):
                acc = (acc * x + c) % q
            assert acc == out[i], (q, n, i)
        digest = hashlib.sha256(struct.pack("<%dQ" % n, *out)).hexdigest()
        cases.append({"q": q, "n": n, "psi": psi, "a": a[:3], "ntt": out[:3], "sha256": digest})
    for m, w in ROOTS_TABLE:
        assert pow(3, (ROOTS_Q - 1) // m, ROOTS_Q) == w and pow(w, m // 2, ROOTS_Q) == ROOTS_Q - 1, m
    return {
        "source": "tests/golden/make_ntt_check_values.py (independent pure-Python restatement of SURVEY.md section 8(a) N2/N3; "
                  "NOT produced by the reference: parity unpinned); input a[i] = splitmix64_i(seed=0xDEADBEEF) mod q",
        "splitmix_seed": SEED,
        "cases": cases,
        "roots_of_unity": {"source": "rust-api/lambda-snark/src/r1cs.rs:527-547: w_m = 3^((q-1)/m) mod q, q = 17592169062401",
                           "q": ROOTS_Q, "generator": 3, "table": ROOTS_TABLE},
    }


def main():
    sys.setrecursionlimit(10000)
    fresh = build()
    path = os.path.join(HERE, "ntt_check_values.json")
    if "--write" in sys.argv:
        with open(path, "w") as f:
            json.dump(fresh, f, indent=1)
        print("wrote", path)
        return 0
    with open(path) as f:
        committed = json.load(f)
    same = committed["cases"] == fresh["cases"] and committed["roots_of_unity"]["table"] == fresh["roots_of_unity"]["table"]
    print("committed file matches the independent computation" if same else "MISMATCH")
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
