"""ctypes binding of the CPU oracle (oracle/liblsr_oracle.so).  TEST INFRASTRUCTURE: imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product package."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "liblsr_oracle.so")

u64, u32, vp, sz, dbl, ci = ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_int


class Oracle:
    def __init__(self, lib):
        L = self.L = lib
        L.oracle_mulmod.restype = u64; L.oracle_mulmod.argtypes = [u64, u64, u64]
        L.oracle_powmod.restype = u64; L.oracle_powmod.argtypes = [u64, u64, u64]
        L.oracle_is_prime.restype = ci; L.oracle_is_prime.argtypes = [u64]
        L.oracle_minimal_primitive_root.restype = u64; L.oracle_minimal_primitive_root.argtypes = [u64, u64]
        L.oracle_largest_prime_1mod.restype = u64; L.oracle_largest_prime_1mod.argtypes = [u64, ci]
        L.oracle_ntt_create.restype = vp; L.oracle_ntt_create.argtypes = [u64, u32]
        L.oracle_ntt_free.argtypes = [vp]
        L.oracle_ntt_root.restype = u64; L.oracle_ntt_root.argtypes = [vp]
        L.oracle_ntt_tables.argtypes = [vp, vp, vp]
        L.oracle_ntt_forward.restype = ci; L.oracle_ntt_forward.argtypes = [vp, vp, u32]
        L.oracle_ntt_inverse.restype = ci; L.oracle_ntt_inverse.argtypes = [vp, vp, u32]
        L.oracle_ntt_mul_pointwise.argtypes = [vp, vp, vp, vp, u32]
        L.oracle_ntt_forward_batch.restype = ci; L.oracle_ntt_forward_batch.argtypes = [vp, vp, sz]
        L.oracle_ntt_inverse_batch.restype = ci; L.oracle_ntt_inverse_batch.argtypes = [vp, vp, sz]
        L.oracle_ntt_forward_naive.argtypes = [vp, vp, vp]
        L.oracle_splitmix_fill.argtypes = [u64, u64, vp, sz]
        L.oracle_chacha20_block.argtypes = [vp, u32, vp, vp]
        L.oracle_stream_words.argtypes = [u64, u32, u64, u64, vp, sz]
        L.oracle_gaussian_cdf.restype = sz; L.oracle_gaussian_cdf.argtypes = [dbl, vp, sz]
        L.oracle_sample_gaussian.restype = ci; L.oracle_sample_gaussian.argtypes = [vp, sz, dbl]
        L.oracle_sample_gaussian_seeded.restype = ci; L.oracle_sample_gaussian_seeded.argtypes = [vp, sz, dbl, u64, u32, u64]
        L.oracle_context_keys.argtypes = [u64, vp, vp, vp]
        L.oracle_commit_key.argtypes = [u64, vp, vp, sz, u64, vp]
        L.oracle_lwe_select_modulus.restype = u64; L.oracle_lwe_select_modulus.argtypes = [u64, u32]
        L.oracle_lwe_create.restype = vp; L.oracle_lwe_create.argtypes = [u64, u32, u32, dbl, u64]
        L.oracle_lwe_free.argtypes = [vp]
        L.oracle_lwe_q.restype = u64; L.oracle_lwe_q.argtypes = [vp]
        L.oracle_lwe_t.restype = u64; L.oracle_lwe_t.argtypes = [vp]
        L.oracle_lwe_commit_words.restype = sz; L.oracle_lwe_commit_words.argtypes = [vp]
        L.oracle_lwe_commit.restype = ci; L.oracle_lwe_commit.argtypes = [vp, vp, sz, u64, vp]
        L.oracle_lwe_verify.restype = ci; L.oracle_lwe_verify.argtypes = [vp, vp, sz, vp, sz]
        L.oracle_lwe_linear_combine.restype = ci; L.oracle_lwe_linear_combine.argtypes = [vp, vp, vp, vp, sz, vp]
        L.oracle_lwe_public_matrix.argtypes = [vp, vp]
        L.oracle_mlwe_matvec.argtypes = [vp, u32, vp, vp, vp, vp]
        L.oracle_prover_modulus.restype = u64; L.oracle_prover_modulus.argtypes = []
        L.oracle_prover_root_2_32.restype = u64; L.oracle_prover_root_2_32.argtypes = []
        L.oracle_root_of_unity.restype = u64; L.oracle_root_of_unity.argtypes = [u64, u64, u64]
        L.oracle_cyclic_ntt_forward.restype = ci; L.oracle_cyclic_ntt_forward.argtypes = [vp, sz, u64, u64]
        L.oracle_cyclic_ntt_inverse.restype = ci; L.oracle_cyclic_ntt_inverse.argtypes = [vp, sz, u64, u64]
        L.oracle_cyclic_ntt_naive.argtypes = [vp, vp, sz, u64, u64]
        L.oracle_eval_poly.restype = u64; L.oracle_eval_poly.argtypes = [vp, sz, u64, u64]
        L.oracle_sparse_mul_vec.argtypes = [vp, vp, vp, sz, vp, u64, vp, sz]
        L.oracle_quotient_ntt_path.restype = sz; L.oracle_quotient_ntt_path.argtypes = [vp, vp, vp, sz, u64, u64, vp]
        self._ntt = {}
        self._lwe = {}

    # ---- NTT ----
    def ntt_handle(self, q, n):
        key = (int(q), int(n))
        if key not in self._ntt:
            self._ntt[key] = self.L.oracle_ntt_create(q, n)
        return self._ntt[key]

    def ntt_create_ok(self, q, n):
        h = self.L.oracle_ntt_create(q, n)
        if h:
            self.L.oracle_ntt_free(h)
        return bool(h)

    def splitmix(self, seed, q, count):
        out = np.zeros(count, dtype=np.uint64)
        self.L.oracle_splitmix_fill(seed, q, out.ctypes.data, count)
        return out

    def ntt_forward(self, q, n, polys):
        a = np.ascontiguousarray(polys, dtype=np.uint64).copy()
        assert self.L.oracle_ntt_forward_batch(self.ntt_handle(q, n), a.ctypes.data, a.size // n) == 0
        return a

    def ntt_inverse(self, q, n, polys):
        a = np.ascontiguousarray(polys, dtype=np.uint64).copy()
        assert self.L.oracle_ntt_inverse_batch(self.ntt_handle(q, n), a.ctypes.data, a.size // n) == 0
        return a

    def ntt_forward_naive(self, q, n, poly):
        a = np.ascontiguousarray(poly, dtype=np.uint64)
        out = np.zeros(n, dtype=np.uint64)
        self.L.oracle_ntt_forward_naive(self.ntt_handle(q, n), a.ctypes.data, out.ctypes.data)
        return out

    def mul_pointwise(self, q, n, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint64); b = np.ascontiguousarray(b, dtype=np.uint64)
        out = np.zeros_like(a)
        self.L.oracle_ntt_mul_pointwise(self.ntt_handle(q, n), out.ctypes.data, a.ctypes.data, b.ctypes.data, a.size)
        return out

    def root(self, q, n):
        return self.L.oracle_ntt_root(self.ntt_handle(q, n))

    # ---- sampler / stream ----
    def chacha20_block(self, key_words, counter, nonce_words):
        k = np.array(key_words, dtype=np.uint32); nn = np.array(nonce_words, dtype=np.uint32)
        out = np.zeros(16, dtype=np.uint32)
        self.L.oracle_chacha20_block(k.ctypes.data, counter, nn.ctypes.data, out.ctypes.data)
        return out

    def stream_words(self, seed, domain, index, first, count):
        out = np.zeros(count, dtype=np.uint64)
        self.L.oracle_stream_words(seed, domain, index, first, out.ctypes.data, count)
        return out

    def gaussian_cdf(self, sigma):
        buf = np.zeros(8192, dtype=np.uint64)
        cnt = self.L.oracle_gaussian_cdf(sigma, buf.ctypes.data, buf.size)
        return buf[:cnt].copy()

    def sample_gaussian(self, length, sigma):
        out = np.zeros(length, dtype=np.uint64)
        rc = self.L.oracle_sample_gaussian(out.ctypes.data if length else None, length, sigma)
        return rc, out.view(np.int64)

    def sample_gaussian_seeded(self, length, sigma, seed, domain, index):
        out = np.zeros(length, dtype=np.uint64)
        assert self.L.oracle_sample_gaussian_seeded(out.ctypes.data, length, sigma, seed, domain, index) == 0
        return out.view(np.int64)

    # ---- commitment ----
    def lwe_handle(self, q, n, k, sigma, key_seed):
        key = (int(q), int(n), int(k), float(sigma), int(key_seed))
        if key not in self._lwe:
            self._lwe[key] = self.L.oracle_lwe_create(q, n, k, sigma, key_seed)
        return self._lwe[key]

    def context_keys(self, key_seed):
        pub, sec, ident = (np.zeros(8, dtype=np.uint32), np.zeros(8, dtype=np.uint32), np.zeros(4, dtype=np.uint32))
        self.L.oracle_context_keys(key_seed, pub.ctypes.data, sec.ctypes.data, ident.ctypes.data)
        return pub, sec, ident

    def commit_key(self, seed, ident, msg, t):
        m = np.array([int(x) for x in msg] or [0], dtype=np.uint64)
        ident = np.ascontiguousarray(ident, dtype=np.uint32)
        out = np.zeros(8, dtype=np.uint32)
        self.L.oracle_commit_key(seed, ident.ctypes.data, m.ctypes.data, len(msg), t, out.ctypes.data)
        return out

    def lwe_commit(self, q, n, k, sigma, key_seed, msg, seed):
        h = self.lwe_handle(q, n, k, sigma, key_seed)
        assert h
        words = self.L.oracle_lwe_commit_words(h)
        out = np.zeros(words, dtype=np.uint64)
        vals = [int(x) for x in msg]
        m = np.array(vals or [0], dtype=np.uint64)          # never hand the oracle a NULL pointer for an empty message
        assert self.L.oracle_lwe_commit(h, m.ctypes.data, len(vals), seed, out.ctypes.data) == 0
        return out

    def lwe_verify(self, q, n, k, sigma, key_seed, comm_words, msg):
        h = self.lwe_handle(q, n, k, sigma, key_seed)
        c = np.ascontiguousarray(comm_words, dtype=np.uint64)
        m = np.array([int(x) for x in msg], dtype=np.uint64)
        return self.L.oracle_lwe_verify(h, c.ctypes.data, c.size, m.ctypes.data, m.size)

    def lwe_linear_combine(self, q, n, k, sigma, key_seed, comms, coeffs):
        h = self.lwe_handle(q, n, k, sigma, key_seed)
        arrs = [np.ascontiguousarray(c, dtype=np.uint64) for c in comms]
        ptrs = (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        lens = (ctypes.c_size_t * len(arrs))(*[a.size for a in arrs])
        cf = np.array([int(c) for c in coeffs], dtype=np.uint64)
        out = np.zeros(self.L.oracle_lwe_commit_words(h), dtype=np.uint64)
        rc = self.L.oracle_lwe_linear_combine(h, ptrs, lens, cf.ctypes.data, len(arrs), out.ctypes.data)
        return rc, out

    def lwe_public_matrix(self, q, n, k, sigma, key_seed):
        h = self.lwe_handle(q, n, k, sigma, key_seed)
        qq = self.L.oracle_lwe_q(h)
        a = np.zeros((k, k, n), dtype=np.uint64)
        self.L.oracle_lwe_public_matrix(h, a.ctypes.data)
        return qq, a

    def mlwe_matvec(self, q, n, k, a_hat, r, e1):
        a_hat = np.ascontiguousarray(a_hat, dtype=np.uint64); r = np.ascontiguousarray(r, dtype=np.uint64)
        e1p = None
        if e1 is not None:
            e1 = np.ascontiguousarray(e1, dtype=np.uint64); e1p = e1.ctypes.data
        u = np.zeros(k * n, dtype=np.uint64)
        self.L.oracle_mlwe_matvec(self.ntt_handle(q, n), k, a_hat.ctypes.data, r.ctypes.data, e1p, u.ctypes.data)
        return u.reshape(k, n)


    # ---- prover-side polynomial path (rust-api/lambda-snark/src/{ntt,r1cs}.rs) ----
    @property
    def prover_q(self):
        return self.L.oracle_prover_modulus()

    def prover_omega(self, n, q=None):
        return self.L.oracle_root_of_unity(n, q or self.prover_q, self.L.oracle_prover_root_2_32())

    def cyclic_forward(self, values, q, omega):
        a = np.ascontiguousarray(values, dtype=np.uint64).copy()
        assert self.L.oracle_cyclic_ntt_forward(a.ctypes.data, a.size, q, omega) == 0
        return a

    def cyclic_inverse(self, values, q, omega):
        a = np.ascontiguousarray(values, dtype=np.uint64).copy()
        assert self.L.oracle_cyclic_ntt_inverse(a.ctypes.data, a.size, q, omega) == 0
        return a

    def cyclic_naive(self, values, q, omega):
        a = np.ascontiguousarray(values, dtype=np.uint64)
        out = np.zeros_like(a)
        self.L.oracle_cyclic_ntt_naive(a.ctypes.data, out.ctypes.data, a.size, q, omega)
        return out

    def eval_poly(self, poly, x, q):
        a = np.ascontiguousarray(poly, dtype=np.uint64)
        return self.L.oracle_eval_poly(a.ctypes.data, a.size, x, q)

    def sparse_mul_vec(self, entries, n_rows, v, q):
        """entries: list of (row, col, value)"""
        rows = np.array([e[0] for e in entries] or [0], dtype=np.uint32); cols = np.array([e[1] for e in entries] or [0], dtype=np.uint32)
        vals = np.array([e[2] for e in entries] or [0], dtype=np.uint64)
        vec = np.ascontiguousarray(v, dtype=np.uint64)
        out = np.zeros(n_rows, dtype=np.uint64)
        self.L.oracle_sparse_mul_vec(rows.ctypes.data, cols.ctypes.data, vals.ctypes.data, len(entries), vec.ctypes.data, q, out.ctypes.data, n_rows)
        return out

    def quotient(self, a_evals, b_evals, c_evals):
        """-> (coefficients padded to m words, trimmed length; 0 = remainder non-zero)"""
        a = np.ascontiguousarray(a_evals, dtype=np.uint64); b = np.ascontiguousarray(b_evals, dtype=np.uint64)
        c = np.ascontiguousarray(c_evals, dtype=np.uint64)
        out = np.zeros(a.size, dtype=np.uint64)
        ln = self.L.oracle_quotient_ntt_path(a.ctypes.data, b.ctypes.data, c.ctypes.data, a.size, self.prover_q,
                                             self.L.oracle_prover_root_2_32(), out.ctypes.data)
        return out, ln


_cached = None


def load():
    global _cached
    if _cached is None:
        if not os.path.exists(LIB):
            subprocess.check_call(["make"], cwd=os.path.join(ROOT, "oracle"))
        _cached = Oracle(ctypes.CDLL(LIB))
    return _cached
