"""Host-side mirror of the reference's safe wrappers over the lambda-snark-sys C-ABI.

The reference's host language is Rust (absent from this image), so the wrappers that sit above the
C-ABI are mirrored here in Python with the same names, argument meaning and error behaviour:

* ``LweContext``  <-> rust-api/lambda-snark/src/context.rs:14-76   (``LweContext::new`` -> ``lwe_context_create``)
* ``Commitment``  <-> rust-api/lambda-snark/src/commitment.rs:31-121 (``new``, ``clone``, ``linear_combine``, ``as_bytes``)
* ``verify_opening_with_context`` <-> rust-api/lambda-snark/src/opening.rs:160-222
* ``NttContext``  <-> the ``ntt_*`` symbols (only exercised by cpp-core/tests/test_ntt.cpp in the reference)
* ``CyclicNtt`` / ``QuotientPlan`` <-> rust-api/lambda-snark/src/ntt.rs:117-233 and r1cs.rs:474-506 (prover path)

All arithmetic happens in liblambda_snark_core.so (HIP, gfx950).  Nothing here computes; there is no
CPU fallback.  (The directory name has a hyphen; load it through ``__graft_entry__.load_package()``.)
"""
import ctypes

import numpy as np

from . import _abi
from ._abi import PROFILE_RING_B, PROFILE_SCALAR_A, LweCommitment, LweOpening, PublicParams

__all__ = [
    "NttContext", "LweContext", "Commitment", "Params", "CoreError", "verify_opening_with_context",
    "sample_gaussian", "verify_openings_batch", "verify_openings_words", "PublicParams", "PROFILE_RING_B", "PROFILE_SCALAR_A",
    "CyclicNtt", "QuotientPlan", "R1csProver", "compute_root_of_unity", "NTT_MODULUS", "NTT_PRIMITIVE_ROOT",
]


class CoreError(RuntimeError):
    """Mirror of lambda_snark_core::Error::{FfiError, CommitmentFailed, InvalidDimensions}."""


def _u64_array(values, name="array"):
    arr = np.ascontiguousarray(values, dtype=np.uint64)
    if arr.ndim == 0:
        raise ValueError(f"{name} must be an array")
    return arr


class NttContext:
    """RAII handle over ``NttContext*`` (cpp-core/include/lambda_snark/ntt.h:25-41)."""

    def __init__(self, q, n, device=-1):
        self._lib = _abi.lib()
        self.q, self.n = int(q), int(n)
        self._h = self._lib.lsr_ntt_context_create_on(self.q, self.n, device) if device >= 0 else self._lib.ntt_context_create(self.q, self.n)
        if not self._h:
            raise CoreError(f"ntt_context_create({q}, {n}) returned NULL: {_abi.last_error()}")

    @property
    def handle(self):
        return self._h

    @property
    def root(self):
        return self._lib.lsr_ntt_context_root(self._h)

    @property
    def uses_f64(self):
        return bool(self._lib.lsr_ntt_context_uses_f64(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ntt_context_free(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # --- reference single-polynomial entry points (host buffers, in place) ---
    def forward(self, coeffs):
        a = _u64_array(coeffs).copy()
        if self._lib.ntt_forward(self._h, a.ctypes.data, a.size) != 0:
            raise CoreError("ntt_forward failed")
        return a

    def inverse(self, evals):
        a = _u64_array(evals).copy()
        if self._lib.ntt_inverse(self._h, a.ctypes.data, a.size) != 0:
            raise CoreError("ntt_inverse failed")
        return a

    def mul_pointwise(self, a, b):
        a, b = _u64_array(a), _u64_array(b)
        out = np.zeros_like(a)
        self._lib.ntt_mul_pointwise(self._h, out.ctypes.data, a.ctypes.data, b.ctypes.data, a.size)
        return out

    # --- batched host entry points ([batch][n]) ---
    def forward_batch(self, polys):
        a = _u64_array(polys).copy().reshape(-1, self.n)
        if self._lib.ntt_forward_batch(self._h, a.ctypes.data, a.shape[0]) != 0:
            raise CoreError("ntt_forward_batch failed: " + _abi.last_error())
        return a

    def inverse_batch(self, polys):
        a = _u64_array(polys).copy().reshape(-1, self.n)
        if self._lib.ntt_inverse_batch(self._h, a.ctypes.data, a.shape[0]) != 0:
            raise CoreError("ntt_inverse_batch failed: " + _abi.last_error())
        return a

    # --- device-resident entry points (raw device pointers, e.g. torch.Tensor.data_ptr()) ---
    def forward_device(self, dptr, batch, stream=0):
        if self._lib.lsr_ntt_forward_batch_device(self._h, dptr, batch, stream) != 0:
            raise CoreError("lsr_ntt_forward_batch_device failed: " + _abi.last_error())

    def inverse_device(self, dptr, batch, stream=0):
        if self._lib.lsr_ntt_inverse_batch_device(self._h, dptr, batch, stream) != 0:
            raise CoreError("lsr_ntt_inverse_batch_device failed: " + _abi.last_error())

    def mul_pointwise_device(self, dres, da, db, count, stream=0):
        if self._lib.lsr_ntt_mul_pointwise_device(self._h, dres, da, db, count, stream) != 0:
            raise CoreError("lsr_ntt_mul_pointwise_device failed: " + _abi.last_error())


class Params:
    """Mirror of lambda_snark_core::Params (rust-api/lambda-snark-core/src/lib.rs:136-196), RingB profile."""

    def __init__(self, security_level=128, q=17592186044417, n=4096, k=2, sigma=3.19, profile=PROFILE_RING_B):
        self.security_level, self.q, self.n, self.k, self.sigma, self.profile = security_level, q, n, k, sigma, profile

    def to_ffi(self):
        # context.rs:18-42: ScalarA is sent as ring_degree = 1, module_rank = 1
        if self.profile == PROFILE_SCALAR_A:
            return PublicParams(PROFILE_SCALAR_A, self.security_level, self.q, 1, 1, self.sigma)
        return PublicParams(PROFILE_RING_B, self.security_level, self.q, self.n, self.k, self.sigma)


class LweContext:
    """Mirror of lambda_snark::LweContext (context.rs:14-76)."""

    def __init__(self, params, key_seed=None, device=-1):
        self._lib = _abi.lib()
        self.params = params
        ffi = params.to_ffi()
        if key_seed is None and device < 0:
            self._h = self._lib.lwe_context_create(ctypes.byref(ffi))
        else:
            self._h = self._lib.lsr_lwe_context_create_seeded(ctypes.byref(ffi), int(key_seed or 0), device)
        if not self._h:
            raise CoreError("FfiError: lwe_context_create returned NULL")   # context.rs:46-48

    @property
    def handle(self):
        return self._h

    def modulus(self):
        """context.rs:90 returns params.q; the ring modulus actually used is ``commit_modulus``."""
        return self.params.q

    @property
    def commit_modulus(self):
        return self._lib.lsr_lwe_modulus(self._h)

    @property
    def plain_modulus(self):
        return self._lib.lsr_lwe_plain_modulus(self._h)

    @property
    def ring_degree(self):
        return self._lib.lsr_lwe_ring_degree(self._h)

    @property
    def module_rank(self):
        return self._lib.lsr_lwe_module_rank(self._h)

    @property
    def pipeline(self):
        """``lsr_lwe_pipeline``: which kernels this context's commitments and openings run on ("tile", "fused", "fused-matvec", "general")."""
        return self._lib.lsr_lwe_pipeline(self._h).decode()

    @property
    def commitment_words(self):
        return self._lib.lsr_lwe_commitment_words(self._h)

    def commit_keys(self, messages, seeds):
        """``lsr_lwe_commit_keys``: the per-commitment 256-bit stream keys exactly as ``lwe_commit`` derives them (seed 0: fresh entropy),
        [batch][4] uint64 — the input of the device-resident ``lsr_lwe_commit_rows_device``."""
        messages = np.ascontiguousarray(messages, dtype=np.uint64)
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        if messages.ndim != 2 or messages.shape[0] != seeds.size:
            raise ValueError("messages must be [batch][msg_len] with one seed per row")
        keys = np.zeros((seeds.size, 4), dtype=np.uint64)
        rc = self._lib.lsr_lwe_commit_keys(self._h, messages.ctypes.data if messages.size else None, messages.shape[1], seeds.size, seeds.ctypes.data,
                                           keys.ctypes.data)
        if rc != 0:
            raise CoreError("lsr_lwe_commit_keys failed: " + _abi.last_error())
        return keys

    def commit_keys_device(self, d_messages, msg_len, seeds, d_keys, stream):
        """``lsr_lwe_commit_keys_device``: the same keys derived on the device from device-resident messages (seeds: host array, all
        non-zero), asynchronous on `stream`."""
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        if self._lib.lsr_lwe_commit_keys_device(self._h, d_messages, msg_len, seeds.size, seeds.ctypes.data, d_keys, stream) != 0:
            raise CoreError("lsr_lwe_commit_keys_device failed: " + _abi.last_error())

    def commit_rows_device(self, d_messages, msg_len, batch, d_keys, d_rows, stream):
        """``lsr_lwe_commit_rows_device``: device pointers (ints) in, wire rows out, asynchronous on `stream`."""
        if self._lib.lsr_lwe_commit_rows_device(self._h, d_messages, msg_len, batch, d_keys, d_rows, stream) != 0:
            raise CoreError("CommitmentFailed: " + _abi.last_error())

    def verify_rows_device(self, d_rows, d_messages, msg_len, count, d_results, stream):
        """``lsr_lwe_verify_rows_device``: int32 verdicts (1 / 0 / -1 as ``lwe_verify_opening``) per row, asynchronous on `stream`."""
        if self._lib.lsr_lwe_verify_rows_device(self._h, d_rows, d_messages, msg_len, count, d_results, stream) != 0:
            raise CoreError("VerificationFailed: " + _abi.last_error())

    def public_matrix(self):
        k, n = self.module_rank, self.ring_degree
        a = np.zeros((k, k, n), dtype=np.uint64)
        if self._lib.lsr_lwe_public_matrix(self._h, a.ctypes.data) != 0:
            raise CoreError("lsr_lwe_public_matrix failed")
        return a

    def replicate(self, device=-1):
        """``lsr_lwe_context_replicate``: the same context (same keys) on another device, for sharded runs."""
        twin = object.__new__(LweContext)
        twin._lib, twin.params = self._lib, self.params
        twin._h = self._lib.lsr_lwe_context_replicate(self._h, device)
        if not twin._h:
            raise CoreError("FfiError: lsr_lwe_context_replicate returned NULL")
        return twin

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lwe_context_free(self._h)
            self._h = None

    __del__ = close


class PinnedArray:
    """A uint64 host array in page-locked memory (``lsr_host_alloc_pinned``): the single gather target of a sharded call."""

    def __init__(self, shape):
        self._lib = _abi.lib()
        self.shape = tuple(int(x) for x in shape)
        count = int(np.prod(self.shape))
        self._p = self._lib.lsr_host_alloc_pinned(max(count, 1) * 8)
        if not self._p:
            raise MemoryError("lsr_host_alloc_pinned failed")
        self.array = np.ctypeslib.as_array(ctypes.cast(self._p, ctypes.POINTER(ctypes.c_uint64)), shape=(count,)).reshape(self.shape)

    @property
    def ptr(self):
        return self._p

    def close(self):
        if getattr(self, "_p", None):
            self.array = None
            self._lib.lsr_host_free_pinned(self._p)
            self._p = None

    __del__ = close


def shard_bounds(batch, shards, index):
    """``lsr_shard_bounds`` -> (first, count) of shard `index`."""
    first, count = ctypes.c_size_t(0), ctypes.c_size_t(0)
    _abi.lib().lsr_shard_bounds(batch, shards, index, ctypes.byref(first), ctypes.byref(count))
    return first.value, count.value


def _handles(ctxs):
    return (ctypes.c_void_p * len(ctxs))(*[c.handle for c in ctxs])


def sharded_ntt(ctxs, polys, inverse=False):
    """``lsr_ntt_forward_batch_sharded`` / ``_inverse_``: `polys` is ONE host array [batch][n], transformed in place."""
    lib = _abi.lib()
    fn = lib.lsr_ntt_inverse_batch_sharded if inverse else lib.lsr_ntt_forward_batch_sharded
    if fn(_handles(ctxs), len(ctxs), polys.ctypes.data, polys.shape[0]) != 0:
        raise CoreError("sharded transform failed: " + _abi.last_error())
    return polys


def sharded_commit_words(ctxs, messages, seeds, out=None):
    """``lsr_lwe_commit_batch_flat_sharded``: rows [batch][words] written into `out` (a host array, e.g. ``PinnedArray.array``)."""
    lib = _abi.lib()
    msgs = _u64_array(messages, "messages")
    sd = _u64_array(seeds, "seeds")
    if out is None:
        out = np.zeros((msgs.shape[0], lib.lsr_lwe_commitment_words(ctxs[0].handle)), dtype=np.uint64)
    if lib.lsr_lwe_commit_batch_flat_sharded(_handles(ctxs), len(ctxs), msgs.ctypes.data, msgs.shape[1], msgs.shape[0], sd.ctypes.data, out.ctypes.data) != 0:
        raise CoreError("CommitmentFailed: " + _abi.last_error())
    return out


def sharded_matvec(ctxs, d_r_ptrs, d_e1_ptrs, batch, host_u):
    """``lsr_mlwe_matvec_batch_sharded``: device-resident inputs per shard, gather into the host array `host_u`;
    returns (longest kernel time of a shard's slice, longest wall time until a slice is in host memory), seconds."""
    lib = _abi.lib()
    r = (ctypes.c_void_p * len(ctxs))(*d_r_ptrs)
    e = (ctypes.c_void_p * len(ctxs))(*d_e1_ptrs)
    seconds = (ctypes.c_double * 2)()
    if lib.lsr_mlwe_matvec_batch_sharded(_handles(ctxs), len(ctxs), r, e, batch, host_u.ctypes.data, seconds) != 0:
        raise CoreError("sharded matvec failed: " + _abi.last_error())
    return seconds[0], seconds[1]


def sharded_matvec_stats(ctxs, d_r_ptrs, d_e1_ptrs, batch, host_u):
    """``lsr_mlwe_matvec_batch_sharded_stats``: the same call, returning [(kernel seconds, wall seconds until gathered)] per shard."""
    lib = _abi.lib()
    r = (ctypes.c_void_p * len(ctxs))(*d_r_ptrs)
    e = (ctypes.c_void_p * len(ctxs))(*d_e1_ptrs)
    stats = (ctypes.c_double * (2 * len(ctxs)))()
    if lib.lsr_mlwe_matvec_batch_sharded_stats(_handles(ctxs), len(ctxs), r, e, batch, host_u.ctypes.data, stats) != 0:
        raise CoreError("sharded matvec failed: " + _abi.last_error())
    return [(stats[2 * g], stats[2 * g + 1]) for g in range(len(ctxs))]


class Commitment:
    """Mirror of lambda_snark::Commitment (commitment.rs:31-121)."""

    def __init__(self, ctx, message=None, seed=0, _raw=None):
        self._lib = _abi.lib()
        self._ctx = ctx
        if _raw is not None:
            self._p = _raw
            return
        # commitment.rs:33-36: every coefficient is reduced mod ctx.modulus() before the call
        msg = np.array([int(m) % ctx.modulus() for m in message], dtype=np.uint64)
        self._p = self._lib.lwe_commit(ctx.handle, msg.ctypes.data, msg.size, int(seed))
        if not self._p:
            raise CoreError("CommitmentFailed")   # commitment.rs:40-42

    @classmethod
    def batch(cls, ctx, messages, seeds):
        """``lwe_commit_batch``: messages [batch][msg_len], seeds [batch]."""
        lib = _abi.lib()
        msgs = np.ascontiguousarray(messages, dtype=np.uint64)
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        out = (ctypes.POINTER(LweCommitment) * msgs.shape[0])()
        if lib.lwe_commit_batch(ctx.handle, msgs.ctypes.data, msgs.shape[1], msgs.shape[0], seeds.ctypes.data, out) != 0:
            raise CoreError("CommitmentFailed: " + _abi.last_error())
        return [cls(ctx, _raw=out[i]) for i in range(msgs.shape[0])]

    @staticmethod
    def batch_words(ctx, messages, seeds):
        """``lsr_lwe_commit_batch_flat``: the same commitments as ``batch`` as one [batch][words] array (no per-commitment
        allocation)."""
        lib = _abi.lib()
        msgs = _u64_array(messages, "messages")
        if msgs.ndim != 2:
            raise ValueError("messages must be [batch][msg_len]")
        sd = _u64_array(seeds, "seeds")
        out = np.zeros((msgs.shape[0], lib.lsr_lwe_commitment_words(ctx.handle)), dtype=np.uint64)
        if lib.lsr_lwe_commit_batch_flat(ctx.handle, msgs.ctypes.data, msgs.shape[1], msgs.shape[0], sd.ctypes.data, out.ctypes.data) != 0:
            raise CoreError("CommitmentFailed: " + _abi.last_error())
        return out

    def clone(self):
        p = self._lib.lwe_commitment_clone(self._p)
        if not p:
            raise CoreError("lwe_commitment_clone returned NULL")   # commitment.rs:22-24 panics
        return Commitment(self._ctx, _raw=p)

    @staticmethod
    def linear_combine(ctx, commitments, coeffs):
        """commitment.rs:60-84 (coefficients reduced mod ctx.modulus() first, :66-69)."""
        if len(commitments) == 0:
            raise ValueError("no commitments provided")                     # commitment.rs:66-68
        if len(commitments) != len(coeffs):
            raise ValueError("commitments/coeffs length mismatch")          # commitment.rs:70-74
        lib = _abi.lib()
        arr = (ctypes.POINTER(LweCommitment) * len(commitments))(*[c._p if c is not None else None for c in commitments])
        cf = np.array([int(c) % ctx.modulus() for c in coeffs], dtype=np.uint64)
        p = lib.lwe_linear_combine(ctx.handle, arr, cf.ctypes.data, len(commitments))
        if not p:
            raise CoreError("CommitmentFailed")   # commitment.rs:80-82
        return Commitment(ctx, _raw=p)

    def as_words(self):
        """commitment.rs:88-93: the flat u64 words (hashed word-by-word by the Fiat–Shamir transcript)."""
        c = self._p.contents
        return np.ctypeslib.as_array(c.data, shape=(c.len,)).copy()

    def as_bytes(self):
        return self.as_words().tobytes()

    def serialize(self):
        """The bincode form of the Rust type (commitment.rs:112-121, pinned by tests/serialization.rs:128-153): a u64
        element count followed by the words, little endian.  There is no deserialisation (it needs an LweContext)."""
        words = self.as_words()
        return int(words.size).to_bytes(8, "little") + words.astype("<u8").tobytes()

    def __len__(self):
        return self._p.contents.len

    def free(self):
        if getattr(self, "_p", None):
            self._lib.lwe_commitment_free(self._p)
            self._p = None

    __del__ = free


def wide_modulus(ring_degree):
    """``lsr_lwe_wide_modulus``: the 60-bit NTT prime to pass as ``Params.q`` for reference-range linear combinations."""
    return int(_abi.lib().lsr_lwe_wide_modulus(int(ring_degree)))


def words_to_limbs(words, limb_bits=16, limbs_per_word=4):
    """``lsr_words_to_limbs``: little-endian limbs of field elements wider than the plaintext modulus, so that a commitment
    binds them in full (a message word >= t is embedded mod t and never opens — commitment.cpp:152,223-226)."""
    lib = _abi.lib()
    w = _u64_array(words, "words").ravel()
    out = np.zeros(w.size * limbs_per_word, dtype=np.uint64)
    if lib.lsr_words_to_limbs(w.ctypes.data, w.size, limb_bits, limbs_per_word, out.ctypes.data) != out.size:
        raise ValueError("invalid limb parameters")
    return out


def verify_opening_with_context(ctx, commitment, message, randomness=None):
    """opening.rs:160-222 -> ``lwe_verify_opening``; returns True/False, raises on -1."""
    lib = _abi.lib()
    msg = np.array([int(m) % ctx.modulus() for m in message], dtype=np.uint64)   # opening.rs:198-201
    rnd = np.zeros(1, dtype=np.uint64) if randomness is None else _u64_array(randomness)
    opening = LweOpening(rnd.ctypes.data_as(_abi.u64p), rnd.size)
    rc = lib.lwe_verify_opening(ctx.handle, commitment._p, msg.ctypes.data, msg.size, ctypes.byref(opening))
    if rc < 0:
        raise CoreError("lwe_verify_opening returned -1")
    return rc == 1


def verify_openings_batch(ctx, commitments, messages):
    """``lwe_verify_opening_batch``: messages [count][msg_len] (reduced mod ctx.modulus() like opening.rs:198-201);
    returns a list of 1 / 0 / -1."""
    lib = _abi.lib()
    msgs = np.ascontiguousarray([[int(m) % ctx.modulus() for m in row] for row in messages], dtype=np.uint64)
    arr = (ctypes.POINTER(LweCommitment) * len(commitments))(*[c._p if c is not None else None for c in commitments])
    out = np.zeros(len(commitments), dtype=np.int32)
    if lib.lwe_verify_opening_batch(ctx.handle, arr, msgs.ctypes.data, msgs.shape[1] if msgs.ndim == 2 else 0, len(commitments), out.ctypes.data) != 0:
        raise CoreError("lwe_verify_opening_batch failed: " + _abi.last_error())
    return [int(x) for x in out]


def verify_openings_words(ctx, words, messages):
    """``lsr_lwe_verify_opening_batch_flat``: commitments as rows of one array (``Commitment.batch_words``)."""
    lib = _abi.lib()
    rows = _u64_array(words, "words")
    msgs = _u64_array(messages, "messages")
    count = rows.shape[0] if rows.ndim == 2 else 0
    out = np.zeros(count, dtype=np.int32)
    if count and lib.lsr_lwe_verify_opening_batch_flat(ctx.handle, rows.ctypes.data, msgs.ctypes.data, msgs.shape[1] if msgs.ndim == 2 else 0, count,
                                                       out.ctypes.data) != 0:
        raise CoreError("lsr_lwe_verify_opening_batch_flat failed: " + _abi.last_error())
    return [int(x) for x in out]


def sample_gaussian(length, sigma, seed=None, domain=16, index=0):
    """``sample_gaussian`` (utils.h:27) or its seeded twin; returns int64 samples."""
    lib = _abi.lib()
    out = np.zeros(length, dtype=np.uint64)
    if seed is None:
        rc = lib.sample_gaussian(out.ctypes.data, length, float(sigma))
    else:
        rc = lib.lsr_sample_gaussian_seeded(out.ctypes.data, length, float(sigma), int(seed), int(domain), int(index))
    if rc != 0:
        raise CoreError("sample_gaussian failed")
    return out.view(np.int64)


# ---- prover-side polynomial path (include/lambda_snark/prover.h) -------------------------------------------------
NTT_MODULUS = 18446744069414584321          # rust-api/lambda-snark-core/src/lib.rs:58
NTT_PRIMITIVE_ROOT = 1753635133440165772    # lib.rs:78


def compute_root_of_unity(n, modulus=NTT_MODULUS, primitive_root=NTT_PRIMITIVE_ROOT):
    """rust-api/lambda-snark/src/ntt.rs:226-233 (host integer arithmetic only)."""
    if n <= 0 or n & (n - 1) or n > 1 << 32:
        raise ValueError("n must be power of 2, n <= 2^32")
    return pow(primitive_root, (1 << 32) // n, modulus)


class CyclicNtt:
    """The transform pair of rust-api/lambda-snark/src/ntt.rs: ``forward(coeffs)`` = ``ntt_forward(coeffs, modulus, omega)``
    (natural order in and out), ``inverse(evals)`` = ``ntt_inverse``.  One handle per (modulus, n, omega)."""

    def __init__(self, n, modulus=NTT_MODULUS, omega=0, device=-1):
        self._lib = _abi.lib()
        self.n, self.modulus = int(n), int(modulus)
        self._h = self._lib.lsr_cyclic_ntt_context_create(self.modulus, self.n, int(omega), device)
        if not self._h:
            raise CoreError(f"lsr_cyclic_ntt_context_create({modulus}, {n}) returned NULL: {_abi.last_error()}")

    @property
    def handle(self):
        return self._h

    @property
    def omega(self):
        return self._lib.lsr_ntt_context_root(self._h)

    def _run(self, fn, values):
        arr = _u64_array(values).copy()
        if arr.size % self.n:
            raise ValueError("length must be a multiple of n")
        if fn(self._h, arr.ctypes.data, arr.size // self.n) != 0:
            raise CoreError("cyclic NTT failed: " + _abi.last_error())
        return arr

    def forward(self, coeffs):
        return self._run(self._lib.lsr_cyclic_ntt_forward_batch, coeffs)

    def inverse(self, evals):
        return self._run(self._lib.lsr_cyclic_ntt_inverse_batch, evals)

    def close(self):
        if self._h:
            self._lib.ntt_context_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class QuotientPlan:
    """Steps 3-6 of ``R1CS::compute_quotient_poly`` (rust-api/lambda-snark/src/r1cs.rs:474-506) on the NTT path
    (``should_use_ntt``: m a power of two, modulus NTT_MODULUS), for batches of independent instances."""

    def __init__(self, m, device=-1):
        self._lib = _abi.lib()
        self.m = int(m)
        self._h = self._lib.lsr_quotient_plan_create(self.m, device)
        if not self._h:
            raise CoreError(f"lsr_quotient_plan_create({m}) returned NULL: {_abi.last_error()}")

    @property
    def handle(self):
        return self._h

    def quotient_batch(self, a_evals, b_evals, c_evals):
        """-> (coefficients [batch][m], lengths [batch]); length 0 marks the reference's Err (remainder non-zero)."""
        a, b, c = (_u64_array(v) for v in (a_evals, b_evals, c_evals))
        if not (a.size == b.size == c.size) or a.size % self.m:
            raise ValueError("a, b, c must hold the same number of m-word instances")
        batch = a.size // self.m
        quot = np.zeros((batch, self.m), dtype=np.uint64)
        lens = np.zeros(batch, dtype=np.uint32)
        if batch and self._lib.lsr_quotient_batch(self._h, a.ctypes.data, b.ctypes.data, c.ctypes.data, batch, quot.ctypes.data, lens.ctypes.data) != 0:
            raise CoreError("lsr_quotient_batch failed: " + _abi.last_error())
        return quot, lens

    def compute_quotient_poly(self, a_evals, b_evals, c_evals):
        """One instance, with the reference's return convention: the trimmed coefficient list, or CoreError."""
        quot, lens = self.quotient_batch(a_evals, b_evals, c_evals)
        if lens[0] == 0:
            raise CoreError("Polynomial division by Z_H: remainder non-zero (witness invalid)")   # r1cs.rs:1050-1054
        return quot[0, :lens[0]].copy()

    def quotient_device(self, da, db, dc, batch, dquot, dlen, stream=0):
        if self._lib.lsr_quotient_batch_device(self._h, da, db, dc, batch, dquot, dlen, stream) != 0:
            raise CoreError("lsr_quotient_batch_device failed: " + _abi.last_error())

    def close(self):
        if self._h:
            self._lib.lsr_quotient_plan_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class R1csProver:
    """``R1CS`` restricted to what the prover's hot loop needs (rust-api/lambda-snark/src/r1cs.rs:88-137, 296-304, 474-506):
    the three matrices on the device, ``compute_constraint_evals`` and ``compute_quotient_poly`` for batches of witnesses.
    ``a``, ``b``, ``c`` are lists of ``(row, col, value)`` entries of m x n matrices; modulus is NTT_MODULUS, m = 2^k."""

    def __init__(self, m, n, a, b, c, device=-1):
        self._lib = _abi.lib()
        self.m, self.n = int(m), int(n)
        keep, mats = [], []
        for entries in (a, b, c):
            arr = (_abi.SparseEntry * max(1, len(entries)))()
            for i, (row, col, value) in enumerate(entries):
                arr[i] = _abi.SparseEntry(int(row), int(col), int(value))
            keep.append(arr)
            mats.append(_abi.SparseMatrix(ctypes.cast(arr, ctypes.POINTER(_abi.SparseEntry)), len(entries), self.m, self.n))
        self._h = self._lib.lsr_r1cs_prover_create(ctypes.byref(mats[0]), ctypes.byref(mats[1]), ctypes.byref(mats[2]), device)
        if not self._h:
            raise CoreError(f"lsr_r1cs_prover_create(m={m}, n={n}) returned NULL: {_abi.last_error()}")

    def _witnesses(self, witnesses):
        w = _u64_array(witnesses)
        if w.size % self.n:
            raise ValueError("Witness length must equal n")      # r1cs.rs:297
        return w, w.size // self.n

    def compute_constraint_evals(self, witnesses):
        w, batch = self._witnesses(witnesses)
        out = [np.zeros((batch, self.m), dtype=np.uint64) for _ in range(3)]
        if batch and self._lib.lsr_r1cs_constraint_evals_batch(self._h, w.ctypes.data, batch, *(o.ctypes.data for o in out)) != 0:
            raise CoreError("lsr_r1cs_constraint_evals_batch failed: " + _abi.last_error())
        return tuple(out)

    def quotient_batch(self, witnesses):
        w, batch = self._witnesses(witnesses)
        quot = np.zeros((batch, self.m), dtype=np.uint64)
        lens = np.zeros(batch, dtype=np.uint32)
        if batch and self._lib.lsr_r1cs_quotient_batch(self._h, w.ctypes.data, batch, quot.ctypes.data, lens.ctypes.data) != 0:
            raise CoreError("lsr_r1cs_quotient_batch failed: " + _abi.last_error())
        return quot, lens

    def compute_quotient_poly(self, witness):
        quot, lens = self.quotient_batch(witness)
        if lens[0] == 0:
            raise CoreError("Witness does not satisfy R1CS constraints")          # r1cs.rs:477-480
        return quot[0, :lens[0]].copy()

    def close(self):
        if self._h:
            self._lib.lsr_r1cs_prover_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

