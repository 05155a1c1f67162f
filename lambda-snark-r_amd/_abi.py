"""ctypes binding of liblambda_snark_core.so — the stub a maintainer of lambda-snark-sys would mirror.

The library is the product; this file only declares its C-ABI (include/lambda_snark/*.h).  There is no
CPU fallback: if the shared library is missing the import fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product library, always: no environment variable redirects the loader (an experiment build of the same sources,
# csrc/Makefile VARIANT=..., is loaded explicitly by the tool that measures it: use_library(path) before the first call).
LIB_PATH = os.path.join(_HERE, "lib", "liblambda_snark_core.so")

u64 = ctypes.c_uint64
u32 = ctypes.c_uint32
c_int = ctypes.c_int
c_size = ctypes.c_size_t
c_double = ctypes.c_double
vp = ctypes.c_void_p
u64p = ctypes.POINTER(ctypes.c_uint64)


class PublicParams(ctypes.Structure):
    """reference cpp-core/include/lambda_snark/types.h:60-67"""
    _fields_ = [
        ("profile", ctypes.c_int),
        ("security_level", ctypes.c_uint32),
        ("modulus", ctypes.c_uint64),
        ("ring_degree", ctypes.c_uint32),
        ("module_rank", ctypes.c_uint32),
        ("sigma", ctypes.c_double),
    ]


class LweCommitment(ctypes.Structure):
    """reference types.h:36-39"""
    _fields_ = [("data", u64p), ("len", ctypes.c_size_t)]


class LweOpening(ctypes.Structure):
    """reference types.h:44-47"""
    _fields_ = [("randomness", u64p), ("rand_len", ctypes.c_size_t)]


class SparseEntry(ctypes.Structure):
    """reference cpp-core/include/lambda_snark/r1cs.h:38-42"""
    _fields_ = [("row", ctypes.c_uint32), ("col", ctypes.c_uint32), ("value", ctypes.c_uint64)]


class SparseMatrix(ctypes.Structure):
    """reference r1cs.h:49-54"""
    _fields_ = [("entries", ctypes.POINTER(SparseEntry)), ("n_entries", ctypes.c_size_t), ("n_rows", ctypes.c_uint32), ("n_cols", ctypes.c_uint32)]


class R1CSWitness(ctypes.Structure):
    """reference r1cs.h:76-79"""
    _fields_ = [("values", u64p), ("len", ctypes.c_size_t)]


PROFILE_SCALAR_A = 0
PROFILE_RING_B = 1

# every symbol declared in include/lambda_snark/*.h: name -> (restype, argtypes)
SIGNATURES = {
    # ntt.h
    "ntt_context_create": (vp, [u64, u32]),
    "ntt_context_free": (None, [vp]),
    "ntt_forward": (c_int, [vp, vp, u32]),
    "ntt_inverse": (c_int, [vp, vp, u32]),
    "ntt_mul_pointwise": (None, [vp, vp, vp, vp, u32]),
    # commitment.h
    "lwe_context_create": (vp, [ctypes.POINTER(PublicParams)]),
    "lwe_context_free": (None, [vp]),
    "lwe_commit": (ctypes.POINTER(LweCommitment), [vp, vp, c_size, u64]),
    "lwe_commitment_free": (None, [ctypes.POINTER(LweCommitment)]),
    "lwe_commitment_clone": (ctypes.POINTER(LweCommitment), [ctypes.POINTER(LweCommitment)]),
    "lwe_verify_opening": (c_int, [vp, ctypes.POINTER(LweCommitment), vp, c_size, ctypes.POINTER(LweOpening)]),
    "lwe_linear_combine": (ctypes.POINTER(LweCommitment), [vp, ctypes.POINTER(ctypes.POINTER(LweCommitment)), vp, c_size]),
    # utils.h
    "sample_gaussian": (c_int, [vp, c_size, c_double]),
    # batch.h
    "lsr_device_count": (c_int, []),
    "lsr_last_error": (ctypes.c_char_p, []),
    "lsr_version": (ctypes.c_char_p, []),
    "lsr_ntt_context_create_on": (vp, [u64, u32, c_int]),
    "lsr_ntt_context_device": (c_int, [vp]),
    "lsr_ntt_context_root": (u64, [vp]),
    "lsr_ntt_context_uses_f64": (c_int, [vp]),
    "lsr_set_arith_mode": (None, [c_int]),
    "ntt_forward_batch": (c_int, [vp, vp, c_size]),
    "ntt_inverse_batch": (c_int, [vp, vp, c_size]),
    "ntt_mul_pointwise_batch": (c_int, [vp, vp, vp, vp, c_size]),
    "lsr_ntt_forward_batch_device": (c_int, [vp, vp, c_size, vp]),
    "lsr_ntt_inverse_batch_device": (c_int, [vp, vp, c_size, vp]),
    "lsr_ntt_mul_pointwise_device": (c_int, [vp, vp, vp, vp, c_size, vp]),
    "lsr_sample_gaussian_seeded": (c_int, [vp, c_size, c_double, u64, u32, u64]),
    "lsr_gaussian_cdf": (c_size, [c_double, vp, c_size]),
    "lsr_lwe_context_create_seeded": (vp, [ctypes.POINTER(PublicParams), u64, c_int]),
    "lsr_lwe_modulus": (u64, [vp]),
    "lsr_lwe_plain_modulus": (u64, [vp]),
    "lsr_lwe_ring_degree": (u32, [vp]),
    "lsr_lwe_module_rank": (u32, [vp]),
    "lsr_lwe_commitment_words": (c_size, [vp]),
    "lsr_lwe_ntt_context": (vp, [vp]),
    "lsr_lwe_public_matrix": (c_int, [vp, vp]),
    "lwe_commit_batch": (c_int, [vp, vp, c_size, c_size, vp, ctypes.POINTER(ctypes.POINTER(LweCommitment))]),
    "lsr_lwe_commit_batch_flat": (c_int, [vp, vp, c_size, c_size, vp, vp]),
    "lwe_verify_opening_batch": (c_int, [vp, ctypes.POINTER(ctypes.POINTER(LweCommitment)), vp, c_size, c_size, vp]),
    "lsr_lwe_verify_opening_batch_flat": (c_int, [vp, vp, vp, c_size, c_size, vp]),
    "lsr_mlwe_matvec_batch_device": (c_int, [vp, vp, vp, vp, c_size, vp, vp]),
    "lsr_lwe_sample_blinding_device": (c_int, [vp, vp, c_size, vp, vp]),
    "lsr_shard_bounds": (None, [c_size, c_int, c_int, vp, vp]),
    "lsr_host_alloc_pinned": (vp, [c_size]),
    "lsr_host_free_pinned": (None, [vp]),
    "lsr_lwe_context_replicate": (vp, [vp, c_int]),
    "lsr_ntt_forward_batch_sharded": (c_int, [vp, c_int, vp, c_size]),
    "lsr_ntt_inverse_batch_sharded": (c_int, [vp, c_int, vp, c_size]),
    "lsr_lwe_commit_batch_flat_sharded": (c_int, [vp, c_int, vp, c_size, c_size, vp, vp]),
    "lsr_mlwe_matvec_batch_sharded": (c_int, [vp, c_int, vp, vp, c_size, vp, vp]),
    "lsr_mlwe_matvec_batch_sharded_stats": (c_int, [vp, c_int, vp, vp, c_size, vp, vp]),
    "lsr_words_to_limbs": (c_size, [vp, c_size, ctypes.c_uint, ctypes.c_uint, vp]),
    "lsr_fill_splitmix_device": (c_int, [vp, c_size, c_size, u64, u64, vp]),
    "lsr_fs_challenge": (c_int, [vp, c_size, ctypes.POINTER(LweCommitment), u64, vp, vp]),
    "lsr_fs_challenge_batch_flat": (c_int, [vp, c_size, vp, c_size, c_size, u64, vp, vp, ctypes.c_uint]),
    "lsr_fs_challenge_batch_device": (c_int, [vp, c_size, vp, c_size, c_size, u64, vp, vp, vp]),
    "lsr_lwe_commit_batch_flat_device": (c_int, [vp, vp, c_size, c_size, vp, vp]),
    "lsr_lwe_commit_keys": (c_int, [vp, vp, c_size, c_size, vp, vp]),
    "lsr_lwe_commit_keys_device": (c_int, [vp, vp, c_size, c_size, vp, vp, vp]),
    "lsr_lwe_commit_rows_device": (c_int, [vp, vp, c_size, c_size, vp, vp, vp]),
    "lsr_lwe_verify_rows_device": (c_int, [vp, vp, vp, c_size, c_size, vp, vp]),
    "lsr_lwe_pipeline": (ctypes.c_char_p, [vp]),
    "lsr_lwe_wide_modulus": (ctypes.c_uint64, [ctypes.c_uint32]),
    "lsr_minimal_primitive_root": (u64, [u64, u32]),
    # r1cs.h (SEAL/NTL-free shim, host only)
    "lambda_snark_r1cs_create": (c_int, [ctypes.POINTER(SparseMatrix), ctypes.POINTER(SparseMatrix), ctypes.POINTER(SparseMatrix), u64, ctypes.POINTER(vp)]),
    "lambda_snark_r1cs_validate_witness": (c_int, [vp, ctypes.POINTER(R1CSWitness), ctypes.POINTER(ctypes.c_bool)]),
    "lambda_snark_r1cs_free": (None, [vp]),
    "lambda_snark_r1cs_num_constraints": (u32, [vp]),
    "lambda_snark_r1cs_num_variables": (u32, [vp]),
    "lsr_select_commit_modulus": (u64, [u64, u32]),
    "lsr_plain_modulus": (u64, [u32]),
    # prover.h (cyclic NTT + quotient polynomial of the Rust prover, additive)
    "lsr_prover_modulus": (u64, []),
    "lsr_prover_root_2_32": (u64, []),
    "lsr_prover_root_of_unity": (u64, [u64]),
    "lsr_cyclic_ntt_context_create": (vp, [u64, u32, u64, c_int]),
    "lsr_ntt_context_is_cyclic": (c_int, [vp]),
    "lsr_cyclic_ntt_forward_batch": (c_int, [vp, vp, c_size]),
    "lsr_cyclic_ntt_inverse_batch": (c_int, [vp, vp, c_size]),
    "lsr_bit_reverse_device": (c_int, [vp, vp, c_int, c_size, vp]),
    "lsr_quotient_plan_create": (vp, [u32, c_int]),
    "lsr_quotient_plan_free": (None, [vp]),
    "lsr_quotient_plan_size": (u32, [vp]),
    "lsr_quotient_batch": (c_int, [vp, vp, vp, vp, c_size, vp, vp]),
    "lsr_quotient_batch_device": (c_int, [vp, vp, vp, vp, c_size, vp, vp, vp]),
    "lsr_r1cs_prover_create": (vp, [ctypes.POINTER(SparseMatrix), ctypes.POINTER(SparseMatrix), ctypes.POINTER(SparseMatrix), c_int]),
    "lsr_r1cs_prover_free": (None, [vp]),
    "lsr_r1cs_prover_num_constraints": (u32, [vp]),
    "lsr_r1cs_prover_num_variables": (u32, [vp]),
    "lsr_r1cs_constraint_evals_batch": (c_int, [vp, vp, c_size, vp, vp, vp]),
    "lsr_r1cs_quotient_batch": (c_int, [vp, vp, c_size, vp, vp]),
}


def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own libamdhip64.so.7.  Two HIP runtimes in one process fight over the device (the
    one that initialises second sees no GPU), so when PyTorch is installed its copy is mapped first and
    liblambda_snark_core.so (NEEDED libamdhip64.so.7) binds to that same runtime.  Without PyTorch the system
    runtime under /opt/rocm is used, as a Rust or C++ caller would."""
    import importlib.util
    if os.environ.get("LAMBDA_SNARK_SYSTEM_HIP"):      # a process that never imports PyTorch: bind to the system runtime like a C caller
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        candidate = os.path.join(libdir, name)
        if os.path.exists(candidate):
            try:
                ctypes.CDLL(candidate, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                return


def load_library(path=None):
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    _share_hip_runtime_with_torch()
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the library does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = load_library()
    return _lib


def use_library(path):
    """Development tools only: bind the package to an experiment build instead of the product library."""
    global _lib
    _lib = load_library(path)
    return _lib


def last_error():
    return lib().lsr_last_error().decode()
