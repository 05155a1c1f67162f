"""CPU suite, part 2: the C-ABI library loads and exports every symbol include/lambda_snark/*.h declares;
host-only logic (parameter selection, root search, CDT table) agrees with the oracle.  No compute calls —
there is no GPU here and the library has no CPU fallback."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INCLUDE = os.path.join(ROOT, "include", "lambda_snark")


def declared_symbols():
    names = set()
    for fn in sorted(os.listdir(INCLUDE)):
        text = open(os.path.join(INCLUDE, fn)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        for m in re.finditer(r"\b([a-z_][a-z0-9_]*)\s*\(", text):
            name = m.group(1)
            if name.startswith(("ntt_", "lwe_", "lsr_", "sample_gaussian", "lambda_snark_r1cs_")):
                names.add(name)
    return names


def test_headers_declare_the_reference_surface():
    syms = declared_symbols()
    # the 12 hot-path symbols of SURVEY.md §8(b) + sample_gaussian
    for s in ["ntt_context_create", "ntt_context_free", "ntt_forward", "ntt_inverse", "ntt_mul_pointwise", "lwe_context_create",
              "lwe_context_free", "lwe_commit", "lwe_commitment_free", "lwe_commitment_clone", "lwe_verify_opening", "lwe_linear_combine",
              "sample_gaussian"]:
        assert s in syms


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg._abi.load_library()
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    assert declared_symbols() == set(pkg._abi.SIGNATURES), "ctypes table and headers disagree"
    out = subprocess.run(["nm", "-D", "--defined-only", pkg._abi.LIB_PATH], stdout=subprocess.PIPE, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    assert declared_symbols() <= exported


def test_struct_layouts_match_types_h(pkg):
    """types.h:36-67 — bindgen consumes these layouts."""
    assert ctypes.sizeof(pkg.PublicParams) == 32
    assert [f[0] for f in pkg.PublicParams._fields_] == ["profile", "security_level", "modulus", "ring_degree", "module_rank", "sigma"]
    assert pkg.PublicParams.modulus.offset == 8 and pkg.PublicParams.ring_degree.offset == 16 and pkg.PublicParams.sigma.offset == 24
    assert ctypes.sizeof(pkg._abi.LweCommitment) == 16 and ctypes.sizeof(pkg._abi.LweOpening) == 16


def test_host_number_theory_matches_oracle(lib, oracle):
    for q, n in [(12289, 256), (12289, 2048), (17592169062401, 4096), (17592182243329, 65536), (1152921504606584833, 131072), (17592169062401, 2)]:
        assert lib.lsr_minimal_primitive_root(q, n) == oracle.L.oracle_minimal_primitive_root(2 * n, q) != 0
    for q, n in [(17592169062401, 65536), (17592186044417, 4096), (12289, 4096), (12289, 0), (12289, 3), (2**64 - 2**32 + 1, 256), (12289, 1), (0, 8)]:
        assert lib.lsr_minimal_primitive_root(q, n) == 0
    for req, n in [(12289, 4096), (17592186044417, 4096), (17592186044423, 4096), (17592169062401, 256), (12289, 65536), (17592182243329, 65536),
                   (1152921504606584833, 8192), (5, 131072), (7, 1), (7, 12)]:
        assert lib.lsr_select_commit_modulus(req, n) == oracle.L.oracle_lwe_select_modulus(req, n)
    for n in [2, 256, 1024, 4096, 32768, 65536, 131072]:
        assert lib.lsr_plain_modulus(n) == oracle.L.oracle_largest_prime_1mod(2 * n, 20) != 0
    assert lib.lsr_plain_modulus(4096) == 1032193


def test_cdt_table_matches_oracle(lib, oracle):
    for sigma in [3.19, 3.2, 0.5, 1.0, 8.0, 64.0]:
        buf = np.zeros(4096, dtype=np.uint64)
        cnt = lib.lsr_gaussian_cdf(sigma, buf.ctypes.data, buf.size)
        assert np.array_equal(buf[:cnt], oracle.gaussian_cdf(sigma))
    buf = np.zeros(16, dtype=np.uint64)
    assert lib.lsr_gaussian_cdf(0.0, buf.ctypes.data, 16) == 0
    assert lib.lsr_gaussian_cdf(float("nan"), buf.ctypes.data, 16) == 0
    assert lib.lsr_gaussian_cdf(3.19, buf.ctypes.data, 16) == 0     # table does not fit


def test_null_and_argument_contract_without_gpu(lib, pkg):
    """Error paths that never reach the device (ntt.cpp:81,96,113; commitment.cpp:103,144,207,240)."""
    buf = np.zeros(8, dtype=np.uint64)
    assert lib.ntt_forward(None, buf.ctypes.data, 8) == -1
    assert lib.ntt_inverse(None, buf.ctypes.data, 8) == -1
    lib.ntt_mul_pointwise(None, buf.ctypes.data, buf.ctypes.data, buf.ctypes.data, 8)   # silent no-op
    lib.ntt_context_free(None)
    assert not lib.lwe_context_create(None)
    lib.lwe_context_free(None)
    assert not lib.lwe_commit(None, None, 0, 0)
    lib.lwe_commitment_free(None)
    assert not lib.lwe_commitment_clone(None)
    assert lib.lwe_verify_opening(None, None, None, 0, None) == -1
    assert not lib.lwe_linear_combine(None, None, None, 0)
    assert lib.sample_gaussian(None, 16, 3.2) == -1
    assert lib.sample_gaussian(buf.ctypes.data, 0, 3.2) == -1
    assert lib.sample_gaussian(buf.ctypes.data, 8, 0.0) == -1
    assert lib.sample_gaussian(buf.ctypes.data, 8, float("inf")) == -1
    # invalid (q, n) are rejected before any device work
    assert not lib.ntt_context_create(17592169062401, 65536)
    assert not lib.ntt_context_create(12289, 3)
    assert not lib.ntt_context_create(12289, 0)


def test_fails_loudly_without_gpu(lib):
    if lib.lsr_device_count() > 0:
        pytest.skip("a GPU is visible")
    assert not lib.ntt_context_create(12289, 256)
    assert b"no HIP device" in lib.lsr_last_error()
    buf = np.zeros(8, dtype=np.uint64)
    assert lib.sample_gaussian(buf.ctypes.data, 8, 3.2) == -1      # no silent CPU fallback


def test_missing_library_fails_loudly(pkg, tmp_path):
    with pytest.raises(ImportError):
        pkg._abi.load_library(str(tmp_path / "nope.so"))


def test_fs_challenge_matches_sha3_transcript(lib, pkg):
    """lsr_fs_challenge is host-only: challenge.rs:102-134 against hashlib's SHA3-256 (FIPS 202)."""
    import prover_replay
    rng = np.random.default_rng(7)
    for n_words, inputs in [(1, []), (5, [1, 471]), (17, [2**64 - 1]), (12293, [1, 91]), (136 // 8 * 3 + 1, list(range(40)))]:
        words = rng.integers(0, 2**64, size=n_words, dtype=np.uint64)
        com = pkg._abi.LweCommitment(words.ctypes.data_as(pkg._abi.u64p), n_words)
        inp = np.array(inputs, dtype=np.uint64)
        alpha = ctypes.c_uint64(0)
        digest = (ctypes.c_uint8 * 32)()
        for q in [17592186044417, 17592169062401, 12289]:
            assert lib.lsr_fs_challenge(inp.ctypes.data if inputs else None, len(inputs), ctypes.byref(com), q, ctypes.byref(alpha), digest) == 0
            want_alpha, want_hash = prover_replay.challenge_derive(inputs, words, q)
            assert alpha.value == want_alpha and bytes(digest) == want_hash
    # the batched form: rows of one array, a pool of host threads
    count, row_words, n_in = 37, 300, 3
    rows = rng.integers(0, 2**64, size=(count, row_words), dtype=np.uint64)
    ins = rng.integers(0, 2**64, size=(count, n_in), dtype=np.uint64)
    alphas = np.zeros(count, dtype=np.uint64); hashes = np.zeros((count, 32), dtype=np.uint8)
    for threads in (0, 1, 5, 64):
        alphas[:] = 0; hashes[:] = 0
        assert lib.lsr_fs_challenge_batch_flat(ins.ctypes.data, n_in, rows.ctypes.data, row_words, count, 17592186044417, alphas.ctypes.data,
                                               hashes.ctypes.data, threads) == 0
        for i in range(count):
            want_alpha, want_hash = prover_replay.challenge_derive([int(x) for x in ins[i]], rows[i], 17592186044417)
            assert alphas[i] == want_alpha and bytes(hashes[i]) == want_hash
    assert lib.lsr_fs_challenge_batch_flat(None, 0, rows.ctypes.data, row_words, count, 12289, alphas.ctypes.data, None, 2) == 0
    assert alphas[0] == prover_replay.challenge_derive([], rows[0], 12289)[0]
    assert lib.lsr_fs_challenge_batch_flat(None, 2, rows.ctypes.data, row_words, count, 12289, alphas.ctypes.data, None, 2) == -1
    assert lib.lsr_fs_challenge_batch_flat(None, 0, None, row_words, count, 12289, alphas.ctypes.data, None, 2) == -1
    assert lib.lsr_fs_challenge_batch_flat(None, 0, rows.ctypes.data, row_words, 0, 12289, alphas.ctypes.data, None, 2) == 0
    assert lib.lsr_fs_challenge(None, 0, None, 12289, ctypes.byref(alpha), None) == -1
    assert lib.lsr_fs_challenge(None, 0, ctypes.byref(com), 0, ctypes.byref(alpha), None) == -1
