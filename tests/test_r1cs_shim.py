"""The SEAL/NTL-free R1CS shim (host only, so it runs without a GPU): behaviour of
cpp-core/src/ffi.cpp + r1cs.cpp, exercised the way rust-api/lambda-snark/tests/test_vectors.rs:46-63 drives it."""
import ctypes
import json
import os

import numpy as np
import pytest

OK, NULL_PTR, INVALID, ALLOC, CRYPTO = 0, 1, 2, 3, 4
Q = 17592186044417


def matrix(pkg, entries, rows, cols):
    arr = (pkg._abi.SparseEntry * max(1, len(entries)))(*[pkg._abi.SparseEntry(r, c, v % 2**64) for r, c, v in entries])
    return pkg._abi.SparseMatrix(arr, len(entries), rows, cols), arr


def system(pkg, lib, a, b, c, rows, cols, q=Q):
    ma, ka = matrix(pkg, a, rows, cols); mb, kb = matrix(pkg, b, rows, cols); mc, kc = matrix(pkg, c, rows, cols)
    h = ctypes.c_void_p()
    rc = lib.lambda_snark_r1cs_create(ctypes.byref(ma), ctypes.byref(mb), ctypes.byref(mc), q, ctypes.byref(h))
    return rc, h


def validate(pkg, lib, h, z):
    vals = np.array([v % 2**64 for v in z], dtype=np.uint64)
    w = pkg._abi.R1CSWitness(vals.ctypes.data_as(pkg._abi.u64p), len(z))
    ok = ctypes.c_bool(False)
    rc = lib.lambda_snark_r1cs_validate_witness(h, ctypes.byref(w), ctypes.byref(ok))
    return rc, ok.value


def test_multiplication_gate(pkg, lib):
    """TV-1 shape: z = [1, 7, 13, 91], a * b = c."""
    rc, h = system(pkg, lib, [(0, 1, 1)], [(0, 2, 1)], [(0, 3, 1)], 1, 4)
    assert rc == OK and lib.lambda_snark_r1cs_num_constraints(h) == 1 and lib.lambda_snark_r1cs_num_variables(h) == 4
    assert validate(pkg, lib, h, [1, 7, 13, 91]) == (OK, True)
    assert validate(pkg, lib, h, [1, 7, 13, 92]) == (OK, False)
    assert validate(pkg, lib, h, [2, 7, 13, 91])[0] == INVALID          # r1cs.cpp:103-105: z[0] must be 1
    assert validate(pkg, lib, h, [1, 7, 13])[0] == INVALID              # r1cs.cpp:96-101: length
    ok = ctypes.c_bool()
    assert lib.lambda_snark_r1cs_validate_witness(None, None, ctypes.byref(ok)) == NULL_PTR
    lib.lambda_snark_r1cs_free(h)
    lib.lambda_snark_r1cs_free(None)
    assert lib.lambda_snark_r1cs_num_constraints(None) == 0 and lib.lambda_snark_r1cs_num_variables(None) == 0


def test_plaquette_with_negative_entries(pkg, lib, golden_dir):
    """TV-2: B holds -1 entries that Rust passes as `-1i64 as u64` (test_vectors.rs:63); the reference reads them
    through static_cast<long> (r1cs.cpp:122-124)."""
    cons = json.load(open(os.path.join(golden_dir, "tv2_constraints.json")))
    c0 = cons["constraints"][0]
    ent = lambda k: [(e["row"], e["col"], e["value"]) for e in c0[k]]
    rc, h = system(pkg, lib, ent("A"), ent("B"), ent("C"), cons["m"], cons["n"], cons["modular_arithmetic"]["q"])
    assert rc == OK
    assert validate(pkg, lib, h, cons["verification"]["witness"]) == (OK, True)
    assert validate(pkg, lib, h, [1, 314, 628, 471, 472]) == (OK, False)
    assert validate(pkg, lib, h, [1, 314 - Q, 628, 471, 471]) == (OK, True)      # a negative witness word is the same residue
    lib.lambda_snark_r1cs_free(h)


def test_create_errors(pkg, lib):
    ma, _ka = matrix(pkg, [(0, 0, 1)], 1, 4)
    mb, _kb = matrix(pkg, [(0, 0, 1)], 2, 4)
    mc, _kc = matrix(pkg, [(0, 0, 1)], 1, 5)
    h = ctypes.c_void_p()
    assert lib.lambda_snark_r1cs_create(None, ctypes.byref(ma), ctypes.byref(ma), Q, ctypes.byref(h)) == NULL_PTR
    assert lib.lambda_snark_r1cs_create(ctypes.byref(ma), ctypes.byref(ma), ctypes.byref(ma), Q, None) == NULL_PTR
    assert lib.lambda_snark_r1cs_create(ctypes.byref(ma), ctypes.byref(mb), ctypes.byref(ma), Q, ctypes.byref(h)) == INVALID   # r1cs.cpp:21-23
    assert lib.lambda_snark_r1cs_create(ctypes.byref(ma), ctypes.byref(ma), ctypes.byref(mc), Q, ctypes.byref(h)) == INVALID   # r1cs.cpp:24-26
    rc, h2 = system(pkg, lib, [(0, 9, 1)], [(0, 0, 1)], [(0, 0, 1)], 1, 4)      # column outside the witness
    assert rc == OK and validate(pkg, lib, h2, [1, 2, 3, 4])[0] == CRYPTO        # r1cs.cpp:159-161 -> ffi.cpp catch (...)
    lib.lambda_snark_r1cs_free(h2)
