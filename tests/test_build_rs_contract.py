"""CPU suite: the contract of integration/lambda-snark-sys/build.rs, as far as it can be checked
without cargo: every header it names exists and is plain C; every function the reference's bindgen allow-list lets through
(rust-api/lambda-snark-sys/build.rs:196-198: lwe_.*, ntt_.*, lambda_snark_r1cs_.*) is declared in one of those headers AND
exported (dynamic symbol table) by liblambda_snark_core.so; the Rust side's own extern declarations
(lambda-snark-core/src/r1cs.rs:121-141, lambda-snark-sys/src/lib.rs:35-42) are among them."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD_RS = os.path.join(ROOT, "integration", "lambda-snark-sys", "build.rs")
INCLUDE = os.path.join(ROOT, "include")
ALLOW = [r"lwe_.*", r"ntt_.*", r"lambda_snark_r1cs_.*"]
# what the reference's Rust code calls through the FFI (SURVEY.md §8(b) "Callers")
RUST_CALLS = ["lwe_context_create", "lwe_context_free", "lwe_commit", "lwe_commitment_free", "lwe_commitment_clone", "lwe_verify_opening",
              "lwe_linear_combine", "ntt_context_create", "ntt_context_free", "ntt_forward", "ntt_inverse", "ntt_mul_pointwise",
              "lambda_snark_r1cs_create", "lambda_snark_r1cs_validate_witness", "lambda_snark_r1cs_free",
              "lambda_snark_r1cs_num_constraints", "lambda_snark_r1cs_num_variables"]


def headers_named_in_build_rs():
    src = open(BUILD_RS).read()
    return re.findall(r'header\(header\("([a-z0-9_]+\.h)"\)\)', src)


def declared_functions(header):
    text = open(os.path.join(INCLUDE, "lambda_snark", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b([a-z][a-z0-9_]*)\s*\(", text)) - {"defined", "sizeof"}


def exported_symbols(pkg):
    import __graft_entry__ as entry
    out = subprocess.run(["nm", "-D", "--defined-only", entry.LIB], stdout=subprocess.PIPE, text=True, check=True).stdout
    return {line.split()[-1] for line in out.splitlines() if " T " in line}


def test_build_rs_names_existing_plain_c_headers():
    names = headers_named_in_build_rs()
    assert names[:4] == ["types.h", "commitment.h", "ntt.h", "r1cs.h"]          # the reference's four, same order (build.rs:186-189)
    assert set(names[4:]) == {"batch.h", "prover.h"}
    for name in names:
        path = os.path.join(INCLUDE, "lambda_snark", name)
        assert os.path.exists(path), name
        # bindgen parses them as C: they must compile as C11 on their own (no NTL / SEAL / C++ headers: reference r1cs.h:26)
        subprocess.run(["gcc", "-std=c11", "-fsyntax-only", "-I" + INCLUDE, "-x", "c", path], check=True)
    code = "\n".join(line for line in open(BUILD_RS).read().splitlines() if not line.lstrip().startswith("//"))
    assert not re.search(r"rustc-link-lib=[^\"]*(seal|zstd|ntl|gmp|=z\b)", code)      # reference build.rs:119-125
    assert "cmake::" not in code and "vcpkg" not in code.lower()                      # reference build.rs:31-104
    for kept in ("rustc-link-lib=dylib=lambda_snark_core", "rustc-link-lib=dylib=amdhip64", "rustc-link-lib=stdc++"):
        assert kept in code


def test_allow_listed_functions_are_declared_and_exported(pkg):
    declared = set()
    for h in ("types.h", "commitment.h", "ntt.h", "r1cs.h"):
        declared |= declared_functions(h)
    exported = exported_symbols(pkg)
    bound = {f for f in declared if any(re.fullmatch(p, f) for p in ALLOW)}
    assert set(RUST_CALLS) <= bound, sorted(set(RUST_CALLS) - bound)
    missing = sorted(f for f in bound if f not in exported)
    assert not missing, missing
    # the additive headers: everything lsr_* they declare is exported too
    extra = {f for h in ("batch.h", "prover.h") for f in declared_functions(h) if re.fullmatch(r"lsr_.*|.*_batch|sample_gaussian", f)}
    assert not sorted(f for f in extra if f not in exported)
