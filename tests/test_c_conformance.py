"""The headers are valid C11 and C++17, and a plain C program (tests/c/abi_conformance.c — the reference's gtest
assertions restated without gtest) links against liblambda_snark_core.so and passes."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "abi_conformance.c")
LIBDIR = os.path.join(ROOT, "lambda-snark-r_amd", "lib")


def build(tmp_path, compiler="gcc", std="-std=c11"):
    exe = str(tmp_path / "abi_conformance")
    cmd = [compiler, std, "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), SRC, "-L" + LIBDIR, "-llambda_snark_core",
           "-Wl,-rpath," + LIBDIR, "-lm", "-o", exe]
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    return exe


def test_headers_compile_as_c_and_cpp(tmp_path, pkg):
    for compiler, std in [("gcc", "-std=c11"), ("g++", "-std=c++17")]:
        probe = tmp_path / ("probe.c" if compiler == "gcc" else "probe.cpp")
        probe.write_text('#include "lambda_snark/batch.h"\n#include "lambda_snark/utils.h"\n#include "lambda_snark/r1cs.h"\n#include "lambda_snark/prover.h"\nint main(void) { return sizeof(PublicParams) == 32 ? 0 : 1; }\n')
        subprocess.run([compiler, std, "-Wall", "-Werror", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), str(probe)], check=True)


def test_c_runner_without_gpu_contract(tmp_path, pkg):
    exe = build(tmp_path)
    out = subprocess.run([exe, "--no-gpu"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert out.returncode == 0, out.stdout


@pytest.mark.gpu
def test_c_runner_full(tmp_path, pkg):
    exe = build(tmp_path)
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert out.returncode == 0, out.stdout
    assert "0 failures" in out.stdout
