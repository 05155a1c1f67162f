"""The library bound to the SYSTEM HIP runtime (/opt/rocm), as a Rust / C++ caller gets it: the rest of the GPU suite runs in
processes that share PyTorch's bundled runtime (lambda-snark-r_amd/_abi.py), so this test starts a process that never imports
PyTorch and repeats the core parity checks there (tests/system_runtime_runner.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parity_on_the_system_hip_runtime(pkg):
    env = dict(os.environ, LAMBDA_SNARK_SYSTEM_HIP="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "system_runtime_runner.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-4000:]
    assert "checks passed" in out.stdout and "MISMATCH" not in out.stdout
    line = [l for l in out.stdout.splitlines() if l.startswith("HIP runtime mapped:")][0]
    assert "torch" not in line and "/opt/rocm" in line, line
