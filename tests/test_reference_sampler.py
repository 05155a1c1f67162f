"""The one piece of the reference that compiles here — cpp-core/src/utils.cpp, built where it lies into
oracle/_ref/libref_utils.so by `make -C oracle ref` — checked against the oracle's restatement of the same sampler.
The reference draws from std::random_device, so the comparison is distributional: its histogram against the
probabilities of the oracle's CDT table (the table itself is in an anonymous namespace and cannot be read out)."""
import ctypes
import math
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libref_utils.so")
if not os.path.exists(REF) and os.path.exists("/root/reference/cpp-core/src/utils.cpp"):     # CPU container: build it on demand
    import subprocess
    subprocess.run(["make", "ref"], cwd=os.path.join(ROOT, "oracle"), check=False, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)

pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built (needs /root/reference at build time)")


@pytest.fixture(scope="module")
def ref():
    lib = ctypes.CDLL(REF)
    lib.sample_gaussian.restype = ctypes.c_int
    lib.sample_gaussian.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double]
    return lib


def table_probabilities(cdf):
    """P(sample = +-k) from the 64-bit CDT: first k with cdf[k] >= u, u uniform on [0, 2^64); sign bit fair for k != 0."""
    edges = [int(x) for x in cdf]
    mass = [edges[0] + 1] + [edges[k] - edges[k - 1] for k in range(1, len(edges))]
    total = float(2**64)
    probs = {0: mass[0] / total}
    for k in range(1, len(mass)):
        probs[k] = probs[-k] = mass[k] / total / 2
    return probs


def chi_square(samples, probs, min_expected=20.0):
    n = len(samples)
    values, counts = np.unique(samples, return_counts=True)
    seen = dict(zip((int(v) for v in values), (int(c) for c in counts)))
    assert set(seen) <= set(probs), "a sample outside the table's support"
    stat, dof, tail_obs, tail_exp = 0.0, 0, 0, 0.0
    for v, p in probs.items():
        exp = p * n
        if exp < min_expected:
            tail_obs += seen.get(v, 0); tail_exp += exp
            continue
        stat += (seen.get(v, 0) - exp) ** 2 / exp
        dof += 1
    if tail_exp >= 1.0:
        stat += (tail_obs - tail_exp) ** 2 / tail_exp
        dof += 1
    return stat, dof - 1


@pytest.mark.parametrize("sigma", [3.19, 3.2, 1.0, 8.0])
def test_reference_sampler_follows_the_restated_table(ref, oracle, sigma):
    n = 400_000
    out = np.zeros(n, dtype=np.uint64)
    assert ref.sample_gaussian(out.ctypes.data, n, sigma) == 0
    samples = out.view(np.int64)
    probs = table_probabilities(oracle.gaussian_cdf(sigma))
    assert abs(sum(probs.values()) - 1.0) < 1e-12
    bound = max(8, math.ceil(12.0 * sigma))
    assert int(np.abs(samples).max()) <= bound == max(probs)                 # utils.cpp:31-37 tail bound
    stat, dof = chi_square(samples, probs)
    # chi-square with `dof` degrees of freedom: mean dof, sd sqrt(2 dof); 6 sd is a < 1e-8 false-alarm rate
    assert stat < dof + 6.0 * math.sqrt(2.0 * dof), (sigma, stat, dof)
    assert abs(float(samples.mean())) < 6.0 * sigma / math.sqrt(n)           # test_utils.cpp:40-50 moments
    assert abs(float(samples.std()) - sigma) < 0.02 * sigma + 0.01


def test_restated_sampler_passes_the_same_check(oracle):
    """The oracle's own sampler (seeded ChaCha20 words through the same CDT scan) under the identical statistic."""
    for sigma in [3.19, 8.0]:
        samples = oracle.sample_gaussian_seeded(400_000, sigma, 0xC0FFEE, 16, 0)
        stat, dof = chi_square(samples, table_probabilities(oracle.gaussian_cdf(sigma)))
        assert stat < dof + 6.0 * math.sqrt(2.0 * dof)


def test_argument_contract_equals_the_reference(ref, oracle):
    buf = np.zeros(8, dtype=np.uint64)
    cases = [(None, 16, 3.2), (buf.ctypes.data, 0, 3.2), (buf.ctypes.data, 8, 0.0), (buf.ctypes.data, 8, -1.0), (buf.ctypes.data, 8, float("inf")),
             (buf.ctypes.data, 8, float("nan")), (buf.ctypes.data, 8, 3.19), (buf.ctypes.data, 1, 0.5)]
    for ptr, length, sigma in cases:
        assert ref.sample_gaussian(ptr, length, sigma) == oracle.L.oracle_sample_gaussian(ptr, length, sigma), (length, sigma)


@pytest.mark.gpu
def test_gpu_sampler_against_the_reference_distribution(ref, pkg):
    """Two-sample check: the library's sample_gaussian (device ChaCha20 + CDT) against the reference binary's output."""
    n, sigma = 400_000, 3.19
    out = np.zeros(n, dtype=np.uint64)
    assert ref.sample_gaussian(out.ctypes.data, n, sigma) == 0
    theirs = out.view(np.int64)
    ours = pkg.sample_gaussian(n, sigma)
    lo, hi = int(min(theirs.min(), ours.min())), int(max(theirs.max(), ours.max()))
    a = np.bincount(theirs - lo, minlength=hi - lo + 1).astype(float)
    b = np.bincount(ours - lo, minlength=hi - lo + 1).astype(float)
    keep = (a + b) >= 40
    stat = float(((a[keep] - b[keep]) ** 2 / (a[keep] + b[keep])).sum())      # two-sample chi-square, equal sizes
    dof = int(keep.sum()) - 1
    assert stat < dof + 6.0 * math.sqrt(2.0 * dof), (stat, dof)
