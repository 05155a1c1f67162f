#!/usr/bin/env python3
"""Parity checks through the C-ABI in a process that never imports PyTorch, so that liblambda_snark_core.so binds to the
SYSTEM HIP runtime (/opt/rocm) exactly as a Rust or C++ caller's would — the configuration the rest of the GPU suite (which
shares PyTorch's bundled runtime) does not exercise.  Device memory comes from hipMalloc via ctypes.
Run by tests/test_system_runtime_gpu.py; exits non-zero on the first mismatch."""
import ctypes
import os
import sys

os.environ["LAMBDA_SNARK_SYSTEM_HIP"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import __graft_entry__ as entry  # noqa: E402
import oracle_binding  # noqa: E402

pkg = entry.load_package()
lib = pkg._abi.lib()
orc = oracle_binding.load()
assert "torch" not in sys.modules, "this runner must not import PyTorch"
hip = ctypes.CDLL("libamdhip64.so")          # already mapped by the library: the same runtime instance
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
hip.hipFree.argtypes = [ctypes.c_void_p]
H2D, D2H = 1, 2
checks = 0


def ok(cond, what):
    global checks
    if not cond:
        print("MISMATCH:", what)
        sys.exit(1)
    checks += 1


def to_device(a):
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), a.nbytes) == 0
    assert hip.hipMemcpy(p, a.ctypes.data, a.nbytes, H2D) == 0
    return p


def to_host(p, shape):
    out = np.empty(shape, dtype=np.uint64)
    assert hip.hipDeviceSynchronize() == 0
    assert hip.hipMemcpy(out.ctypes.data, p, out.nbytes, D2H) == 0
    return out


with open("/proc/self/maps") as f:
    runtimes = sorted({line.split()[-1] for line in f if "libamdhip64" in line})
print("HIP runtime mapped:", runtimes)
ok(len(runtimes) == 1 and "torch" not in runtimes[0], "exactly one HIP runtime, not PyTorch's")

# transforms, host arrays
for q, n, batch in [(12289, 256, 9), (17592169062401, 4096, 5), (17592182243329, 65536, 3), (17592182243329, 8192, 4)]:
    ctx = pkg.NttContext(q, n)
    a = orc.splitmix(0x5151 + n, q, batch * n).reshape(batch, n)
    f = ctx.forward_batch(a.copy())
    ok(np.array_equal(f, orc.ntt_forward(q, n, a)), f"forward {q} {n}")
    ok(np.array_equal(ctx.inverse_batch(f.copy()), a), f"inverse {q} {n}")
    b = orc.splitmix(77, q, n)
    ok(np.array_equal(ctx.mul_pointwise(a[0], b), orc.mul_pointwise(q, n, a[0], b)), f"pointwise {q} {n}")
    # device-resident entry points on memory from hipMalloc
    d = to_device(a)
    ok(lib.lsr_ntt_forward_batch_device(ctx.handle, d, batch, None) == 0, "device forward rc")
    ok(np.array_equal(to_host(d, a.shape), f), f"device forward {q} {n}")
    ok(lib.lsr_ntt_inverse_batch_device(ctx.handle, d, batch, None) == 0, "device inverse rc")
    ok(np.array_equal(to_host(d, a.shape), a), f"device inverse {q} {n}")
    hip.hipFree(d)
    ctx.close()

# sampler
ok(np.array_equal(pkg.sample_gaussian(1000, 3.19, seed=0x1234, domain=5, index=1), orc.sample_gaussian_seeded(1000, 3.19, 0x1234, 5, 1)), "seeded sampler")

# commitments at the reference's parameters
q, n, k, key = 17592186044417, 4096, 2, 0xABCDEF
lctx = pkg.LweContext(pkg.Params(q=q, n=n, k=k, sigma=3.19), key_seed=key)
msgs = (np.arange(7 * 6, dtype=np.uint64).reshape(7, 6) * 7919) % 1000
seeds = np.arange(1, 8, dtype=np.uint64)
rows = pkg.Commitment.batch_words(lctx, msgs, seeds)
for i in (0, 3, 6):
    ok(np.array_equal(rows[i], orc.lwe_commit(q, n, k, 3.19, key, msgs[i], int(seeds[i]))), f"flat commit {i}")
c0 = pkg.Commitment(lctx, msgs[0], seed=int(seeds[0]))
ok(np.array_equal(c0.as_words(), rows[0]), "lwe_commit == flat row")
ok(pkg.verify_opening_with_context(lctx, c0, msgs[0]), "verify")
wrong = msgs[0].copy(); wrong[2] ^= 1
ok(not pkg.verify_opening_with_context(lctx, c0, wrong), "verify rejects")
ok(pkg.verify_openings_words(lctx, rows, msgs) == [1] * 7, "flat verify")
c1 = pkg.Commitment(lctx, msgs[1], seed=int(seeds[1]))
comb = pkg.Commitment.linear_combine(lctx, [c0, c1], [2, 3])
rc, want = orc.lwe_linear_combine(q, n, k, 3.19, key, [c0.as_words(), c1.as_words()], [2, 3])
ok(rc == 0 and np.array_equal(comb.as_words(), want), "linear combine")
twin = lctx.replicate(0)
pinned = pkg.PinnedArray(rows.shape)
pkg.sharded_commit_words([lctx, twin], msgs, seeds, out=pinned.array)
ok(np.array_equal(pinned.array, rows), "sharded commit over two replicas")
pinned.close(); twin.close(); lctx.close()

# the matrix–vector workload at n = 2^16, rank 4 (fused pipeline), device memory from hipMalloc
q, n, k, batch = 17592182243329, 65536, 4, 3
lctx = pkg.LweContext(pkg.Params(q=q, n=n, k=k, sigma=3.19), key_seed=0xC0DE)
a_hat = lctx.public_matrix()
r = np.stack([orc.splitmix(0xC0FFEE + j, q, k * n).reshape(k, n) for j in range(batch)])
seeds = np.array([11, 22, 33], dtype=np.uint64)
e1 = np.stack([np.stack([orc.sample_gaussian_seeded(n, 3.19, int(seeds[j]), 5, i) for i in range(k)]) for j in range(batch)])
e1 = np.where(e1 < 0, e1 + q, e1).astype(np.uint64)
d_r, d_u = to_device(r), to_device(np.zeros_like(r))
ok(lib.lsr_mlwe_matvec_batch_device(lctx.handle, d_r, None, d_u, batch, seeds.ctypes.data, None) == 0, "matvec rc")
u = to_host(d_u, r.shape)
for j in range(batch):
    ok(np.array_equal(u[j], orc.mlwe_matvec(q, n, k, a_hat, r[j], e1[j])), f"matvec vector {j}")
hip.hipFree(d_r); hip.hipFree(d_u)
lctx.close()
print(f"system-runtime runner: {checks} checks passed")
