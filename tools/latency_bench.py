"""Single-call latency of the legacy (host-pointer) C-ABI at the reference's parameters n=4096, k=2."""
import os, sys, time, ctypes, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package(); lib = pkg._abi.lib()
def bench(fn, reps=200):
    for _ in range(10): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e6
q, n = 17592169062401, 4096
ctx = pkg.NttContext(q, n)
a = (np.arange(n, dtype=np.uint64) * 7919) % q
print(f"ntt_forward  n=4096 host ptr : {bench(lambda: lib.ntt_forward(ctx.handle, a.ctypes.data, n)):8.1f} us")
print(f"ntt_inverse  n=4096 host ptr : {bench(lambda: lib.ntt_inverse(ctx.handle, a.ctypes.data, n)):8.1f} us")
b = a.copy(); r = a.copy()
print(f"ntt_mul_pointwise n=4096     : {bench(lambda: lib.ntt_mul_pointwise(ctx.handle, r.ctypes.data, a.ctypes.data, b.ctypes.data, n)):8.1f} us")
t0 = time.perf_counter(); lctx = pkg.LweContext(pkg.Params(q=17592186044417, n=4096, k=2, sigma=3.19)); t1 = time.perf_counter()
print(f"lwe_context_create           : {(t1-t0)*1e6:8.1f} us")
msg = np.array([1, 314, 628, 471, 471], dtype=np.uint64)
held = []
def commit():
    p = lib.lwe_commit(lctx.handle, msg.ctypes.data, msg.size, 0x5678); lib.lwe_commitment_free(p)
print(f"lwe_commit (+free) n=4096 k=2: {bench(commit):8.1f} us   (reference: ~4000-6000 us, ROADMAP.md:344-359)")
c1 = lib.lwe_commit(lctx.handle, msg.ctypes.data, msg.size, 1); c2 = lib.lwe_commit(lctx.handle, msg.ctypes.data, msg.size, 2)
print(f"lwe_verify_opening           : {bench(lambda: lib.lwe_verify_opening(lctx.handle, c1, msg.ctypes.data, msg.size, None)):8.1f} us")
arr = (ctypes.POINTER(pkg._abi.LweCommitment) * 2)(c1, c2); cf = np.array([2, 3], dtype=np.uint64)
def comb():
    p = lib.lwe_linear_combine(lctx.handle, arr, cf.ctypes.data, 2); lib.lwe_commitment_free(p)
print(f"lwe_linear_combine (2 terms) : {bench(comb):8.1f} us")
many = [lib.lwe_commit(lctx.handle, msg.ctypes.data, msg.size, 100 + i) for i in range(256)]
arr256 = (ctypes.POINTER(pkg._abi.LweCommitment) * 256)(*many); cf256 = np.arange(256, dtype=np.uint64) % 3
def comb256():
    p = lib.lwe_linear_combine(lctx.handle, arr256, cf256.ctypes.data, 256); lib.lwe_commitment_free(p)
print(f"lwe_linear_combine (256 terms): {bench(comb256, 20):8.1f} us")
buf = np.zeros(4096, dtype=np.uint64)
print(f"sample_gaussian(4096)        : {bench(lambda: lib.sample_gaussian(buf.ctypes.data, 4096, 3.19), 50):8.1f} us")
# full commitments (u and v, sampling included, words returned to the host) at the reference's parameters
batch = 2048
msgs = (np.arange(batch * 8, dtype=np.uint64).reshape(batch, 8) * 7919) % 1000003
seeds = np.arange(1, batch + 1, dtype=np.uint64)
out = (ctypes.POINTER(pkg._abi.LweCommitment) * batch)()
def cb():
    assert lib.lwe_commit_batch(lctx.handle, msgs.ctypes.data, 8, batch, seeds.ctypes.data, out) == 0
    for i in range(batch): lib.lwe_commitment_free(out[i])
cb()
t0 = time.perf_counter(); cb(); dt = time.perf_counter() - t0
print(f"lwe_commit_batch n=4096 k=2 x{batch} (host words out): {dt*1e3:.1f} ms = {batch/dt/1e3:.1f} K commits/s  (reference: ~0.2 K commits/s/core implied, BASELINE.md §2)")
flat = np.zeros((batch, lib.lsr_lwe_commitment_words(lctx.handle)), dtype=np.uint64)
def cf():
    assert lib.lsr_lwe_commit_batch_flat(lctx.handle, msgs.ctypes.data, 8, batch, seeds.ctypes.data, flat.ctypes.data) == 0
t = bench(cf, 5)
print(f"lsr_lwe_commit_batch_flat n=4096 k=2 x{batch} (one host array out): {t/1e3:.1f} ms = {batch/t*1e3:.1f} K commits/s")
res = np.zeros(batch, dtype=np.int32)
def vf():
    assert lib.lsr_lwe_verify_opening_batch_flat(lctx.handle, flat.ctypes.data, msgs.ctypes.data, 8, batch, res.ctypes.data) == 0
t = bench(vf, 5)
assert (res == 1).all()
print(f"lsr_lwe_verify_opening_batch_flat x{batch}: {t/1e3:.1f} ms = {batch/t*1e3:.1f} K openings/s")
assert lib.lwe_commit_batch(lctx.handle, msgs.ctypes.data, 8, batch, seeds.ctypes.data, out) == 0
def vb():
    assert lib.lwe_verify_opening_batch(lctx.handle, out, msgs.ctypes.data, 8, batch, res.ctypes.data) == 0
t = bench(vb, 5)
print(f"lwe_verify_opening_batch x{batch} (LweCommitment* array): {t/1e3:.1f} ms = {batch/t*1e3:.1f} K openings/s")

