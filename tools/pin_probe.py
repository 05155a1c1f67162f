import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package(); lib = pkg._abi.lib()
ctx = pkg.LweContext(pkg.Params(q=17592186044417, n=4096, k=2, sigma=3.19), key_seed=5, device=0)
nb = 2048
msgs = (np.arange(nb * 8, dtype=np.uint64).reshape(nb, 8) * 7919) % 1000003
seeds = np.arange(1, nb + 1, dtype=np.uint64)
words = lib.lsr_lwe_commitment_words(ctx.handle)
def wall(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, np.median(ts) * 1e3
rows = np.zeros((nb, words), dtype=np.uint64)
print("pageable", wall(lambda: lib.lsr_lwe_commit_batch_flat(ctx.handle, msgs.ctypes.data, 8, nb, seeds.ctypes.data, rows.ctypes.data)))
pin = pkg.PinnedArray(rows.shape)
print("lsr pinned", wall(lambda: lib.lsr_lwe_commit_batch_flat(ctx.handle, msgs.ctypes.data, 8, nb, seeds.ctypes.data, pin.ptr)))
tp = torch.empty((nb, words), dtype=torch.int64, pin_memory=True)
print("torch pinned", wall(lambda: lib.lsr_lwe_commit_batch_flat(ctx.handle, msgs.ctypes.data, 8, nb, seeds.ctypes.data, tp.data_ptr())))
d = torch.empty((nb, words), dtype=torch.int64, device="cuda")
print("device", wall(lambda: lib.lsr_lwe_commit_batch_flat_device(ctx.handle, msgs.ctypes.data, 8, nb, seeds.ctypes.data, d.data_ptr())))
# plain copies
def cp(dst):
    torch.cuda.synchronize(); t0 = time.perf_counter(); dst.copy_(d, non_blocking=True); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
print("torch D2H to torch-pinned", [round(cp(tp), 2) for _ in range(3)])
pa = torch.from_numpy(pin.array.view(np.int64))
print("torch D2H to lsr-pinned (as pageable view)", [round(cp(pa), 2) for _ in range(3)])
# bench.py's order: commit pageable, verify pageable, allocate pinned, commit pinned, verify pinned — three times
ver = np.zeros(nb, dtype=np.int32)
for rnd in range(3):
    a = wall(lambda: lib.lsr_lwe_commit_batch_flat(ctx.handle, msgs.ctypes.data, 8, nb, seeds.ctypes.data, rows.ctypes.data), reps=3)
    b = wall(lambda: lib.lsr_lwe_verify_opening_batch_flat(ctx.handle, rows.ctypes.data, msgs.ctypes.data, 8, nb, ver.ctypes.data), reps=3)
    p2 = pkg.PinnedArray(rows.shape)
    c = wall(lambda: lib.lsr_lwe_commit_batch_flat(ctx.handle, msgs.ctypes.data, 8, nb, seeds.ctypes.data, p2.ptr), reps=3)
    d2 = wall(lambda: lib.lsr_lwe_verify_opening_batch_flat(ctx.handle, p2.ptr, msgs.ctypes.data, 8, nb, ver.ctypes.data), reps=3)
    print("round", rnd, "commit pageable", a, "verify", b, "commit pinned", c, "verify pinned", d2)
    p2.close()
