import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package(); lib = pkg._abi.lib()
nb = 2048
msgs = (np.arange(nb * 8, dtype=np.uint64).reshape(nb, 8) * 7919) % 1000003
seeds = np.arange(1, nb + 1, dtype=np.uint64)
def wall(fn, reps=5):
    fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return round(float(np.median(ts)) * 1e3, 3)
def measure(tag):
    ctx = pkg.LweContext(pkg.Params(q=17592186044417, n=4096, k=2, sigma=3.19), key_seed=5, device=0)
    words = lib.lsr_lwe_commitment_words(ctx.handle)
    rows = np.zeros((nb, words), dtype=np.uint64)
    a = wall(lambda: lib.lsr_lwe_commit_batch_flat(ctx.handle, msgs.ctypes.data, 8, nb, seeds.ctypes.data, rows.ctypes.data))
    pin = pkg.PinnedArray(rows.shape)
    b = wall(lambda: lib.lsr_lwe_commit_batch_flat(ctx.handle, msgs.ctypes.data, 8, nb, seeds.ctypes.data, pin.ptr))
    print(tag, "pageable", a, "pinned", b, flush=True)
    pin.close(); ctx.close()
s = torch.cuda.current_stream().cuda_stream
measure("fresh process")
x = torch.empty((4096, 65536), dtype=torch.int64, device="cuda"); del x; torch.cuda.empty_cache()
measure("after a 2 GiB torch tensor")
n1 = pkg.NttContext(17592182243329, 65536, device=0)
p = torch.zeros((64, 65536), dtype=torch.int64, device="cuda")
n1.forward_device(p.data_ptr(), 64, s); torch.cuda.synchronize()
measure("after an NTT context")
l1 = pkg.LweContext(pkg.Params(q=17592182243329, n=65536, k=4, sigma=3.19), key_seed=7, device=0)
r = torch.zeros((128, 4, 65536), dtype=torch.int64, device="cuda"); u = torch.empty_like(r)
lib.lsr_mlwe_matvec_batch_device(l1.handle, r.data_ptr(), r.data_ptr(), u.data_ptr(), 128, None, s); torch.cuda.synchronize()
measure("after a mixed-launch matvec (side stream)")
sd = np.arange(1, 129, dtype=np.uint64)
lib.lsr_mlwe_matvec_batch_device(l1.handle, r.data_ptr(), None, u.data_ptr(), 128, sd.ctypes.data, s); torch.cuda.synchronize()
measure("after a sampled matvec")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(40)]
for e in ev: e.record()
torch.cuda.synchronize()
measure("after 40 timing events")
st = [torch.cuda.Stream() for _ in range(3)]
measure("after 3 more torch streams")
