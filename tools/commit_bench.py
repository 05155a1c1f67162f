"""Development microbench for config 3 (rank-4 matvec commitment) and the sampler."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
if os.environ.get("LIBVARIANT"):          # an experiment build of the library (csrc/Makefile VARIANT=...), loaded explicitly
    pkg._abi.use_library(os.path.join(os.path.dirname(pkg._abi.LIB_PATH), "liblambda_snark_core_%s.so" % os.environ["LIBVARIANT"]))
lib = pkg._abi.lib()
Q, N, K = 17592182243329, 65536, int(os.environ.get("K", 4))
J = int(os.environ.get("J", 1024))
lctx = pkg.LweContext(pkg.Params(q=Q, n=N, k=K, sigma=3.19), key_seed=7, device=0)
r = torch.randint(0, Q, (J, K, N), dtype=torch.int64, device="cuda")
e1 = torch.randint(0, 8, (J, K, N), dtype=torch.int64, device="cuda")
u = torch.empty_like(r)
s = torch.cuda.current_stream().cuda_stream
seeds = np.arange(1, J + 1, dtype=np.uint64)
def t(fn, reps=10):
    for _ in range(2): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts)//2]
ms = t(lambda: lib.lsr_mlwe_matvec_batch_device(lctx.handle, r.data_ptr(), e1.data_ptr(), u.data_ptr(), J, None, s))
print(f"matvec commit (e1 given)   : {ms:.3f} ms per {J} -> {J/ms:.1f} K commits/s  roofline {J/ms*1e3*3*K*N*8/8e12:.3f}")
ms2 = t(lambda: lib.lsr_mlwe_matvec_batch_device(lctx.handle, r.data_ptr(), None, u.data_ptr(), J, seeds.ctypes.data, s), reps=5)
print(f"matvec commit (e1 on device): {ms2:.3f} ms per {J} -> {J/ms2:.1f} K commits/s  roofline(4.19MB) {J/ms2*1e3*2*K*N*8/8e12:.3f}")
print(f"  => on-device sampling of {J*K*N/1e6:.0f} M gaussians costs {ms2-ms:.3f} ms = {J*K*N/(ms2-ms)/1e6:.1f} G samples/s")
