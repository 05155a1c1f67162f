"""Creation / destruction latency of the handles a drop-in caller makes per proof (the reference's CLI builds a context per run):
ntt_context_create, lwe_context_create at the reference's parameters and at config 3's, lsr_quotient_plan_create."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package(); lib = pkg._abi.lib()
import torch; torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
def timed(make, close, reps=5):
    ts, tc = [], []
    for _ in range(reps):
        t0 = time.perf_counter(); h = make(); t1 = time.perf_counter(); close(h); t2 = time.perf_counter()
        ts.append(t1 - t0); tc.append(t2 - t1)
    return ts[0] * 1e3, float(np.median(ts[1:])) * 1e3, float(np.median(tc)) * 1e3
for label, make in (("ntt_context_create(q44, 4096)", lambda: pkg.NttContext(17592169062401, 4096)),
                    ("ntt_context_create(q44, 65536)", lambda: pkg.NttContext(17592182243329, 65536)),
                    ("lwe_context_create(n=4096, k=2)", lambda: pkg.LweContext(pkg.Params(q=17592186044417, n=4096, k=2, sigma=3.19))),
                    ("lwe_context_create(n=4096, k=2, key_seed)", lambda: pkg.LweContext(pkg.Params(q=17592186044417, n=4096, k=2, sigma=3.19), key_seed=5)),
                    ("lwe_context_create(n=65536, k=4)", lambda: pkg.LweContext(pkg.Params(q=17592182243329, n=65536, k=4, sigma=3.19))),
                    ("lsr_quotient_plan_create(4096)", lambda: pkg.QuotientPlan(4096, device=0))):
    first, later, close = timed(make, lambda h: h.close())
    print(f"{label:45s} first {first:8.2f} ms, then {later:8.2f} ms; free {close:6.2f} ms")
ctx = pkg.LweContext(pkg.Params(q=17592186044417, n=4096, k=2, sigma=3.19))
msg = np.array([1, 314, 628, 471, 471], dtype=np.uint64)
t0 = time.perf_counter(); p = lib.lwe_commit(ctx.handle, msg.ctypes.data, msg.size, 7); t1 = time.perf_counter()
print(f"first lwe_commit on a fresh context           {1e3*(t1-t0):8.2f} ms")
t0 = time.perf_counter(); rc = lib.lwe_verify_opening(ctx.handle, p, msg.ctypes.data, msg.size, None); t1 = time.perf_counter()
print(f"first lwe_verify_opening                      {1e3*(t1-t0):8.2f} ms (rc {rc})")
