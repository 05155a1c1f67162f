"""Development microbench: forward / inverse NTT n=2^16 device-resident, HIP-event timed (torch current stream)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
Q, N = 17592182243329, int(os.environ.get("N", 65536))
if N <= 4096: Q = 17592169062401
B = int(os.environ.get("B", 4096))
ctx = pkg.NttContext(Q, N, device=0)
x = torch.randint(0, Q, (B, N), dtype=torch.int64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def t(fn, reps=20):
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts)//2], ts[0]
f = t(lambda: ctx.forward_device(x.data_ptr(), B, s))
i = t(lambda: ctx.inverse_device(x.data_ptr(), B, s))
print(f"chunk={os.environ.get('LAMBDA_SNARK_NTT_CHUNK_MIB','64')} N={N} B={B} fwd {f[0]:.3f} ms (min {f[1]:.3f}) = {B/f[0]/1e3:.3f} M NTT/s | inv {i[0]:.3f} ms (min {i[1]:.3f}) = {B/i[0]/1e3:.3f} M NTT/s", flush=True)
