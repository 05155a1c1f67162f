"""Would chunk lanes help the n = 2^16 transform?  The same 4096-polynomial batch as ONE call, and as two / four calls on streams of their
own (the context is stateless, so concurrent calls are allowed), under the chunk size of LAMBDA_SNARK_NTT_CHUNK_MIB.  HIP events."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
Q, N, B = 17592182243329, 65536, 4096
ctx = pkg.NttContext(Q, N, device=0)
polys = torch.randint(0, Q, (B, N), dtype=torch.int64, device="cuda")
main = torch.cuda.current_stream()
def run(lanes, inverse=False):
    streams = [torch.cuda.Stream() for _ in range(lanes)] if lanes > 1 else [main]
    per = B // lanes
    def once():
        if lanes > 1:
            for st in streams: st.wait_stream(main)
        for i, st in enumerate(streams):
            ptr = polys[i * per].data_ptr()
            (ctx.inverse_device if inverse else ctx.forward_device)(ptr, per, st.cuda_stream)
        if lanes > 1:
            for st in streams: main.wait_stream(st)
    for _ in range(3): once()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(main); once(); b.record(main); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
print("chunk MiB", os.environ.get("LAMBDA_SNARK_NTT_CHUNK_MIB", "256"), " ".join(f"lanes={l}: fwd {run(l):.3f} ms inv {run(l, True):.3f} ms |" for l in (1, 2, 4)))
