"""Would two lanes help the quotient pipeline (six dependent launches per pass)?  One plan on 4096 instances of m = 4096 against two plans
on 2048 instances each, on streams of their own.  HIP events."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
m, batch = int(os.environ.get("M", 4096)), int(os.environ.get("B", 4096))
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
a, b = (torch.randint(-2**63, 2**63 - 1, (batch, m), dtype=torch.int64, device="cuda", generator=gen) for _ in range(2))
c = torch.empty_like(a)
field = pkg.CyclicNtt(m)
assert pkg._abi.lib().lsr_ntt_mul_pointwise_device(field.handle, c.data_ptr(), a.data_ptr(), b.data_ptr(), batch * m, 0) == 0
dq = torch.empty_like(a); dl = torch.empty(batch, dtype=torch.int32, device="cuda")
main = torch.cuda.current_stream()
def run(lanes):
    plans = [pkg.QuotientPlan(m, device=0) for _ in range(lanes)]
    streams = [torch.cuda.Stream() for _ in range(lanes)] if lanes > 1 else [main]
    per = batch // lanes
    def once():
        if lanes > 1:
            for st in streams: st.wait_stream(main)
        for i, (p, st) in enumerate(zip(plans, streams)):
            p.quotient_device(a[i * per].data_ptr(), b[i * per].data_ptr(), c[i * per].data_ptr(), per, dq[i * per].data_ptr(), dl[i * per:].data_ptr(), st.cuda_stream)
        if lanes > 1:
            for st in streams: main.wait_stream(st)
    for _ in range(3): once()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main); once(); e1.record(main); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ok = bool((dl > 0).all().item())
    for p in plans: p.close()
    return float(np.median(ts)), ok
print(" ".join("lanes=%d: %.3f ms (valid %s) |" % ((l,) + run(l)) for l in (1, 2, 3, 4, 8, 1, 4)))
