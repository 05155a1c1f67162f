"""Throughput of the device-resident quotient path (development aid).  env M (constraints), B (instances)."""
import os, sys, time
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry

pkg = entry.load_package()
if os.environ.get("LIBVARIANT"):          # an experiment build of the library (csrc/Makefile VARIANT=...), loaded explicitly
    pkg._abi.use_library(os.path.join(os.path.dirname(pkg._abi.LIB_PATH), "liblambda_snark_core_%s.so" % os.environ["LIBVARIANT"]))
m = int(os.environ.get("M", 4096)); batch = int(os.environ.get("B", 4096)); reps = int(os.environ.get("REPS", 10))
plan = pkg.QuotientPlan(m, device=0)
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
def rnd():
    return torch.randint(-2**63, 2**63 - 1, (batch, m), dtype=torch.int64, device="cuda", generator=gen)   # any 64-bit word
a, b, c = rnd(), rnd(), rnd()
if os.environ.get("VALID", "1") == "1":      # satisfied instances: c = a * b on the domain
    field = pkg.CyclicNtt(max(m, 2))
    assert pkg._abi.lib().lsr_ntt_mul_pointwise_device(field.handle, c.data_ptr(), a.data_ptr(), b.data_ptr(), batch * m, 0) == 0
    torch.cuda.synchronize()
dq = torch.empty_like(a); dl = torch.empty(batch, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    plan.quotient_device(a.data_ptr(), b.data_ptr(), c.data_ptr(), batch, dq.data_ptr(), dl.data_ptr(), s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    plan.quotient_device(a.data_ptr(), b.data_ptr(), c.data_ptr(), batch, dq.data_ptr(), dl.data_ptr(), s)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
words = batch * m
assert (dl.cpu() > 0).all() == (os.environ.get("VALID", "1") == "1")
print(f"m={m} batch={batch}: {dt*1e3:.3f} ms/pass, {batch/dt:.0f} quotients/s, {words/dt/1e9:.2f} G constraints/s, "
      f"input+output bytes at {4*words*8/dt/1e9:.0f} GB/s", flush=True)
