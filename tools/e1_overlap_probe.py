"""How much of a stand-alone sampler pass hides under the mixed-launch commit pipeline?  Two contexts with the same keys (calls on ONE
context are ordered behind each other): the e1-given workload on one stream, lsr_lwe_sample_blinding_device for as many vectors on
another, alone and together.  HIP events on the default stream around both."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package(); lib = pkg._abi.lib()
Q, N, K, J = 17592182243329, 65536, 4, 1024
a = pkg.LweContext(pkg.Params(q=Q, n=N, k=K, sigma=3.19), key_seed=7, device=0)
b = a.replicate(0)
r = torch.randint(0, Q, (J, K, N), dtype=torch.int64, device="cuda")
e1 = torch.randint(0, 8, (J, K, N), dtype=torch.int64, device="cuda")
u = torch.empty_like(r); e1b = torch.empty_like(r)
seeds = np.arange(1, J + 1, dtype=np.uint64)
main = torch.cuda.current_stream(); s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(priority=int(os.environ.get("PRIO", "0")))
def commit(): assert lib.lsr_mlwe_matvec_batch_device(a.handle, r.data_ptr(), e1.data_ptr(), u.data_ptr(), J, None, s1.cuda_stream) == 0
def sample(): assert lib.lsr_lwe_sample_blinding_device(b.handle, e1b.data_ptr(), J, seeds.ctypes.data, s2.cuda_stream) == 0
def fused(): assert lib.lsr_mlwe_matvec_batch_device(a.handle, r.data_ptr(), None, u.data_ptr(), J, seeds.ctypes.data, s1.cuda_stream) == 0
def timed(fns):
    def once():
        s1.wait_stream(main); s2.wait_stream(main)
        for f in fns: f()
        main.wait_stream(s1); main.wait_stream(s2)
    for _ in range(2): once()
    torch.cuda.synchronize(); ts = []
    for _ in range(7):
        x, y = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        x.record(main); once(); y.record(main); torch.cuda.synchronize(); ts.append(x.elapsed_time(y))
    return float(np.median(ts))
print(f"commit (e1 given) alone {timed([commit]):.3f} ms | sampler alone {timed([sample]):.3f} ms | both at once {timed([commit, sample]):.3f} ms | "
      f"product path with e1 sampled in the strided rounds {timed([fused]):.3f} ms")
