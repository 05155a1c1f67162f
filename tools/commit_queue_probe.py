"""Development probe (round 2): the commit loop timed after each piece of bench.py's set-up.  It found the 7 % gap between bench.py and
tools/commit_bench.py: with one more stream open in the process the pipeline's two side streams shared one hardware queue
(profiles/r02_commit_hw_queues.txt)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package(); lib = pkg._abi.lib()
Q, N, K, J = 17592182243329, 65536, 4, 1024
seeds = np.arange(1, J + 1, dtype=np.uint64) * np.uint64(0x9E3779B9)
def commit_timing(tag):
    s = torch.cuda.current_stream().cuda_stream
    lctx = pkg.LweContext(pkg.Params(q=Q, n=N, k=K, sigma=3.19), key_seed=0xC0DE + 1, device=0)
    r = torch.empty((J, K, N), dtype=torch.int64, device="cuda")
    lib.lsr_fill_splitmix_device(r.data_ptr(), J, K * N, 0xC0FFEE, Q, s)
    e1 = torch.empty_like(r)
    lib.lsr_lwe_sample_blinding_device(lctx.handle, e1.data_ptr(), J, seeds.ctypes.data, s)
    u = torch.empty_like(r); r_work = torch.empty_like(r); r_work.copy_(r)
    def run(reps):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record(); lib.lsr_mlwe_matvec_batch_device(lctx.handle, r_work.data_ptr(), e1.data_ptr(), u.data_ptr(), J, None, s); b.record()
        torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in ev]
    run(3); ts = run(10)
    print(f"{tag:52s} median {np.median(ts):.3f} min {min(ts):.3f} max {max(ts):.3f} ms", flush=True)
    lctx.close(); del r, e1, u, r_work; torch.cuda.empty_cache()
commit_timing("fresh process")
torch.cuda.set_device(0)
commit_timing("+ torch.cuda.set_device(0)")
import torch.distributed as dist
commit_timing("+ import torch.distributed")
gen = torch.Generator(device="cuda"); gen.manual_seed(0xDEADBEEF)
commit_timing("+ torch.Generator(cuda)")
ctx = pkg.NttContext(Q, N, device=0)
s = torch.cuda.current_stream().cuda_stream
polys = torch.empty((64, N), dtype=torch.int64, device="cuda")
lib.lsr_fill_splitmix_device(polys.data_ptr(), 64, N, 0xDEADBEEF, Q, s)
torch.cuda.synchronize()
commit_timing("+ NttContext, polys")
reference_copy = polys[:8].clone()
commit_timing("+ tensor clone")
ctx.forward_device(polys.data_ptr(), 64, s); torch.cuda.synchronize()
commit_timing("+ forward_device")
host = polys[:64].cpu().numpy().view(np.uint64).copy()
commit_timing("+ .cpu() copy")
ok = bool(torch.equal(polys[:8], reference_copy))
commit_timing("+ torch.equal")
