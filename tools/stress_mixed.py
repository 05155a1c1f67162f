"""Stress run of the mixed-launch commitment schedule: the same batch many times, every word compared with the three-launch schedule's
result (development aid; looks for rare ordering / workspace-reuse hazards between launches and lanes).  env J, K, REPS."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package(); lib = pkg._abi.lib()
Q, N = 17592182243329, 65536
J, K, REPS = int(os.environ.get("J", 1000)), int(os.environ.get("K", 4)), int(os.environ.get("REPS", 300))
os.environ["LAMBDA_SNARK_COMMIT_MIXED"] = "0"       # read once per context, at creation
three = pkg.LweContext(pkg.Params(q=Q, n=N, k=K, sigma=3.19), key_seed=11, device=0)
os.environ["LAMBDA_SNARK_COMMIT_MIXED"] = "1"
lctx = pkg.LweContext(pkg.Params(q=Q, n=N, k=K, sigma=3.19), key_seed=11, device=0)
s = torch.cuda.current_stream().cuda_stream
r = torch.empty((J, K, N), dtype=torch.int64, device="cuda")
assert lib.lsr_fill_splitmix_device(r.data_ptr(), J, K * N, 0xABCDEF, Q, s) == 0
e1 = torch.empty_like(r)
seeds = np.arange(1, J + 1, dtype=np.uint64) * np.uint64(0x9E3779B9)
assert lib.lsr_lwe_sample_blinding_device(lctx.handle, e1.data_ptr(), J, seeds.ctypes.data, s) == 0
want = torch.empty_like(r); got = torch.empty_like(r)
assert lib.lsr_mlwe_matvec_batch_device(three.handle, r.data_ptr(), e1.data_ptr(), want.data_ptr(), J, None, s) == 0
torch.cuda.synchronize()
bad = 0
side = torch.cuda.Stream()
for it in range(REPS):
    lanes = "2"
    got.zero_()
    # back-to-back calls without a host synchronisation in between every third iteration, and one on another stream
    st = side.cuda_stream if it % 5 == 4 else s
    if it % 5 == 4: side.wait_stream(torch.cuda.current_stream())
    assert lib.lsr_mlwe_matvec_batch_device(lctx.handle, r.data_ptr(), e1.data_ptr(), got.data_ptr(), J, None, st) == 0
    if it % 5 == 4: torch.cuda.current_stream().wait_stream(side)
    if not torch.equal(got, want):
        bad += 1
        print("MISMATCH at iteration", it, "lanes", lanes, flush=True)
    if it % 50 == 49: print(f"{it + 1} iterations, {bad} mismatches", flush=True)
print(f"done: {REPS} iterations of {J} rank-{K} vectors, {bad} mismatches")
sys.exit(1 if bad else 0)
