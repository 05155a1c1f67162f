"""Development aid: device timeline of the two-lane commitment schedule (library built with -DLSR_LANE_TRACE;
LAMBDA_SNARK_CORE_LIB points at it).  One warm call, then one traced call of 512 rank-4 witness vectors."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package(); lib = pkg._abi.lib()
Q, N, K, J = 17592182243329, 65536, 4, int(os.environ.get("J", 512))
lctx = pkg.LweContext(pkg.Params(q=Q, n=N, k=K, sigma=3.19), key_seed=7, device=0)
r = torch.randint(0, Q, (J, K, N), dtype=torch.int64, device="cuda")
e1 = torch.randint(0, 8, (J, K, N), dtype=torch.int64, device="cuda")
u = torch.empty_like(r)
s = torch.cuda.current_stream().cuda_stream
for i in range(3):
    print(f"--- call {i}", file=sys.stderr, flush=True)
    lib.lsr_mlwe_matvec_batch_device(lctx.handle, r.data_ptr(), e1.data_ptr(), u.data_ptr(), J, None, s)
    torch.cuda.synchronize()
