"""Development probe: lsr_lwe_commit_rows_device / lsr_lwe_verify_rows_device captured into HIP graphs (torch.cuda.graph) and replayed
several times, at n = 4096 (tile pipeline) and n = 2^16 (fused pipeline).  Found the captured-hipMemsetAsync problem (profiles/README.md)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
q, n, k, batch, msg_len = 17592169062401, 4096, 2, 12, 9
ctx = pkg.LweContext(pkg.Params(q=q, n=n, k=k, sigma=3.19), key_seed=0x5EED)
rng = np.random.default_rng(8)
msgs = rng.integers(0, ctx.plain_modulus, size=(batch, msg_len), dtype=np.uint64)
seeds = rng.integers(1, 2**63, size=batch, dtype=np.uint64)
keys = ctx.commit_keys(msgs, seeds)
d_msgs = torch.from_numpy(msgs.view(np.int64)).cuda(); d_keys = torch.from_numpy(keys.view(np.int64)).cuda()
rows = torch.zeros((batch, ctx.commitment_words), dtype=torch.int64, device="cuda")
res = torch.zeros(batch, dtype=torch.int32, device="cuda")
side = torch.cuda.Stream()
commit = lambda st: ctx.commit_rows_device(d_msgs.data_ptr(), msg_len, batch, d_keys.data_ptr(), rows.data_ptr(), st)
verify = lambda st: ctx.verify_rows_device(rows.data_ptr(), d_msgs.data_ptr(), msg_len, batch, res.data_ptr(), st)
with torch.cuda.stream(side):
    commit(side.cuda_stream); verify(side.cuda_stream)
side.synchronize()
want = rows.clone()
print("eager", res.tolist())
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    verify(torch.cuda.current_stream().cuda_stream)
for rep in range(4):
    res.fill_(7); torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    print("verify graph replay", rep, res.tolist())
res.fill_(7); verify(torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize(); print("eager after", res.tolist())
res.fill_(7); torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize(); print("replay after eager", res.tolist())
res.fill_(7); torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize(); print("replay again", res.tolist())
# the same at n = 2^16 (fused pipeline: clear + three launches per chunk + verdict)
ctx2 = pkg.LweContext(pkg.Params(q=17592182243329, n=65536, k=2, sigma=3.19), key_seed=0x5EED)
keys2 = ctx2.commit_keys(msgs, seeds); d_keys2 = torch.from_numpy(keys2.view(np.int64)).cuda()
rows2 = torch.zeros((batch, ctx2.commitment_words), dtype=torch.int64, device="cuda")
ctx2.commit_rows_device(d_msgs.data_ptr(), msg_len, batch, d_keys2.data_ptr(), rows2.data_ptr(), side.cuda_stream)
v2 = lambda st: ctx2.verify_rows_device(rows2.data_ptr(), d_msgs.data_ptr(), msg_len, batch, res.data_ptr(), st)
v2(side.cuda_stream); side.synchronize(); print("n=65536 eager", res.tolist(), "t", ctx.plain_modulus, ctx2.plain_modulus, "rows with every word below t2:", [bool((msgs[j] < ctx2.plain_modulus).all()) for j in range(batch)])
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2, stream=side):
    v2(torch.cuda.current_stream().cuda_stream)
for rep in range(3):
    res.fill_(7); torch.cuda.synchronize(); g2.replay(); torch.cuda.synchronize()
    print("n=65536 verify graph replay", rep, res.tolist())
