"""Host-side Fiat–Shamir throughput: lsr_fs_challenge_batch_flat over 2048 reference-size commitments (98 KB rows)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package(); lib = pkg._abi.lib()
count, W = 2048, 12293
rows = np.random.default_rng(1).integers(0, 2**64, size=(count, W), dtype=np.uint64)
al = np.zeros(count, dtype=np.uint64)
for th in (1, 2, 4, 8, 16, 32):
    t = time.perf_counter()
    assert lib.lsr_fs_challenge_batch_flat(None, 0, rows.ctypes.data, W, count, 17592186044417, al.ctypes.data, None, th) == 0
    dt = time.perf_counter() - t
    print(f"{th:2d} threads: {dt*1e3:7.1f} ms = {count/dt/1e3:6.1f} K transcripts/s, {count*W*8/dt/1e9:5.2f} GB/s", flush=True)

# the same rows hashed on the GPU, one lane per transcript (lsr_fs_challenge_batch_device)
import torch
if torch.cuda.is_available():
    for cnt in (2048, 16384):
        big = torch.randint(-2**63, 2**63 - 1, (cnt, W), dtype=torch.int64, device="cuda")
        d_al = torch.zeros(cnt, dtype=torch.int64, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(2): lib.lsr_fs_challenge_batch_device(None, 0, big.data_ptr(), W, cnt, 17592186044417, d_al.data_ptr(), None, s)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(3): lib.lsr_fs_challenge_batch_device(None, 0, big.data_ptr(), W, cnt, 17592186044417, d_al.data_ptr(), None, s)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
        print(f"GPU, {cnt:6d} transcripts: {dt*1e3:7.1f} ms = {cnt/dt/1e3:7.1f} K transcripts/s, {cnt*W*8/dt/1e9:6.2f} GB/s", flush=True)
