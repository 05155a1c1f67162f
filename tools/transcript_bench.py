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
