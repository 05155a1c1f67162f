"""PCIe-inclusive rate of the host-buffer batched entry points (ntt_forward_batch) — never the headline value."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package(); lib = pkg._abi.lib()
for q, n, batch in [(17592182243329, 65536, 1024), (17592169062401, 4096, 16384)]:
    ctx = pkg.NttContext(q, n)
    a = np.random.default_rng(1).integers(0, q, size=(batch, n), dtype=np.uint64)
    lib.ntt_forward_batch(ctx.handle, a.ctypes.data, batch)
    t0 = time.perf_counter(); reps = 3
    for _ in range(reps): lib.ntt_forward_batch(ctx.handle, a.ctypes.data, batch)
    dt = (time.perf_counter() - t0) / reps
    gb = batch * n * 8 / 1e9
    print(f"ntt_forward_batch host buffers n={n} batch={batch} ({gb:.2f} GB each way): {dt*1e3:.1f} ms = {batch/dt/1e3:.1f} K NTT/s, {2*gb/dt:.1f} GB/s over PCIe (pageable memory)")
