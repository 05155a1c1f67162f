"""PCIe-inclusive rate of lsr_lwe_commit_batch_flat / lsr_lwe_verify_opening_batch_flat at the reference's parameters (n = 4096, k = 2)
for several message lengths (env MSGS, default "8,256,4096"; B = batch), pageable and page-locked rows, with the rate of the host key
derivation alone beside it (batches of >= 2^16 embedded words derive their keys on the device after the upload)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package(); lib = pkg._abi.lib()
ctx = pkg.LweContext(pkg.Params(q=17592186044417, n=4096, k=2, sigma=3.19), key_seed=11, device=0)
nb = int(os.environ.get("B", 2048))
words = ctx.commitment_words
def wall(fn, reps=5):
    fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))
for msg_len in [int(x) for x in os.environ.get("MSGS", "8,256,4096").split(",")]:
    rng = np.random.default_rng(msg_len)
    msgs = rng.integers(0, ctx.plain_modulus, size=(nb, msg_len), dtype=np.uint64)
    seeds = np.arange(1, nb + 1, dtype=np.uint64)
    rows = np.zeros((nb, words), dtype=np.uint64); verdicts = np.zeros(nb, dtype=np.int32); keys = np.zeros((nb, 4), dtype=np.uint64)
    t_c = wall(lambda: lib.lsr_lwe_commit_batch_flat(ctx.handle, msgs.ctypes.data, msg_len, nb, seeds.ctypes.data, rows.ctypes.data))
    t_v = wall(lambda: lib.lsr_lwe_verify_opening_batch_flat(ctx.handle, rows.ctypes.data, msgs.ctypes.data, msg_len, nb, verdicts.ctypes.data))
    t_k = wall(lambda: lib.lsr_lwe_commit_keys(ctx.handle, msgs.ctypes.data, msg_len, nb, seeds.ctypes.data, keys.ctypes.data), reps=3)
    pin = pkg.PinnedArray((nb, words))
    t_p = wall(lambda: lib.lsr_lwe_commit_batch_flat(ctx.handle, msgs.ctypes.data, msg_len, nb, seeds.ctypes.data, pin.array.ctypes.data))
    same = bool(np.array_equal(pin.array, rows)); pin.close()
    print(f"msg_len {msg_len:5d}: commit_batch_flat {nb/t_c/1e3:7.1f} K/s (page-locked rows {nb/t_p/1e3:7.1f} K/s, equal {same}), verify {nb/t_v/1e3:7.1f} K/s, "
          f"all open {bool((verdicts == 1).all())}; host key derivation alone {nb/t_k/1e3:7.1f} K/s")
