"""Device-resident full commitments and openings: rows/s of lsr_lwe_commit_rows_device / lsr_lwe_verify_rows_device (keys, messages
and rows all in device memory; HIP events on the calling stream), for the fused pipelines and for the general kernels of the same
library (a context created with LAMBDA_SNARK_COMMIT_FUSED=0).  env: N (ring degree), K (rank), J (batch), MSG (message words),
REPS, GENERAL=0|1 (also time the general kernels).  Prints one JSON line."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry

pkg = entry.load_package()
if os.environ.get("LIBVARIANT"):          # an experiment build of the library (csrc/Makefile VARIANT=...), loaded explicitly
    pkg._abi.use_library(os.path.join(os.path.dirname(pkg._abi.LIB_PATH), "liblambda_snark_core_%s.so" % os.environ["LIBVARIANT"]))
lib = pkg._abi.lib()
N, K = int(os.environ.get("N", 4096)), int(os.environ.get("K", 2))
J, MSG, REPS = int(os.environ.get("J", 16384)), int(os.environ.get("MSG", 16)), int(os.environ.get("REPS", 10))
Q = 17592169062401 if N <= 4096 else 0
if os.environ.get("WIDE") == "1":          # the 60-bit prime for reference-range linear combinations (general kernels, u64 flavour)
    Q = int(lib.lsr_lwe_wide_modulus(N))


def make(fused):
    if fused:
        os.environ.pop("LAMBDA_SNARK_COMMIT_FUSED", None)
    else:
        os.environ["LAMBDA_SNARK_COMMIT_FUSED"] = "0"
    ctx = pkg.LweContext(pkg.Params(q=Q, n=N, k=K, sigma=3.19), key_seed=99, device=0)
    os.environ.pop("LAMBDA_SNARK_COMMIT_FUSED", None)
    return ctx


def timed(fn, reps):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


out = {"n": N, "k": K, "batch": J, "msg_words": MSG}
rng = np.random.default_rng(1)
ctxs = [("fused", make(True))] + ([("general", make(False))] if os.environ.get("GENERAL", "1") == "1" else [])
words = lib.lsr_lwe_commitment_words(ctxs[0][1].handle)
msgs = rng.integers(0, ctxs[0][1].plain_modulus, size=(J, MSG), dtype=np.uint64)
seeds = rng.integers(1, 2**63, size=J, dtype=np.uint64)
keys = np.zeros((J, 4), dtype=np.uint64)
assert lib.lsr_lwe_commit_keys(ctxs[0][1].handle, msgs.ctypes.data, MSG, J, seeds.ctypes.data, keys.ctypes.data) == 0
d_msgs = torch.from_numpy(msgs.view(np.int64)).cuda()
d_keys = torch.from_numpy(keys.view(np.int64)).cuda()
s = torch.cuda.current_stream().cuda_stream
ref_rows = None
for name, ctx in ctxs:
    out[name + "_pipeline"] = lib.lsr_lwe_pipeline(ctx.handle).decode()
    d_rows = torch.zeros((J, words), dtype=torch.int64, device="cuda")
    res = torch.zeros(J, dtype=torch.int32, device="cuda")
    commit = lambda: lib.lsr_lwe_commit_rows_device(ctx.handle, d_msgs.data_ptr(), MSG, J, d_keys.data_ptr(), d_rows.data_ptr(), s)
    verify = lambda: lib.lsr_lwe_verify_rows_device(ctx.handle, d_rows.data_ptr(), d_msgs.data_ptr(), MSG, J, res.data_ptr(), s)
    assert commit() == 0
    ms_c = timed(commit, REPS)
    assert verify() == 0
    ms_v = timed(verify, REPS)
    torch.cuda.synchronize()
    assert int(res.sum().item()) == J, "every row must open"
    if ref_rows is None:
        ref_rows = d_rows
    else:
        out["rows_equal"] = bool(torch.equal(ref_rows, d_rows))
    row_bytes = words * 8
    out[name] = {"commit_ms": ms_c, "commits_per_s": J / ms_c * 1e3, "commit_roofline_frac": J * row_bytes / (ms_c * 1e-3) / 8e12,
                 "verify_ms": ms_v, "openings_per_s": J / ms_v * 1e3, "verify_roofline_frac": J * row_bytes / (ms_v * 1e-3) / 8e12}
out["row_bytes"] = words * 8
print(json.dumps(out))
