"""Quick GPU parity probe (development aid; the real tests are tests/test_*_gpu.py)."""
import ctypes, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(ROOT, "lambda-snark-r_amd/lib/liblambda_snark_core.so"))
O = ctypes.CDLL(os.path.join(ROOT, "oracle/liblsr_oracle.so"))
u64, u32, vp, sz = ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t
L.ntt_context_create.restype = vp; L.ntt_context_create.argtypes = [u64, u32]
L.ntt_context_free.argtypes = [vp]
L.ntt_forward.argtypes = [vp, vp, u32]; L.ntt_inverse.argtypes = [vp, vp, u32]
L.ntt_forward_batch.argtypes = [vp, vp, sz]; L.ntt_inverse_batch.argtypes = [vp, vp, sz]
L.ntt_mul_pointwise.argtypes = [vp, vp, vp, vp, u32]
L.lsr_set_arith_mode.argtypes = [ctypes.c_int]
L.lsr_ntt_context_uses_f64.argtypes = [vp]
L.lsr_last_error.restype = ctypes.c_char_p
O.oracle_ntt_create.restype = vp; O.oracle_ntt_create.argtypes = [u64, u32]
O.oracle_ntt_forward_batch.argtypes = [vp, vp, sz]; O.oracle_ntt_inverse_batch.argtypes = [vp, vp, sz]
O.oracle_splitmix_fill.argtypes = [u64, u64, vp, sz]
O.oracle_ntt_mul_pointwise.argtypes = [vp, vp, vp, vp, u32]

ok = True
cases = [(12289, 256, 5), (12289, 2, 3), (12289, 4, 3), (12289, 8, 2), (12289, 16, 2), (12289, 32, 2), (12289, 64, 2), (12289, 128, 2),
         (12289, 512, 3), (12289, 1024, 5), (12289, 2048, 3), (17592169062401, 4096, 7), (17592169062401, 2048, 3), (17592169062401, 64, 70),
         (17592182243329, 8192, 3), (17592182243329, 16384, 2), (17592182243329, 32768, 2), (17592182243329, 65536, 3)]
# a 60-bit prime = 1 mod 2^18 for the u64-only path
for mode in (0, 1):
    L.lsr_set_arith_mode(mode)
    for q, n, B in cases + [(1152921504606584833, 4096, 2), (1152921504606584833, 65536, 2), (1152921504606584833, 131072, 1)]:
        t = O.oracle_ntt_create(q, n)
        if not t:
            print("oracle rejects", q, n); continue
        c = L.ntt_context_create(q, n)
        if not c:
            print("FAIL create", q, n, L.lsr_last_error()); ok = False; continue
        a = np.zeros(B * n, dtype=np.uint64)
        O.oracle_splitmix_fill(0xDEADBEEF + n, q, a.ctypes.data, B * n)
        a[0] = q - 1; a[-1] = q - 1; a[1] = 0
        ref = a.copy(); O.oracle_ntt_forward_batch(t, ref.ctypes.data, B)
        got = a.copy(); rc = L.ntt_forward_batch(c, got.ctypes.data, B)
        f_ok = rc == 0 and (got == ref).all()
        back = got.copy(); rc2 = L.ntt_inverse_batch(c, back.ctypes.data, B)
        i_ok = rc2 == 0 and (back == a).all()
        one = a[:n].copy(); L.ntt_forward(c, one.ctypes.data, n)
        s_ok = (one == ref[:n]).all()
        pw = np.zeros(n, dtype=np.uint64); pr = np.zeros(n, dtype=np.uint64)
        L.ntt_mul_pointwise(c, pw.ctypes.data, a[:n].ctypes.data, ref[:n].ctypes.data, n)
        O.oracle_ntt_mul_pointwise(t, pr.ctypes.data, a[:n].ctypes.data, ref[:n].ctypes.data, n)
        p_ok = (pw == pr).all()
        print(f"mode={mode} f64={L.lsr_ntt_context_uses_f64(c)} q={q} n={n} B={B} fwd={f_ok} inv={i_ok} single={s_ok} pw={p_ok}", flush=True)
        if not (f_ok and i_ok and s_ok and p_ok):
            ok = False
            bad = np.nonzero(got != ref)[0]
            print("   first mismatches", bad[:8], got[bad[:4]], ref[bad[:4]])
        L.ntt_context_free(c)
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
