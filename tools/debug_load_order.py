import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
order = sys.argv[1]
def load():
    L = ctypes.CDLL(os.path.join(ROOT, "lambda-snark-r_amd/lib/liblambda_snark_core.so"))
    L.lsr_last_error.restype = ctypes.c_char_p
    L.ntt_context_create.restype = ctypes.c_void_p; L.ntt_context_create.argtypes = [ctypes.c_uint64, ctypes.c_uint32]
    return L
if order == "lib_first":
    L = load(); print("lib first: device_count", L.lsr_device_count())
    import torch; print("torch avail", torch.cuda.is_available(), "lib count now", L.lsr_device_count())
    print("ctx", L.ntt_context_create(12289, 256), L.lsr_last_error())
elif order == "lib_first_nocount":
    L = load()
    import torch; print("torch avail", torch.cuda.is_available(), "lib count now", L.lsr_device_count())
    print("ctx", L.ntt_context_create(12289, 256), L.lsr_last_error())
else:
    import torch; print("torch avail", torch.cuda.is_available())
    L = load(); print("torch first: device_count", L.lsr_device_count()); print("ctx", L.ntt_context_create(12289, 256), L.lsr_last_error())
with open("/proc/self/maps") as f:
    libs = sorted({l.split()[-1] for l in f if "libamdhip64" in l or "libhsa-runtime" in l})
print(libs)
