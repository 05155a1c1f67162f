"""Writes oracle_regression.json from the CPU oracle of this repo (run from the repo root:
`python tests/golden/make_golden.py`).  The reference cannot run here (SEAL absent), so these are
oracle-regression vectors, not reference outputs."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding  # noqa: E402


def sha(a):
    return hashlib.sha256(a.tobytes()).hexdigest()


def main():
    o = oracle_binding.load()
    out = {"ntt": [], "sampler": {}, "commit": []}
    for q, n in [(12289, 8), (12289, 256), (17592169062401, 1024), (17592169062401, 4096), (17592182243329, 8192)]:
        a = o.splitmix(0xC0FFEE, q, n)
        f = o.ntt_forward(q, n, a)
        out["ntt"].append({"q": q, "n": n, "seed": 0xC0FFEE, "psi": int(o.root(q, n)), "fwd_head": [int(x) for x in f[:8]], "fwd_sha256": sha(f),
                           "inv_of_input_sha256": sha(o.ntt_inverse(q, n, a))})
    cdf = o.gaussian_cdf(3.19)
    out["sampler"] = {"sigma": 3.19, "cdf_len": int(cdf.size), "cdf_head": [int(x) for x in cdf[:6]], "cdf_sha256": sha(cdf),
                      "seeded_head": [int(x) for x in o.sample_gaussian_seeded(32, 3.19, 0x1234, 5, 1)],
                      "stream_head": [int(x) for x in o.stream_words(0x1234, 5, 1, 0, 4)]}
    for (q, n, k, key_seed, seed, msg) in [(17592186044417, 4096, 2, 0x1234, 0x5678, [1, 314, 628, 471, 471]), (12289, 256, 2, 7, 9, [1, 2, 3, 4]),
                                          (17592186044417, 1024, 4, 11, 13, list(range(40)))]:
        c = o.lwe_commit(q, n, k, 3.19, key_seed, msg, seed)
        out["commit"].append({"q": q, "n": n, "k": k, "sigma": 3.19, "key_seed": key_seed, "seed": seed, "msg": msg, "words": int(c.size),
                              "head": [int(x) for x in c[:8]], "sha256": sha(c)})
    with open(os.path.join(HERE, "oracle_regression.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote oracle_regression.json")


if __name__ == "__main__":
    main()
