#!/bin/bash
# Round-3 profile artefacts (copied to profiles/r03_* afterwards), every JSON stamped with the provenance of the library that was
# measured (tools/provenance.py: source hash, library hash, git HEAD recorded at build time):
#   bench.json                  the bench line of the default command
#   bench_kernel_stats.csv      rocprofv3 --kernel-trace --stats of the same command (--no-cpu: the CPU leg is not a GPU kernel)
#   roofline_inputs.json        FETCH_SIZE / WRITE_SIZE (separate --pmc passes) of the n = 2^16 transform
#   pmc_commit_traffic.json     the same for the config-3 commitment pipelines (mixed launches; e1 sampled in the pass)
#   pmc_full_commit.json        the same for whole commitments / openings (tile pipeline at n = 4096, fused at n = 2^16)
# rocprofv3 gets `python3 <script>` directly after `--` (no env / shell wrappers: the profiler's preloaded library initialises the GPU).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r03p
rm -rf $out && mkdir -p $out
python3 bench.py > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
echo "bench done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu > $out/bench_under_rocprof.json 2> $out/stats.log
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/bench_kernel_stats.csv && rm -rf $out/stats
echo "kernel stats done"
export J=256
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $out/pmc_ntt_$c -- python3 tools/ntt_bench.py > $out/pmc_ntt_$c.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $out/pmc_commit_$c -- python3 tools/commit_bench.py > $out/pmc_commit_$c.log 2>&1
  N=4096 K=2 J=4096 GENERAL=0 REPS=2 timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $out/pmc_rows4096_$c -- python3 tools/commit_rows_bench.py > $out/pmc_rows4096_$c.log 2>&1
  N=65536 K=4 J=256 GENERAL=0 REPS=2 timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $out/pmc_rows65536_$c -- python3 tools/commit_rows_bench.py > $out/pmc_rows65536_$c.log 2>&1
  echo "pmc $c done"
done
python3 - $out <<'PY'
import csv, glob, sys, collections, json, os
sys.path.insert(0, "tools")
import provenance
out = sys.argv[1]
stamp = provenance.provenance()
def per_kernel(prefix, total=False):
    res = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(f"{out}/{prefix}_{c}/*/*counter_collection.csv")
        if not files: continue
        acc = collections.defaultdict(float); cnt = collections.Counter()
        for r in csv.DictReader(open(files[0])):
            if "lsr" not in r["Kernel_Name"] or r["Counter_Name"] != c: continue
            k = r["Kernel_Name"].split("(")[0].replace("void lsr::", "")
            acc[k] += float(r["Counter_Value"]); cnt[k] += 1
        for k in acc:
            res[k][c + "_KiB_total"] = acc[k]; res[k][c + "_KiB_per_dispatch"] = acc[k] / cnt[k]; res[k]["dispatches"] = cnt[k]
    return res
def moved(d, key="_KiB_per_dispatch"):   # bytes: FETCH_SIZE doubled (gfx950 reports half of streamed read bytes, MI355X_MICROARCH.md) + WRITE_SIZE
    return (2 * d.get("FETCH_SIZE" + key, 0) + d.get("WRITE_SIZE" + key, 0)) * 1024
ntt = per_kernel("pmc_ntt")
fwd = [k for k in ntt if ("strided_round" in k and "false, false, true" in k) or "tile_forward" in k]
inv = [k for k in ntt if ("strided_round" in k and "true, true, false" in k) or "tile_inverse" in k]
json.dump({"provenance": stamp,
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/ntt_bench.py, n=2^16, 4096 polys, 512 polys per dispatch; FETCH_SIZE doubled per MI355X_MICROARCH.md",
           "forward_bytes_per_transform": sum(moved(ntt[k]) for k in fwd) / 512, "inverse_bytes_per_transform": sum(moved(ntt[k]) for k in inv) / 512,
           "algorithmic_bytes_per_transform": 1048576, "kernels_forward": fwd, "kernels_inverse": inv, "per_kernel": ntt}, open(f"{out}/roofline_inputs.json", "w"), indent=1)
com = per_kernel("pmc_commit")
# commit_bench.py (J = 256): 2 warm-up + 10 timed calls of the e1-given pipeline (mixed launches) and 2 + 5 of the e1-sampled one
mixed = [k for k in com if "mlwe_mixed" in k]
per_commit = sum(moved(com[k], "_KiB_total") for k in mixed) / (12 * 256)
sampled = [k for k in com if "mlwe_mid_fused8" in k or "ntt_strided_round_sampl" in k]
per_commit_sampled = sum(moved(com[k], "_KiB_total") for k in sampled) / (7 * 256)
json.dump({"provenance": stamp,
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on J=256 tools/commit_bench.py, rank 4, n=2^16; all mlwe_mixed dispatches of the 12 e1-given calls, all dispatches of the 7 e1-sampled calls; FETCH_SIZE doubled",
           "bytes_per_commit": per_commit, "algorithmic_bytes_per_commit": 6291456,
           "bytes_per_commit_e1_sampled_in_pass": per_commit_sampled, "algorithmic_bytes_per_commit_e1_sampled": 4194304,
           "kernel": "mlwe_mixed<4>: one launch = middle stage of chunk t (12 fwd stages x 4, A^T product, 12 inv stages x 4) + forward strided round of chunk t+1 + inverse strided round (+ e1) of chunk t-1; 32-vector chunks, two lanes",
           "per_kernel": com}, open(f"{out}/pmc_commit_traffic.json", "w"), indent=1)
full = {}
for label, prefix, calls, batch, row_bytes in (("n4096_k2", "pmc_rows4096", 5, 4096, 98344), ("n65536_k4", "pmc_rows65536", 5, 256, 2621480)):
    d = per_kernel(prefix)
    ck = [k for k in d if k.startswith("commit_") or "mlwe_mid_general<4, 4>" in k or "mlwe_mid_general<4, 1>" in k]
    vk = [k for k in d if k.startswith("verify_")]
    # commit_rows_bench.py (REPS = 2): 1 + 2 warm + 2 timed commit calls = 5, 1 + 2 + 2 verify calls = 5 (the one-column middle stage of
    # an opening at n = 2^16 shares its kernel with the commitment's second pass: attributed to the commitment here)
    full[label] = {"row_bytes": row_bytes, "commit_bytes_per_row": sum(moved(d[k], "_KiB_total") for k in ck) / (calls * batch),
                   "verify_bytes_per_row": sum(moved(d[k], "_KiB_total") for k in vk) / (calls * batch), "per_kernel": d}
json.dump({"provenance": stamp, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/commit_rows_bench.py; FETCH_SIZE doubled",
           **full}, open(f"{out}/pmc_full_commit.json", "w"), indent=1)
b = json.load(open(f"{out}/bench.json"))
print("forward bytes/transform", sum(moved(ntt[k]) for k in fwd) / 512, "commit bytes", per_commit / 2**20, "MiB; e1 sampled", per_commit_sampled / 2**20, "MiB")
print({k: (v["commit_bytes_per_row"], v["verify_bytes_per_row"], v["row_bytes"]) for k, v in full.items()})
e = b["extra"]
print("value", b["value"], "frac", b["roofline"]["frac"], "commits/s", e["commits_per_s"], e["commit_roofline_frac"], "e1dev", e.get("commits_per_s_e1_on_device"))
print("bench provenance == stamp:", e["provenance"]["source_sha256"] == stamp["source_sha256"])
PY
rm -rf $out/pmc_*_FETCH_SIZE $out/pmc_*_WRITE_SIZE
grep -E "mlwe_|strided|tile_forward|tile_inverse|commit_t|verify_t" $out/bench_kernel_stats.csv | cut -d, -f1-4 | cut -c1-110
