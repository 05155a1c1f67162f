#!/bin/bash
# Round-2 profile artefacts (copied to profiles/r02_* afterwards): kernel-trace stats of the exact bench command, FETCH_SIZE /
# WRITE_SIZE (separate --pmc passes) of the n = 2^16 transform and of the commit pipeline, and the bench line itself.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02
rm -rf $out && mkdir -p $out
python3 bench.py > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu > $out/bench_under_rocprof.json 2> $out/stats.log
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/bench_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $out/pmc_ntt_$c -- python3 tools/ntt_bench.py > $out/pmc_ntt_$c.log 2>&1
  J=256 LAMBDA_SNARK_COMMIT_STREAMS=1 timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $out/pmc_commit_$c -- python3 tools/commit_bench.py > $out/pmc_commit_$c.log 2>&1
done
python3 - $out <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
def per_kernel(prefix):
    res = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"{out}/{prefix}_{c}/*/*counter_collection.csv")[0]
        acc = collections.defaultdict(float); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void lsr::", "")
            if "lsr" not in r["Kernel_Name"] or r["Counter_Name"] != c: continue
            acc[k] += float(r["Counter_Value"]); cnt[k] += 1
        for k in acc:
            res[k][c + "_KiB_per_dispatch"] = acc[k] / cnt[k]; res[k]["dispatches"] = cnt[k]
    return res
ntt = per_kernel("pmc_ntt")
json.dump(ntt, open(f"{out}/pmc_traffic.json", "w"), indent=1)
def moved(d):   # bytes: FETCH_SIZE doubled (gfx950 reports half of streamed read bytes, MI355X_MICROARCH.md) + WRITE_SIZE
    return (2 * d.get("FETCH_SIZE_KiB_per_dispatch", 0) + d.get("WRITE_SIZE_KiB_per_dispatch", 0)) * 1024
fwd = [k for k in ntt if ("strided_round" in k and "false, false, true" in k) or "tile_forward" in k]
inv = [k for k in ntt if ("strided_round" in k and "true, true, false" in k) or "tile_inverse" in k]
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/ntt_bench.py, n=2^16, 4096 polys, 512 polys per dispatch; FETCH_SIZE doubled per MI355X_MICROARCH.md",
           "forward_bytes_per_transform": sum(moved(ntt[k]) for k in fwd) / 512, "inverse_bytes_per_transform": sum(moved(ntt[k]) for k in inv) / 512,
           "algorithmic_bytes_per_transform": 1048576, "kernels_forward": fwd, "kernels_inverse": inv}, open(f"{out}/roofline_inputs.json", "w"), indent=1)
com = per_kernel("pmc_commit")
# commit_bench.py runs the e1-given pipeline (forward round, middle stage, inverse round + e1) and the e1-sampled one (the inverse
# round is then ntt_strided_round_sampled); the forward round and the middle stage are the same kernels in both
pipeline = [k for k in com if "mlwe_mid" in k or ("ntt_strided_round<" in k)]
per_commit = sum(moved(com[k]) for k in pipeline) / 64          # 64 witness vectors per dispatch (128 MiB chunks at rank 4)
sampled = [k for k in com if "mlwe_mid" in k or "ntt_strided_round_sampl" in k]     # ..._sampling (forward) + ..._sampled (inverse)
per_commit_sampled = sum(moved(com[k]) for k in sampled) / 64
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on J=256 tools/commit_bench.py, rank 4, n=2^16, 64 witness vectors per dispatch; FETCH_SIZE doubled",
           "bytes_per_commit": per_commit, "algorithmic_bytes_per_commit": 6291456,
           "bytes_per_commit_e1_sampled_in_pass": per_commit_sampled, "algorithmic_bytes_per_commit_e1_sampled": 4194304,
           "kernel": "ntt_strided_round<F64,4> fwd (r -> workspace) + mlwe_mid_fused8<4> + ntt_strided_round<F64,4> inv (+ e1), per 64-vector chunk",
           "per_kernel": {k: com[k] for k in com}}, open(f"{out}/pmc_commit_traffic.json", "w"), indent=1)
print("forward bytes/transform", sum(moved(ntt[k]) for k in fwd) / 512, "commit bytes", per_commit, per_commit / 2**20, "MiB", "sampled", per_commit_sampled / 2**20, "MiB")
PY
python3 -c "
import json; d=json.load(open('$out/bench.json')); e=d['extra']
print('value', d['value'], 'frac', d['roofline']['frac'], 'commits/s', e['commits_per_s'], e['commit_roofline_frac'], 'e1dev', e.get('commits_per_s_e1_on_device'), 'cfg4', e.get('config4'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['gpu_forward_matches_cpu'], d['cpu_baseline'].get('commit'))"
grep -E "mlwe_mid|strided|tile_forward|tile_inverse|gaussian" $out/bench_kernel_stats.csv | cut -d, -f1-4 | cut -c1-100
