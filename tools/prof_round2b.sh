#!/bin/bash
# Late round-2 profile artefacts with the mixed-launch commitment schedule as default (copied to profiles/r02b_* afterwards)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02b
rm -rf $out && mkdir -p $out
python3 bench.py > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu > $out/bench_under_rocprof.json 2> $out/stats.log
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/bench_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  J=256 timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $out/pmc_commit_$c -- python3 tools/commit_bench.py > $out/pmc_commit_$c.log 2>&1
done
python3 - $out <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
tot = {}
per = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/pmc_commit_{c}/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "lsr" not in r["Kernel_Name"] or r["Counter_Name"] != c: continue
        k = r["Kernel_Name"].split("(")[0].replace("void lsr::", "")
        acc[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in acc:
        per[k][c + "_KiB_total"] = acc[k]; per[k]["dispatches"] = cnt[k]
# commit_bench.py (J = 256): 2 warm-up + 10 timed calls of the e1-given pipeline = 12 x 256 witness vectors through mlwe_mixed
mixed = [k for k in per if "mlwe_mixed" in k]
calls = 12
moved = sum((2 * per[k].get("FETCH_SIZE_KiB_total", 0) + per[k].get("WRITE_SIZE_KiB_total", 0)) * 1024 for k in mixed)
per_commit = moved / (calls * 256)
old = json.load(open("profiles/r02_pmc_commit_traffic.json"))
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on J=256 tools/commit_bench.py, rank 4, n=2^16; all mlwe_mixed dispatches of the 12 e1-given calls (12 x 256 witness vectors); FETCH_SIZE doubled",
           "bytes_per_commit": per_commit, "algorithmic_bytes_per_commit": 6291456,
           "bytes_per_commit_e1_sampled_in_pass": old.get("bytes_per_commit_e1_sampled_in_pass"), "algorithmic_bytes_per_commit_e1_sampled": 4194304,
           "kernel": "mlwe_mixed<4>: one launch = middle stage of chunk t (12 fwd stages x 4, A^T product, 12 inv stages x 4) + forward strided round of chunk t+1 + inverse strided round (+ e1) of chunk t-1; 32-vector chunks, two lanes",
           "per_kernel": {k: per[k] for k in per}}, open(f"{out}/pmc_commit_traffic.json", "w"), indent=1)
print("commit bytes", per_commit, per_commit / 2**20, "MiB")
PY
python3 -c "
import json; d=json.load(open('$out/bench.json')); e=d['extra']
print('value', d['value'], 'frac', d['roofline']['frac'], 'commits/s', e['commits_per_s'], e['commit_roofline_frac'], 'e1dev', e.get('commits_per_s_e1_on_device'), 'cfg4', e.get('config4'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['gpu_forward_matches_cpu'], d['cpu_baseline'].get('commit'))"
grep -E "mlwe_|strided|tile_forward|tile_inverse|gaussian" $out/bench_kernel_stats.csv | cut -d, -f1-4 | cut -c1-100
