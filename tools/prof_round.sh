#!/bin/bash
# Produces the per-round profile artefacts on the GPU box: kernel-trace stats of the exact bench command and
# FETCH_SIZE / WRITE_SIZE (separate --pmc passes) of the n = 2^16 NTT kernels.  Usage: tools/prof_round.sh r01
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag
rm -rf $out && mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu > $out/bench_under_rocprof.json 2> $out/stats.log
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 tools/ntt_bench.py > $out/pmc_$c.log 2>&1
  J=256 timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $out/pmc_commit_$c -- python3 tools/commit_bench.py > $out/pmc_commit_$c.log 2>&1
done
for mm in 4096 65536; do
  M=$mm B=$((16777216/mm)) timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/quot_$mm -- python3 tools/quotient_bench.py > $out/quotient_m$mm.log 2>&1
  cp $(ls $out/quot_$mm/*/*kernel_stats.csv | head -1) $out/quotient_m${mm}_kernel_stats.csv
done
for mm in 2 64 1024 4096 16384 65536 131072; do M=$mm B=$((16777216/mm)) timeout -k 10 120 python3 tools/quotient_bench.py >> $out/quotient_sweep.txt 2>&1; done
python3 - $out <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/pmc_{c}/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void lsr::", "")
        if "ntt_" not in k or r["Counter_Name"] != c: continue
        acc[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in acc: res[k][c + "_KiB_per_dispatch"] = acc[k] / cnt[k]; res[k]["dispatches"] = cnt[k]
json.dump(res, open(f"{out}/pmc_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
# commit workload (J = 256 witness vectors per dispatch sequence): bytes per kernel, FETCH_SIZE already doubled
com = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/pmc_commit_{c}/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void lsr::", "")
        if "lsr" not in r["Kernel_Name"] or r["Counter_Name"] != c: continue
        acc[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in acc: com[k][c + "_KiB_total"] = acc[k]; com[k]["dispatches"] = cnt[k]
json.dump(com, open(f"{out}/pmc_commit_traffic.json", "w"), indent=1)
PY
