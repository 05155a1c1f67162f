#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_commit_c
rm -rf $out && mkdir -p $out
{
for pad in 0 16384; do for st in 1 2 3 4; do for mib in 64 128; do
  echo "== pad=$pad streams=$st chunk=$mib"; LAMBDA_SNARK_COMMIT_MID_LDS_PAD=$pad LAMBDA_SNARK_COMMIT_STREAMS=$st LAMBDA_SNARK_COMMIT_CHUNK_MIB=$mib timeout -k 10 120 python3 tools/commit_bench.py 2>&1 | grep "e1 given"
done; done; done
} > $out/sweep.txt 2>&1
cat $out/sweep.txt
J=256 LAMBDA_SNARK_COMMIT_MID_LDS_PAD=16384 LAMBDA_SNARK_COMMIT_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 tools/commit_bench.py > $out/stats.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/stats/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "lsr" in r["Name"]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
f = glob.glob(sys.argv[1] + "/stats/*/*kernel_trace.csv")[0]
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:50]
    if k in seen or "lsr" not in k: continue
    seen.add(k); print(k, {x: r[x] for x in r if "GPR" in x or "LDS" in x or "Scratch" in x})
PY
