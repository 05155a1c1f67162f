#!/bin/bash
# kernel trace of the CU-masked two-lane pipeline: per-kernel durations and the timeline of a few chunks
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "412 12" "412 16" "88 16"; do set -- $cfg
out=gpurun_out/r02_cumask_trace_$1_$2
rm -rf $out && mkdir -p $out
export J=512 LAMBDA_SNARK_COMMIT_SPLIT=$1 LAMBDA_SNARK_COMMIT_TWO_LANE=1 LAMBDA_SNARK_COMMIT_OUTER_CUS=$2
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 tools/commit_bench.py > $out/t.log 2>&1
echo "== split $1 outer_cus $2"
python3 - $out/t <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if any(x in r["Kernel_Name"] for x in ("mlwe_mid", "cols8", "ntt_strided_round"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[24 * 12: 24 * 15]          # e1-given repetitions (8 chunks x 3 kernels each), away from the warm-up
dur = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    key = "mid" if "mlwe_mid" in n else ("inv" if ("true, true" in n or "cols8_inverse" in n) else "fwd")
    dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print({k: round(sum(v) / len(v), 1) for k, v in dur.items()})
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[:24]:
    n = r["Kernel_Name"]
    key = "mid" if "mlwe_mid" in n else ("inv" if ("true, true" in n or "cols8_inverse" in n) else "fwd")
    print("   ", key, "queue", r.get("Queue_Id"), (int(r["Start_Timestamp"]) - t0) // 1000, "->", (int(r["End_Timestamp"]) - t0) // 1000)
PY
done
