#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_twolane
rm -rf $out && mkdir -p $out
t=$(LAMBDA_SNARK_COMMIT_SPLIT=88 LAMBDA_SNARK_COMMIT_TWO_LANE=1 LAMBDA_SNARK_COMMIT_MID_WAVES=4 timeout -k 10 300 python -m pytest tests/test_commitment_gpu.py -m gpu -x -q -k "config3 or fused_pipeline" 2>&1 | tail -1)
echo "two-lane tests: $t"
for rep in 1 2; do
for cfg in "1 4 128" "1 4 64" "1 8 128" "0 8 128"; do set -- $cfg
  echo -n "two_lane=$1 mid_waves=$2 chunk=$3: "; LAMBDA_SNARK_COMMIT_SPLIT=88 LAMBDA_SNARK_COMMIT_TWO_LANE=$1 LAMBDA_SNARK_COMMIT_MID_WAVES=$2 LAMBDA_SNARK_COMMIT_CHUNK_MIB=$3 timeout -k 10 120 python3 tools/commit_bench.py 2>&1 | grep -E "e1 given|on device" | tr '\n' ' '; echo
done; done
J=512 LAMBDA_SNARK_COMMIT_SPLIT=88 LAMBDA_SNARK_COMMIT_TWO_LANE=1 LAMBDA_SNARK_COMMIT_MID_WAVES=4 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 tools/commit_bench.py > $out/t.log 2>&1
python3 - $out/t <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if any(x in r["Kernel_Name"] for x in ("mlwe_mid", "cols8"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-72:]
dur = collections.defaultdict(list)
for r in rows:
    dur[r["Kernel_Name"].split("(")[0].split("::")[-1][:16]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print({k: round(sum(v) / len(v), 1) for k, v in dur.items()})
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[:24]:
    print("   ", r["Kernel_Name"].split("(")[0].split("::")[-1][:14], "queue", r.get("Queue_Id"), (int(r["Start_Timestamp"]) - t0) // 1000, "->", (int(r["End_Timestamp"]) - t0) // 1000)
PY
