#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_trace
rm -rf $out && mkdir -p $out
for cfg in "4 3" "4 1" "8 2"; do set -- $cfg
  J=256 LAMBDA_SNARK_COMMIT_MID_WAVES=$1 LAMBDA_SNARK_COMMIT_STREAMS=$2 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/t_$1_$2 -- python3 tools/commit_bench.py > $out/t_$1_$2.log 2>&1
  python3 - $out/t_$1_$2 $1 $2 <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if any(x in r["Kernel_Name"] for x in ("mlwe_mid", "cols8"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# take the last 60 launches (steady state)
rows = rows[-96:]
dur = collections.defaultdict(list)
for r in rows:
    dur[r["Kernel_Name"].split("(")[0].split("::")[-1][:16]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
span = (max(int(r["End_Timestamp"]) for r in rows) - min(int(r["Start_Timestamp"]) for r in rows)) / 1e3
total = sum(sum(v) for v in dur.values())
print(f"waves={sys.argv[2]} streams={sys.argv[3]}: span {span:.0f} us for {len(rows)} launches, sum of durations {total:.0f} us (overlap factor {total/span:.2f});",
      {k: round(sum(v) / len(v), 1) for k, v in dur.items()})
# show a short timeline
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[:12]:
    print("   ", r["Kernel_Name"].split("(")[0].split("::")[-1][:14], "queue", r.get("Queue_Id"), (int(r["Start_Timestamp"]) - t0) // 1000, "->", (int(r["End_Timestamp"]) - t0) // 1000)
PY
done
