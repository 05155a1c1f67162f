#!/bin/bash
# rocprofv3 kernel statistics of one tool run: tools/prof_stats.sh <name> <script> [env...]; summary -> gpurun_out/r03/<name>_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
name=$1; script=$2; shift 2
for kv in "$@"; do export "$kv"; done
out=gpurun_out/r03/$name
rm -rf $out && mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $script > $out.log 2>&1
f=$(ls $out/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -z "$f" ]; then tail -20 $out.log; exit 1; fi
cp $f gpurun_out/r03/${name}_kernel_stats.csv
tail -4 $out.log
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:10.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} {r['Percentage']}%")
PY
rm -rf $out
