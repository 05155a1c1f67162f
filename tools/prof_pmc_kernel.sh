#!/bin/bash
# PMC passes (separate runs per counter set) over one tool run, per-dispatch averages for kernels matching a pattern:
#   tools/prof_pmc_kernel.sh <name> <kernel-regex> <script> [ENV=VALUE ...]   -> gpurun_out/r03/<name>_pmc.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
name=$1; pat=$2; script=$3; shift 3
for kv in "$@"; do export "$kv"; done
mkdir -p gpurun_out/r03
report=gpurun_out/r03/${name}_pmc.txt
: > $report
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE"; do
  out=gpurun_out/r03/pmc_tmp_$i
  rm -rf $out
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $out -- python3 $script > $out.log 2>&1
  f=$(ls $out/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -z "$f" ]; then echo "no counters for: $set" >> $report; tail -3 $out.log >> $report; else
  python3 - "$f" "$pat" >> $report <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0].replace('void lsr::','')[:80]
    if not re.search(sys.argv[2], k): continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k, d in acc.items():
    for c, v in d.items(): print(f"{k:60s} {c:24s} per-dispatch {v / cnt[(k, c)]:18.1f}   (dispatches {cnt[(k,c)]})")
PY
  fi
  rm -rf $out $out.log
  i=$((i+1))
done
cat $report
