#!/bin/bash
# PMC passes for the tile kernel (n=4096 => tile kernel only).  Counters in separate runs (SQ has 8 slots).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export N=${N:-4096} B=${B:-65536}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE"; do
  out=gpurun_out/pmc_$i
  rm -rf $out
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $out -- python3 tools/ntt_bench.py > $out.log 2>&1
  f=$(ls $out/*/*counter_collection.csv | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0].replace('void lsr::','')[:70]
    if 'ntt_' not in k: continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k, d in acc.items():
    print(k)
    for c, v in d.items(): print(f"    {c:24s} per-dispatch {v / cnt[(k, c)]:16.1f}   (dispatches {cnt[(k,c)]})")
PY
  i=$((i+1))
done
