#!/bin/bash
# SQ / GRBM counters of the mixed-role commitment kernel (one lane, 64-vector chunks, so that a dispatch = one full mixed launch)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_mixed_pmc
rm -rf $out && mkdir -p $out
P1="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
P2="SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"
P3="GRBM_GUI_ACTIVE GRBM_COUNT"
n=1
for P in "$P1" "$P2" "$P3"; do
  J=256 LAMBDA_SNARK_COMMIT_MIX_LANES=1 LAMBDA_SNARK_COMMIT_MIX_CHUNK_MIB=128 timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $out/pmc_$n -- python3 tools/commit_bench.py > $out/pmc_$n.log 2>&1
  n=$((n+1))
done
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
# full launches only: grid = (128 + 256/2... ) identify by the largest Grid_Size among mlwe_mixed dispatches
rows = []
for f in glob.glob(f"{out}/pmc_*/*/*counter_collection.csv"):
    rows += [r for r in csv.DictReader(open(f)) if "mlwe_mixed" in r["Kernel_Name"]]
big = max(int(r["Grid_Size"]) for r in rows)
acc = collections.defaultdict(float); cnt = collections.Counter()
for r in rows:
    if int(r["Grid_Size"]) != big: continue
    acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
print(f"mlwe_mixed<4>, full launches (grid {big} work-items = {big // 512} workgroups: 1024 middle + 1024 forward (2 groups) + 2048 inverse), per dispatch:")
for c in sorted(acc): print(f"  {c:24s} {acc[c] / cnt[c]:16.1f}   ({cnt[c]} dispatches)")
PY
