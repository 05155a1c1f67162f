#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_split_pmc
rm -rf $out && mkdir -p $out
P1="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"
for sp in 88 412; do n=1; for P in "$P1" "$P2"; do
  J=128 LAMBDA_SNARK_COMMIT_SPLIT=$sp LAMBDA_SNARK_COMMIT_STREAMS=1 timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $out/pmc${sp}_$n -- python3 tools/commit_bench.py > $out/pmc${sp}_$n.log 2>&1
  n=$((n+1)); done; done
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sp in (88, 412):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
    for f in glob.glob(f"{out}/pmc{sp}_*/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void lsr::", "")[:28]
            if not any(x in k for x in ("mlwe_mid", "cols8", "strided")): continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    for k in acc:
        print(sp, k, {c.replace("SQ_", ""): round(v / cnt[k][c] / 1e6, 2) for c, v in sorted(acc[k].items())})
PY
