#!/bin/bash
# HBM-side traffic of the commit pipeline per kernel (FETCH_SIZE / WRITE_SIZE in separate passes, L2 hit/miss in a third).
# usage: tools/exp_pmc_commit.sh [variant-name]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/lambda-snark-r_amd/lib
name=${1:-core}
lib=$V/liblambda_snark_core_$name.so; [ $name = core ] && lib=$V/liblambda_snark_core.so
out=gpurun_out/r02_pmc_commit_$name
rm -rf $out && mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  J=256 LAMBDA_SNARK_CORE_LIB=$lib LAMBDA_SNARK_COMMIT_STREAMS=1 timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$tag -- python3 tools/commit_bench.py > $out/pmc_$tag.log 2>&1
done
python3 - $out <<'PY' | tee $out/summary.txt
import csv, glob, sys, collections, json
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob(f"{out}/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void lsr::", "")
        if "mlwe_mid" not in k and "strided" not in k and "gaussian" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
res = {}
for k in acc:
    per = {c: v / cnt[k][c] for c, v in acc[k].items()}
    res[k] = {"dispatches": max(cnt[k].values()), **{c: round(v, 1) for c, v in per.items()}}
    print(k[:70], res[k])
json.dump(res, open(f"{out}/per_kernel.json", "w"), indent=1)
PY
