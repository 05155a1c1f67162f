#!/bin/bash
# Ablation of the fused middle kernel: which of its memory streams keeps the FP64 pipe waiting?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/lambda-snark-r_amd/lib
out=gpurun_out/r02_ablate
rm -rf $out && mkdir -p $out
for lib in "" _abl1 _abl2 _abl3 _abl7; do
  J=256 LAMBDA_SNARK_CORE_LIB=$V/liblambda_snark_core$lib.so LAMBDA_SNARK_COMMIT_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats$lib -- python3 tools/commit_bench.py > $out/stats$lib.log 2>&1
done
python3 - $out <<'PY' | tee $out/summary.txt
import csv, glob, sys
out = sys.argv[1]
names = {"": "full kernel", "_abl1": "no operand loads", "_abl2": "no matrix loads", "_abl3": "no operand + no matrix loads", "_abl7": "no loads, no stores"}
for lib, what in names.items():
    st = glob.glob(f"{out}/stats{lib}/*/*kernel_stats.csv")[0]
    dur = [r["AverageNs"] for r in csv.DictReader(open(st)) if "mlwe_mid" in r["Name"]]
    print(f"mlwe_mid_fused8<4>, {what:32s}: {float(dur[0])/1e3:7.1f} us per 64 witness vectors")
PY
