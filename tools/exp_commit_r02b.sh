#!/bin/bash
# Round-2 commit-pipeline experiment B: touch-prefetch on/off, SQ counters of the fused middle kernel.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_commit_b
rm -rf $out && mkdir -p $out
V=$GRAFT_REPO_ROOT/lambda-snark-r_amd/lib
timeout -k 10 600 python -m pytest tests/test_commitment_gpu.py -m gpu -x -q -k "config3" > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
{
for lib in "" _notouch; do for st in 1 2; do
  echo "== lib=core$lib streams=$st"; LAMBDA_SNARK_CORE_LIB=$V/liblambda_snark_core$lib.so LAMBDA_SNARK_COMMIT_STREAMS=$st timeout -k 10 120 python3 tools/commit_bench.py
done; done
} > $out/sweep.txt 2>&1
grep -E "==|e1 given" $out/sweep.txt
for lib in "" _notouch; do
  J=256 LAMBDA_SNARK_CORE_LIB=$V/liblambda_snark_core$lib.so LAMBDA_SNARK_COMMIT_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats$lib -- python3 tools/commit_bench.py > $out/stats$lib.log 2>&1
  cp $(ls $out/stats$lib/*/*kernel_stats.csv | head -1) $out/kernel_stats$lib.csv
  echo "-- core$lib"; grep -E "mlwe_mid|strided" $out/kernel_stats$lib.csv | cut -c1-60,200-400 | awk -F, '{print $(NF-6), $(NF-4)}'
done
rocprofv3 -L > $out/counters.txt 2>&1
grep -oE "SQ_[A-Z_0-9]+" $out/counters.txt | sort -u | tr '\n' ' ' | head -c 3000; echo
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"
P3="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
n=1
for P in "$P1" "$P2" "$P3"; do
  J=128 LAMBDA_SNARK_COMMIT_STREAMS=1 timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $out/pmc$n -- python3 tools/commit_bench.py > $out/pmc$n.log 2>&1
  n=$((n+1))
done
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for n in (1, 2, 3):
    fs = glob.glob(f"{out}/pmc{n}/*/*counter_collection.csv")
    if not fs:
        print("pmc", n, "no output"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void lsr::", "")[:40]
        if "mlwe_mid" not in k and "strided" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    for k in acc:
        print(k, {c: round(v / cnt[k][c]) for c, v in acc[k].items()})
PY
