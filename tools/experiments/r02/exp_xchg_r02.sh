#!/bin/bash
# Round-2 experiment: the [3,6) <-> [0,3) exchange of the fused commitment kernel through LDS (default) or inside the wavefront
# (__shfl_xor / explicit DPP).  Bit-exactness first, then time and SQ instruction counts.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/lambda-snark-r_amd/lib
out=gpurun_out/r02_xchg
rm -rf $out && mkdir -p $out
for lib in "" _xchg1 _xchg2; do
  LAMBDA_SNARK_CORE_LIB=$V/liblambda_snark_core$lib.so timeout -k 10 300 python -m pytest tests/test_commitment_gpu.py -m gpu -x -q -k "config3" > $out/tests$lib.log 2>&1
  echo "core$lib tests: $(tail -1 $out/tests$lib.log)"
done
for rep in 1 2; do for lib in "" _xchg1 _xchg2; do
  echo -n "core$lib streams=2: "; LAMBDA_SNARK_CORE_LIB=$V/liblambda_snark_core$lib.so LAMBDA_SNARK_COMMIT_STREAMS=2 timeout -k 10 120 python3 tools/commit_bench.py 2>&1 | grep "e1 given"
done; done | tee $out/times.txt
P="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE"
for lib in "" _xchg1 _xchg2; do
  J=128 LAMBDA_SNARK_CORE_LIB=$V/liblambda_snark_core$lib.so LAMBDA_SNARK_COMMIT_STREAMS=1 timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $out/pmc$lib -- python3 tools/commit_bench.py > $out/pmc$lib.log 2>&1
  J=256 LAMBDA_SNARK_CORE_LIB=$V/liblambda_snark_core$lib.so LAMBDA_SNARK_COMMIT_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats$lib -- python3 tools/commit_bench.py > $out/stats$lib.log 2>&1
done
python3 - $out <<'PY' | tee $out/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
for lib in ("", "_xchg1", "_xchg2"):
    fs = glob.glob(f"{out}/pmc{lib}/*/*counter_collection.csv")
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if "mlwe_mid" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
    st = glob.glob(f"{out}/stats{lib}/*/*kernel_stats.csv")[0]
    dur = [r["AverageNs"] for r in csv.DictReader(open(st)) if "mlwe_mid" in r["Name"]]
    print(f"core{lib}: mlwe_mid_fused8<4> avg {float(dur[0])/1e3:.1f} us per 64 vectors;", {c: round(v / cnt[c]) for c, v in sorted(acc.items())})
PY
