#!/bin/bash
# A/B: streaming (nt) loads of the once-read commit inputs (r out of place, e1) in the strided rounds — product vs -DLSR_NT_COMMIT_INPUTS=0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_nt_commit_inputs.txt; : > $out
for rep in 1 2 3; do
  for v in "" _ntin0; do
    lib=$PWD/lambda-snark-r_amd/lib/liblambda_snark_core$v.so
    echo -n "lib${v:-_product}: " | tee -a $out
    LAMBDA_SNARK_CORE_LIB=$lib timeout -k 10 120 python3 tools/commit_bench.py 2>&1 | grep -E "e1 given|on device" | sed -e 's/matvec commit //' | tr '\n' ' ' | tee -a $out; echo | tee -a $out
  done
done
for v in "" _ntin0; do
  lib=$PWD/lambda-snark-r_amd/lib/liblambda_snark_core$v.so
  echo -n "lib${v:-_product} ntt_bench: " | tee -a $out
  LAMBDA_SNARK_CORE_LIB=$lib timeout -k 10 120 python3 tools/ntt_bench.py 2>&1 | tail -3 | tr '\n' ' ' | tee -a $out; echo | tee -a $out
done
