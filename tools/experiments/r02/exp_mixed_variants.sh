#!/bin/bash
# build variants of the middle stage under the mixed-launch schedule
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/lambda-snark-r_amd/lib
run() { echo -n "$1: "; shift; env "$@" timeout -k 10 120 python3 tools/commit_bench.py 2>&1 | grep "e1 given"; }
for rep in 1 2 3; do
  run "product              " X=1
  run "LDS exchange (xchg0) " LAMBDA_SNARK_CORE_LIB=$V/liblambda_snark_core_xchg0.so
  run "no touch prefetch    " LAMBDA_SNARK_CORE_LIB=$V/liblambda_snark_core_touch0.so
  run "default cache policy " LAMBDA_SNARK_CORE_LIB=$V/liblambda_snark_core_noahead.so
done
