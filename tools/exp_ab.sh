#!/bin/bash
# A/B of the product against a variant build on the same box: tools/exp_ab.sh <variant> [test-filter]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/lambda-snark-r_amd/lib
run() { echo -n "$1: "; shift; env "$@" timeout -k 10 120 python3 tools/commit_bench.py 2>&1 | grep "e1 given"; }
timeout -k 10 600 python -m pytest tests/test_commitment_gpu.py tests/test_ntt_gpu.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2 3 4; do
  run "product      " X=1
  run "variant $1" LAMBDA_SNARK_CORE_LIB=$V/liblambda_snark_core_$1.so
done
