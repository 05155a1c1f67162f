#!/bin/bash
# A/B of the product against variant builds on the same box: tools/exp_ab.sh <variant> [<variant> ...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/lambda-snark-r_amd/lib
run() { echo -n "$1: "; shift; env "$@" timeout -k 10 120 python3 tools/commit_bench.py 2>&1 | grep "e1 given"; }
for rep in 1 2 3; do
  run "product      " X=1
  for v in "$@"; do run "variant $v" LAMBDA_SNARK_CORE_LIB=$V/liblambda_snark_core_$v.so; done
done
