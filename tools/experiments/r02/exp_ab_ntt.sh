#!/bin/bash
# A/B of the product against variant builds on the same box, transform batches: tools/exp_ab_ntt.sh <variant> [<variant> ...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/lambda-snark-r_amd/lib
run() { echo -n "$1: "; shift; env "$@" timeout -k 10 120 python3 tools/ntt_bench.py 2>&1 | tail -1; }
for rep in 1 2 3; do
  run "product      " X=1
  for v in "$@"; do run "variant $v" LAMBDA_SNARK_CORE_LIB=$V/liblambda_snark_core_$v.so; done
  run "product  n=4096" N=4096 B=65536
  for v in "$@"; do run "variant $v n=4096" N=4096 B=65536 LAMBDA_SNARK_CORE_LIB=$V/liblambda_snark_core_$v.so; done
done
