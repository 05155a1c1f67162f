#!/bin/bash
# mixed launches at ranks 1-3: block-order ratio fixed at 4 against the proportional default
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { echo -n "$1: "; shift; env "$@" timeout -k 10 120 python3 tools/commit_bench.py 2>&1 | grep "e1 given"; }
for rep in 1 2; do for k in 2 3 1; do
  run "K=$k three-launch schedule " K=$k J=2048 LAMBDA_SNARK_COMMIT_MIXED=0
  run "K=$k mixed ratio=4         " K=$k J=2048 LAMBDA_SNARK_COMMIT_MIX_RATIO=4
  run "K=$k mixed ratio=2         " K=$k J=2048 LAMBDA_SNARK_COMMIT_MIX_RATIO=2
  run "K=$k mixed default ratio   " K=$k J=2048
done; done
