#!/bin/bash
# mixed launches with the roles of the 8 + 8 split (mlwe_mixed88) against those of the 4 + 12 split (mlwe_mixed)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_commitment_gpu.py -m gpu -x -q -k "mixed_launch" 2>&1 | tail -3
run() { echo -n "$1: "; shift; env "$@" timeout -k 10 120 python3 tools/commit_bench.py 2>&1 | grep "e1 given"; }
for rep in 1 2 3; do
  run "mixed 4+12 (default)        " X=1
  run "mixed 8+8  ratio=4 lanes=2  " LAMBDA_SNARK_COMMIT_MIX_SPLIT=88
  run "mixed 8+8  ratio=2 lanes=2  " LAMBDA_SNARK_COMMIT_MIX_SPLIT=88 LAMBDA_SNARK_COMMIT_MIX_RATIO=2
  run "mixed 8+8  ratio=6 lanes=2  " LAMBDA_SNARK_COMMIT_MIX_SPLIT=88 LAMBDA_SNARK_COMMIT_MIX_RATIO=6
  run "mixed 8+8  ratio=4 lanes=1  " LAMBDA_SNARK_COMMIT_MIX_SPLIT=88 LAMBDA_SNARK_COMMIT_MIX_LANES=1 LAMBDA_SNARK_COMMIT_MIX_CHUNK_MIB=128
done
