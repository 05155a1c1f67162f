#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for cfg in "412 8 0" "88 8 0" "88 4 0"; do set -- $cfg
    echo -n "split=$1 mid_waves=$2 two_lane=$3: "; LAMBDA_SNARK_COMMIT_SPLIT=$1 LAMBDA_SNARK_COMMIT_MID_WAVES=$2 LAMBDA_SNARK_COMMIT_TWO_LANE=$3 LAMBDA_SNARK_COMMIT_STREAMS=2 timeout -k 10 120 python3 tools/commit_bench.py 2>&1 | grep -E "e1 given"
  done
done | tee gpurun_out/r02_final_ab.txt
