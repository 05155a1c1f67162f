#!/bin/bash
# Two-lane commitment pipeline with CU-masked lanes: sweep of the outer lane's CUs per XCD, both splits (round 2).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_cumask_lanes.txt
: > $out
t=$(LAMBDA_SNARK_COMMIT_TWO_LANE=1 LAMBDA_SNARK_COMMIT_OUTER_CUS=12 timeout -k 10 300 python -m pytest tests/test_commitment_gpu.py -m gpu -x -q -k "config3 or fused_pipeline" 2>&1 | tail -1)
echo "two-lane (masked, 12 outer CUs/XCD) tests, split 412: $t" | tee -a $out
t=$(LAMBDA_SNARK_COMMIT_SPLIT=88 LAMBDA_SNARK_COMMIT_TWO_LANE=1 LAMBDA_SNARK_COMMIT_OUTER_CUS=12 timeout -k 10 300 python -m pytest tests/test_commitment_gpu.py -m gpu -x -q -k "config3 or fused_pipeline" 2>&1 | tail -1)
echo "two-lane (masked, 12 outer CUs/XCD) tests, split 88: $t" | tee -a $out
run() { echo -n "split=$1 two_lane=$2 outer_cus=$3 chunk=$4: " | tee -a $out
  LAMBDA_SNARK_COMMIT_SPLIT=$1 LAMBDA_SNARK_COMMIT_TWO_LANE=$2 LAMBDA_SNARK_COMMIT_OUTER_CUS=$3 LAMBDA_SNARK_COMMIT_CHUNK_MIB=$4 timeout -k 10 120 python3 tools/commit_bench.py 2>&1 | grep -E "e1 given|on device" | sed -e 's/matvec commit //' | tr '\n' ' ' | tee -a $out; echo | tee -a $out; }
run 412 0 0 128
run 88 0 0 128
for o in 8 10 12 14 16; do run 412 1 $o 128; done
for o in 8 10 12 14 16; do run 88 1 $o 128; done
run 412 1 12 64
run 412 1 12 32
run 88 1 12 64
run 412 0 0 128
