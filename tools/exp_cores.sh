#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in 4 8; do
  t=$(LAMBDA_SNARK_COMMIT_MID_WAVES=$w timeout -k 10 300 python -m pytest tests/test_commitment_gpu.py -m gpu -x -q -k "config3 or fused_pipeline" 2>&1 | tail -1)
  echo "waves=$w tests: $t"
done
for w in 4 8; do for st in 1 2 3 4; do for mib in 64 128; do
  echo -n "split=88 waves=$w streams=$st chunk=$mib: "; LAMBDA_SNARK_COMMIT_MID_WAVES=$w LAMBDA_SNARK_COMMIT_STREAMS=$st LAMBDA_SNARK_COMMIT_CHUNK_MIB=$mib timeout -k 10 120 python3 tools/commit_bench.py 2>&1 | grep "e1 given"
done; done; done | tee gpurun_out/r02_coresidency_sweep.txt
