#!/bin/bash
# quotient pipeline with the elementwise stages fused into the transforms' read-in (LAMBDA_SNARK_QUOTIENT_FUSE, default 1) vs apart
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_prover_gpu.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do for mb in "4096 4096" "1024 16384" "64 262144" "8192 2048"; do
  set -- $mb
  for f in 0 1; do echo -n "m=$1 B=$2 fuse=$f: "; M=$1 B=$2 LAMBDA_SNARK_QUOTIENT_FUSE=$f timeout -k 10 120 python3 tools/quotient_bench.py 2>&1 | tail -1; done
done; done
