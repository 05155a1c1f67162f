#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/lambda-snark-r_amd/lib
out=gpurun_out/r02_nt.txt
: > $out
timeout -k 10 600 python -m pytest tests/test_ntt_gpu.py -m gpu -x -q > gpurun_out/r02_nt_tests.log 2>&1 || { tail -30 gpurun_out/r02_nt_tests.log; exit 1; }
tail -2 gpurun_out/r02_nt_tests.log
for lib in "" _nont; do
  for nb in "65536 4096" "4096 65536" "256 1048576" "8192 32768" "131072 2048"; do set -- $nb
    echo -n "core$lib " >> $out; LAMBDA_SNARK_CORE_LIB=$V/liblambda_snark_core$lib.so N=$1 B=$2 timeout -k 10 120 python3 tools/ntt_bench.py >> $out 2>&1
  done
  echo -n "core$lib commit: " >> $out; LAMBDA_SNARK_CORE_LIB=$V/liblambda_snark_core$lib.so LAMBDA_SNARK_COMMIT_STREAMS=2 timeout -k 10 120 python3 tools/commit_bench.py 2>&1 | grep "e1 given" >> $out
done
cat $out
