#!/bin/bash
# Round-2 commit-pipeline experiment: correctness of the fused path, then stream-count / chunk sweep, then kernel stats.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_commit
rm -rf $out && mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_commitment_gpu.py -m gpu -x -q -k "config3" > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -3 $out/tests.log
{
echo "== unfused (round-1 pipeline)"; LAMBDA_SNARK_COMMIT_FUSED=0 timeout -k 10 120 python3 tools/commit_bench.py
for st in 1 2 3 4; do for mib in 64 128 256; do
  echo "== fused streams=$st chunk=${mib}MiB"; LAMBDA_SNARK_COMMIT_STREAMS=$st LAMBDA_SNARK_COMMIT_CHUNK_MIB=$mib timeout -k 10 120 python3 tools/commit_bench.py
done; done
} > $out/sweep.txt 2>&1
grep -E "==|e1 given" $out/sweep.txt
LAMBDA_SNARK_COMMIT_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1 -- python3 tools/commit_bench.py > $out/stats1.log 2>&1
cp $(ls $out/stats1/*/*kernel_stats.csv | head -1) $out/kernel_stats_streams1.csv
head -12 $out/kernel_stats_streams1.csv
