#!/bin/bash
# 8 + 8 commitment pipeline: 16- vs 32-column outer passes against the default 4 + 12 split (late round 2)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_cols
rm -rf $out && mkdir -p $out
for cols in 16 32; do
  t=$(LAMBDA_SNARK_COMMIT_SPLIT=88 LAMBDA_SNARK_COMMIT_COLS=$cols timeout -k 10 300 python -m pytest tests/test_commitment_gpu.py -m gpu -x -q -k "config3 or fused_pipeline or alternative" 2>&1 | tail -1)
  echo "split=88 cols=$cols tests: $t"
done
for rep in 1 2 3; do for cfg in "412 16" "88 16" "88 32"; do
  set -- $cfg
  for st in 2 1; do
    echo -n "split=$1 cols=$2 streams=$st: "; LAMBDA_SNARK_COMMIT_SPLIT=$1 LAMBDA_SNARK_COMMIT_COLS=$2 LAMBDA_SNARK_COMMIT_STREAMS=$st timeout -k 10 120 python3 tools/commit_bench.py 2>&1 | grep "e1 given"
  done
done; done
for cfg in "88 16" "88 32"; do
  set -- $cfg
  J=256 LAMBDA_SNARK_COMMIT_SPLIT=$1 LAMBDA_SNARK_COMMIT_COLS=$2 LAMBDA_SNARK_COMMIT_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats$1_$2 -- python3 tools/commit_bench.py > $out/stats$1_$2.log 2>&1
  python3 -c "
import csv,glob
f=glob.glob('$out/stats$1_$2/*/*kernel_stats.csv')[0]
print('split=$1 cols=$2', ' '.join(f\"{r['Name'].split('(')[0].split('::')[-1][:26]}={float(r['AverageNs'])/1e3:.1f}us\" for r in csv.DictReader(open(f)) if any(x in r['Name'] for x in ('mlwe_mid','strided','cols8'))))"
done
