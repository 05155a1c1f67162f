#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_split
rm -rf $out && mkdir -p $out
for sp in 88 412; do
  t=$(LAMBDA_SNARK_COMMIT_SPLIT=$sp timeout -k 10 300 python -m pytest tests/test_commitment_gpu.py -m gpu -x -q -k "config3 or fused_pipeline" 2>&1 | tail -1)
  echo "split=$sp tests: $t"
done
for rep in 1 2; do for sp in 88 412; do
  echo -n "split=$sp: "; LAMBDA_SNARK_COMMIT_SPLIT=$sp LAMBDA_SNARK_COMMIT_STREAMS=2 timeout -k 10 120 python3 tools/commit_bench.py 2>&1 | grep "e1 given"
done; done
for sp in 88 412; do
  J=256 LAMBDA_SNARK_COMMIT_SPLIT=$sp LAMBDA_SNARK_COMMIT_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats$sp -- python3 tools/commit_bench.py > $out/stats$sp.log 2>&1
  python3 -c "
import csv,glob
f=glob.glob('$out/stats$sp/*/*kernel_stats.csv')[0]
print('split=$sp', ' '.join(f\"{r['Name'].split('(')[0].split('::')[-1][:22]}={float(r['AverageNs'])/1e3:.1f}us\" for r in csv.DictReader(open(f)) if any(x in r['Name'] for x in ('mlwe_mid','strided','cols8'))))"
done
