#!/bin/bash
# usage: tools/exp_variants.sh name1 name2 ...   ("" = product library): config-3 parity, commit_bench, kernel time of each variant build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/lambda-snark-r_amd/lib
out=gpurun_out/r02_variants
rm -rf $out && mkdir -p $out
for name in core "$@"; do
  lib=$V/liblambda_snark_core_$name.so; [ $name = core ] && lib=$V/liblambda_snark_core.so
  t=$(LAMBDA_SNARK_CORE_LIB=$lib timeout -k 10 300 python -m pytest tests/test_commitment_gpu.py -m gpu -x -q -k "config3" 2>&1 | tail -1)
  b=$(LAMBDA_SNARK_CORE_LIB=$lib LAMBDA_SNARK_COMMIT_STREAMS=2 timeout -k 10 120 python3 tools/commit_bench.py 2>&1 | grep "e1 given")
  J=256 LAMBDA_SNARK_CORE_LIB=$lib LAMBDA_SNARK_COMMIT_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$name -- python3 tools/commit_bench.py > $out/stats_$name.log 2>&1
  k=$(python3 -c "
import csv,glob
f=glob.glob('$out/stats_$name/*/*kernel_stats.csv')[0]
print(' '.join(f\"{r['Name'].split('<')[0].split('::')[-1][:18]}={float(r['AverageNs'])/1e3:.1f}us\" for r in csv.DictReader(open(f)) if 'mlwe_mid' in r['Name'] or 'strided' in r['Name']))")
  echo "$name | $t | $b | $k"
done | tee $out/summary.txt
