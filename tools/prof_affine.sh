#!/bin/bash
# usage: tools/prof_affine.sh  (on the GPU box) — kernel-time sums for the XCD-affine / chunk experiment
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "0 256" "1 256" "0 24" "1 24" "1 16" "1 32" "1 48"; do
  set -- $cfg
  export LAMBDA_SNARK_NTT_XCD_AFFINE=$1 LAMBDA_SNARK_NTT_CHUNK_MIB=$2
  out=gpurun_out/prof_affine_$1_$2
  rm -rf $out
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/ntt_bench.py > $out.log 2>&1
  echo "== affine=$1 chunk=$2: $(grep -v amdgpu.ids $out.log | grep chunk= | tail -1)"
  f=$(ls $out/*/*kernel_stats.csv | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if 'ntt_' in r['Name']:
        name = r['Name'].split('<')[0].replace('void lsr::','') + ('<inv>' if ', true,' in r['Name'].split('(')[0] or 'inverse' in r['Name'] else '<fwd>')
        print(f"   {name:32s} calls {r['Calls']:>6s} total {float(r['TotalDurationNs'])/1e6:9.3f} ms avg {float(r['AverageNs'])/1e3:9.2f} us")
PY
done
