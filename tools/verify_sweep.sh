#!/bin/bash
# openings at n = 2^16: chunk lanes x chunk size (library variants built with -DLSR_VERIFY_STREAMS / -DLSR_VERIFY_CHUNK_MIB)
cd $GRAFT_REPO_ROOT
for k in 4 2; do
  for v in "" s2c128 s2c192 s2c256; do
    r=$(LIBVARIANT=$v N=65536 K=$k J=1024 GENERAL=0 REPS=10 timeout -k 10 120 python3 tools/commit_rows_bench.py 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['fused']; print('verify %.3f ms  %.0f K/s  frac %.3f | commit %.3f ms' % (f['verify_ms'], f['openings_per_s']/1e3, f['verify_roofline_frac'], f['commit_ms']))")
    echo "k=$k variant=${v:-product(s2c64)}: $r"
  done
done
