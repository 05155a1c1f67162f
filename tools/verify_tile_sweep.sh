#!/bin/bash
# openings at n = 4096 (one workgroup per opening): register / LDS trade of the last round's multipliers (library variants
# -DLSR_VERIFY_TILE_WAVES / -DLSR_VERIFY_TILE_TW_REGS)
cd $GRAFT_REPO_ROOT
for k in 2 4 1; do
  for v in w4r0 w4r1 ""; do
    r=$(LIBVARIANT=$v N=4096 K=$k J=16384 GENERAL=0 REPS=20 timeout -k 10 120 python3 tools/commit_rows_bench.py 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['fused']; print('verify %.3f ms  %.2f M/s  frac %.3f | commit %.3f ms' % (f['verify_ms'], f['openings_per_s']/1e6, f['verify_roofline_frac'], f['commit_ms']))")
    echo "k=$k variant=${v:-product(w6r1)}: $r"
  done
done
