#!/bin/bash
# A/B of the quotient pipeline: library variant "base" (previous sources) against the product library, kernel stats of both
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/q3; mkdir -p $out
for m in 4096 1024 64; do
  b=$((16777216 / m))
  for v in base ""; do
    echo "M=$m variant=${v:-product}: $(M=$m B=$b LIBVARIANT=$v timeout -k 10 120 python3 tools/quotient_bench.py 2>/dev/null | tail -1)"
  done
done
export M=4096 B=4096
for v in base product; do
  if [ $v = base ]; then export LIBVARIANT=base; else unset LIBVARIANT; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/st_$v -- python3 tools/quotient_bench.py > $out/st_$v.log 2>&1
  echo "== $v"; cut -d, -f1-4 $(ls $out/st_$v/*/*kernel_stats.csv | head -1) | grep -v "at::\|elementwise" | cut -c1-150 | head -12
done
