#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of every dispatch of tools/bin/ubench_fused2 (separate passes); last dispatch of each (kernel, grid) pair
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/fused2_$c; rm -rf $out
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $out -- tools/bin/ubench_fused2 > $out.log 2>&1
  f=$(ls $out/*/*counter_collection.csv | head -1)
  python3 - "$f" $c <<'PY'
import csv, sys, collections
last = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0][:60]
    if 'fused' not in k and 'pass' not in k: continue
    last[(k, r['Grid_Size'])] = float(r['Counter_Value'])
for (k, g), v in last.items():
    # rocprofv3 reports KiB; FETCH_SIZE counts half the bytes on gfx950 (MI355X_MICROARCH.md, HBM section): doubled here
    gib = v / 2**20 * (2 if sys.argv[2] == 'FETCH_SIZE' else 1)
    print(f"{sys.argv[2]} {k:60s} grid {g:>8s}  {gib:8.3f} GiB")
PY
done
