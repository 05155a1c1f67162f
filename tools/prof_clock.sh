#!/bin/bash
# effective shader clock of the forward tile kernel (GRBM_GUI_ACTIVE / 8 XCDs / duration) under the ablation flags
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export N=4096 B=65536
for d in 0 1 2 3; do
  export LSR_DBG_TILE=$d
  rm -rf gpurun_out/clk_$d gpurun_out/clkt_$d
  timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/clk_$d -- python3 tools/ntt_bench.py > gpurun_out/clk_$d.log 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/clkt_$d -- python3 tools/ntt_bench.py > gpurun_out/clkt_$d.log 2>&1
  python3 - $d <<'PY'
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(f"gpurun_out/clk_{d}/*/*counter_collection.csv")[0]
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "ntt_tile_forward" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
g = glob.glob(f"gpurun_out/clkt_{d}/*/*kernel_stats.csv")[0]
dur = [float(r["AverageNs"]) for r in csv.DictReader(open(g)) if "ntt_tile_forward" in r["Name"]][0]
cyc = sum(vals) / len(vals) / 8
print(f"dbg={d}: GUI_ACTIVE/8 = {cyc/1e6:.3f} Mcycles, avg duration {dur/1e3:.1f} us (un-profiled pass) -> effective clock {cyc/dur:.2f} GHz")
PY
done
