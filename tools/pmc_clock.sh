#!/bin/bash
# usage: pmc_clock.sh VAR=value ...: held clock and MFMA pipe utilisation per x3 kernel over one cfg2 step under the given environment
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_clk
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_clk -o p -- python3 bench.py --workload cfg2 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_clk.log 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/pmc_clk/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = acc[r["Kernel_Name"][:90]][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_clk/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): dur[r["Kernel_Name"][:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = []
for k, c in acc.items():
    if "GRBM_GUI_ACTIVE" not in c or not dur[k]: continue
    us = sum(dur[k]) / len(dur[k]); gui = c["GRBM_GUI_ACTIVE"][0] / c["GRBM_GUI_ACTIVE"][1]
    mf = c["SQ_VALU_MFMA_BUSY_CYCLES"][0] / c["SQ_VALU_MFMA_BUSY_CYCLES"][1]
    rows.append((us * len(dur[k]), us, gui / 8 / us / 1e3, mf / (1024 * gui / 8), k))
for tot, us, ghz, util, k in sorted(rows, reverse=True)[:8]:
    print(f"{us:9.1f} us  clock {ghz:5.3f} GHz  MFMA pipe {util:5.3f}  {k}")
PY
rm -rf gpurun_out/pmc_clk
