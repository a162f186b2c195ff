#!/bin/bash
# instruction-cache requests / misses per kernel over one config-2 step (is straight-line epilogue code streaming through the I-cache?)
set -e
export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d gpurun_out/pmc_ic -o p -- python3 bench.py --workload cfg2 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_ic.log 2>&1 || { tail -5 gpurun_out/pmc_ic.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("gpurun_out/pmc_ic/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:110]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQC_ICACHE_REQ": n[k] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQC_ICACHE_MISSES", 0))[:14]
for k, c in rows:
    req, mis = c.get("SQC_ICACHE_REQ", 0), c.get("SQC_ICACHE_MISSES", 0)
    print(f"{mis / max(n[k],1):12.0f} misses/launch  {100 * mis / max(req, 1):6.2f} % of {req / max(n[k],1):12.0f} req  {k}")
PY
rm -rf gpurun_out/pmc_ic
