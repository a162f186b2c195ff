#!/bin/bash
# round-3 reference measurements at HEAD: hbm table, cfg5 at N = 1, cfg1, PMC passes
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_h
(cd $R && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h -- python3 bench.py --workload hbm --steps 3 --warmup 1 > /tmp/prof_h.log 2>&1) || { tail -5 /tmp/prof_h.log; exit 1; }
cp $(find /tmp/prof_h -name '*kernel_stats.csv' | head -1) $R/gpurun_out/r03_kernel_stats_hbm_workload.csv
cd $R
python3 tools/hbm_table.py gpurun_out/r03_kernel_stats_hbm_workload.csv gpurun_out/r03_hbm.json || exit 1
timeout -k 10 500 python bench.py --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_cfg5_n1.json 2> gpurun_out/r03_bench_cfg5_n1.err || { tail -5 gpurun_out/r03_bench_cfg5_n1.err; exit 1; }
timeout -k 10 300 python bench.py --workload cfg1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03_bench_cfg1.json 2>/dev/null || exit 1
timeout -k 10 600 bash tools/pmc_r03.sh > gpurun_out/r03_pmc.log 2>&1 || { tail -20 gpurun_out/r03_pmc.log; exit 1; }
python3 - <<'PY'
import json
for f in ("gpurun_out/r03_bench_cfg5_n1.json", "gpurun_out/r03_bench_cfg1.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["value"], 1), round(d["ms_per_step"], 1))
PY
