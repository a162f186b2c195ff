#!/bin/bash
# rocprofv3 kernel stats of one cfg2 run -> gpurun_out/r03_${1}_kernel_stats_cfg2.csv (env passes through)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_x
(cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -- python3 bench.py --workload cfg2 --steps 2 --warmup 1 --no-cpu-baseline > /tmp/prof_x.log 2>&1) || { tail -5 /tmp/prof_x.log; exit 1; }
cp $(find /tmp/prof_x -name '*kernel_stats.csv' | head -1) $R/gpurun_out/r03_${1}_kernel_stats_cfg2.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/r03_${1}_kernel_stats_cfg2.csv")))
for r in rows[:24]:
    n=r['Name'].replace('(anonymous namespace)::','').replace('tdx::','')
    print(f"{float(r['TotalDurationNs'])/1e6:8.1f} ms {int(r['Calls']):5d} x {float(r['AverageNs'])/1e3:8.1f} us  {n[:110]}")
PY
