#!/bin/bash
# GPU tests, then the default bench, cfg2 bench and the cfg2 / default kernel stats -> gpurun_out/r03_<tag>_*
TAG=${1:-a}
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03_${TAG}_tests.log 2>&1 || { tail -40 gpurun_out/r03_${TAG}_tests.log; exit 1; }
tail -3 gpurun_out/r03_${TAG}_tests.log
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03_${TAG}_bench_default.json 2> gpurun_out/r03_${TAG}_bench_default.err || { tail -20 gpurun_out/r03_${TAG}_bench_default.err; exit 1; }
timeout -k 10 300 python bench.py --workload cfg2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03_${TAG}_bench_cfg2.json 2>/dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_final
(cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final -- python3 bench.py --workload cfg2 --steps 2 --warmup 1 --no-cpu-baseline > /tmp/final_prof2.log 2>&1) || { tail -5 /tmp/final_prof2.log; exit 1; }
cp $(find /tmp/prof_final -name '*kernel_stats.csv' | head -1) $R/gpurun_out/r03_${TAG}_kernel_stats_cfg2.csv
rm -rf /tmp/prof_final
(cd $R && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/final_prof.log 2>&1) || { tail -5 /tmp/final_prof.log; exit 1; }
cp $(find /tmp/prof_final -name '*kernel_stats.csv' | head -1) $R/gpurun_out/r03_${TAG}_kernel_stats_default.csv
cd $R
python - <<PY
import json
for f in ("gpurun_out/r03_${TAG}_bench_default.json","gpurun_out/r03_${TAG}_bench_cfg2.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["value"],1), round(d["ms_per_step"],1), d["roofline"]["achieved"], d.get("stage_ms_per_step_rank0"))
PY
