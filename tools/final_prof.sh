#!/bin/bash
# the round's reference measurements: default bench line, its rocprofv3 kernel stats, cfg2 and cfg1 lines
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 400 python bench.py --steps 5 --warmup 2 > gpurun_out/final_bench_default.json 2> gpurun_out/final_bench_default.err || exit 1
timeout -k 10 300 python bench.py --workload cfg2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/final_bench_cfg2.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --workload cfg1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/final_bench_cfg1.json 2>/dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_final
(cd $R && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/final_prof.log 2>&1) || { tail -5 /tmp/final_prof.log; exit 1; }
cp $(find /tmp/prof_final -name '*kernel_stats.csv' | head -1) $R/gpurun_out/final_kernel_stats_default.csv
rm -rf /tmp/prof_final
(cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final -- python3 bench.py --workload cfg2 --steps 2 --warmup 1 --no-cpu-baseline > /tmp/final_prof2.log 2>&1) || { tail -5 /tmp/final_prof2.log; exit 1; }
cp $(find /tmp/prof_final -name '*kernel_stats.csv' | head -1) $R/gpurun_out/final_kernel_stats_cfg2.csv
