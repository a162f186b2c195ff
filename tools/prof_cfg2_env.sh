#!/bin/bash
# usage: prof_cfg2_env.sh <tag> [VAR=value ...]: rocprofv3 kernel stats of `bench.py --workload cfg2 --steps 2 --warmup 1` under the given environment
TAG=$1; shift
R=$GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_e
(cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e -- python3 bench.py --workload cfg2 --steps 2 --warmup 1 --no-cpu-baseline > /tmp/prof_e.log 2>&1) || { tail -5 /tmp/prof_e.log; exit 1; }
cp $(find /tmp/prof_e -name '*kernel_stats.csv' | head -1) $R/gpurun_out/kstats_cfg2_$TAG.csv
