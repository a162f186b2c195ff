#!/bin/bash
# per-kernel times of `bench.py --workload cfg2` with the wide (TDX_H3A=0) and the half-height (TDX_H3A=1) attention kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for a in ${CFGS:-0 1}; do
  export TDX_H3A=$a
  rm -rf /tmp/prof_h3a
  (cd $R && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h3a -- python3 bench.py --workload cfg2 --steps 2 --warmup 1 --no-cpu-baseline > /tmp/bench_$a.log 2>&1) || { tail -5 /tmp/bench_$a.log; exit 1; }
  f=$(find /tmp/prof_h3a -name '*kernel_stats.csv' | head -1)
  cp $f $R/gpurun_out/kernel_stats_cfg2_h3a$a.csv
  echo "== TDX_H3A=$a"; tail -1 /tmp/bench_$a.log | cut -c1-300
  python3 $R/tools/kstats.py $f 2>/dev/null | head -12 | cut -c1-170
done
