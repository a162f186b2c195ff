#!/bin/bash
# per-kernel times of ERes2NetV2 forwards (B = 60, F = 998; 4 forwards in the run) -> gpurun_out/eres_prof_<tag>.csv
cd /tmp && export TMPDIR=/tmp
for s in "${@:-a}"; do
  rm -rf /tmp/eprof
  (cd $GRAFT_REPO_ROOT && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/eprof -- python3 tools/eres_bench.py child 60 /tmp/e.npy > /tmp/eprof.log 2>&1) || { tail -5 /tmp/eprof.log; exit 1; }
  cp $(find /tmp/eprof -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/eres_prof_$s.csv
done
