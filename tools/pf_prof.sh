#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pfprof
(cd $GRAFT_REPO_ROOT && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pfprof -- python3 tools/pf_bench.py 120 > /tmp/pfprof.log 2>&1) || { tail -5 /tmp/pfprof.log; exit 1; }
grep "encoder" /tmp/pfprof.log
cp $(find /tmp/pfprof -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/pf_prof.csv
