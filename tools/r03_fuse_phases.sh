#!/bin/bash
cd $GRAFT_REPO_ROOT
for d in 0 4 8 12 16 28 32; do
  TDX_H3_DEBUG=$d timeout -k 10 300 python bench.py --workload cfg2 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dbg=$d', round(d['ms_per_step'],2), 'ms', round(d['roofline']['ms_per_launch'],3), 'ms/launch')" || echo "dbg=$d failed"
done
