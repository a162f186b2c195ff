#!/bin/bash
# A/B of the fused conv epilogues (to_hidden + conv17<4>, to_out + conv17<1>, to_u|to_v + conv17<0>): parity subset, then cfg2 and the
# default workload with TDX_FUSE_CONV=0 / 1 on the same box
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_mossformer2.py tests/test_gpu_h3.py -x -q > gpurun_out/r03_fuse_tests.log 2>&1 || { tail -40 gpurun_out/r03_fuse_tests.log; exit 1; }
tail -2 gpurun_out/r03_fuse_tests.log
for f in 0 1 0 1; do
  TDX_FUSE_CONV=$f timeout -k 10 300 python bench.py --workload cfg2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 fuse=$f', round(d['ms_per_step'],2), 'ms', round(d['roofline']['ms_per_launch'],3), 'ms/launch')" || exit 1
done
for f in 0 1; do
  TDX_FUSE_CONV=$f timeout -k 10 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default fuse=$f', round(d['value'],1), 'RTF', round(d['ms_per_step'],1), 'ms', d['stage_ms_per_step_rank0'])" || exit 1
done
