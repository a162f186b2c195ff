#!/bin/bash
# A/B of the fused to_hidden + conv17<4> epilogue: parity subset, then cfg2 bench with TDX_FUSE_CONV=0 / 1
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_mossformer2.py tests/test_gpu_h3.py -x -q > gpurun_out/r03_fuse_tests.log 2>&1 || { tail -40 gpurun_out/r03_fuse_tests.log; exit 1; }
tail -2 gpurun_out/r03_fuse_tests.log
for f in 0 1 0 1; do
  TDX_FUSE_CONV=$f timeout -k 10 300 python bench.py --workload cfg2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse=$f', round(d['ms_per_step'],2), 'ms', round(d['roofline']['ms_per_launch'],3), 'ms/launch')" || exit 1
done
