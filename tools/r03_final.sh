#!/bin/bash
# end-of-round reference run at HEAD: GPU tests, default bench (with cpu baseline), cfg2, kernel stats, PMC passes, phase stamps
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r03_final_tests.log 2>&1 || { tail -40 gpurun_out/r03_final_tests.log; exit 1; }
tail -2 gpurun_out/r03_final_tests.log
