#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_sep
(cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sep -- python3 tools/cfg1_sep_profile.py > /tmp/sep.log 2>&1) || { tail -5 /tmp/sep.log; exit 1; }
f=$(find /tmp/prof_sep -name '*kernel_stats.csv' | head -1)
cp $f $R/gpurun_out/kernel_stats_cfg1_sep.csv
python3 $R/tools/kstats.py $f 5 30 | cut -c1-190
