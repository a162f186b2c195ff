#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in ${WHICH:-enc dec emb}; do
  rm -rf /tmp/prof_asr
  (cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_asr -- python3 tools/cfg1_asr_profile.py $w eager > /tmp/asr.log 2>&1) || { tail -5 /tmp/asr.log; exit 1; }
  f=$(find /tmp/prof_asr -name '*kernel_stats.csv' | head -1)
  echo "== $w: $(grep 'per call' /tmp/asr.log)"
  python3 $R/tools/kstats.py $f 13 24 2>/dev/null | cut -c1-200
done
