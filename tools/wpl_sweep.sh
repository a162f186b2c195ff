#!/bin/bash
# default bench (BASELINE configs[3]) with different numbers of 10 s windows per MossFormer2 launch: do intermediates that fit the
# 256 MB Infinity Cache pay for the smaller launches?
for w in ${WPL:-2 4 8 15 30}; do
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --windows-per-launch $w > /tmp/wpl_$w.json 2>/tmp/wpl_$w.err || { tail -3 /tmp/wpl_$w.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("/tmp/wpl_$w.json").read().strip().split("\n")[-1])
print("windows per launch $w: RTF %.1f, %.0f ms/step, stages %s, to_hidden %.0f TF" % (d["value"], d["ms_per_step"], {k: round(v) for k, v in d["stage_ms_per_step_rank0"].items()}, d["roofline"]["achieved"]), flush=True)
PY
done
