#!/bin/bash
# default bench with larger ERes2NetV2 / Paraformer launch sequences
for cfg in "40000 32768" "80000 32768" "160000 32768" "40000 65536" "40000 131072"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --embed-frames-per-launch $1 --asr-rows-per-launch $2 > /tmp/bs.json 2>/tmp/bs.err || { tail -3 /tmp/bs.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("/tmp/bs.json").read().strip().split("\n")[-1])
print("embed frames $1, asr rows $2: RTF %.1f, %.0f ms/step, stages %s" % (d["value"], d["ms_per_step"], {k: round(v) for k, v in d["stage_ms_per_step_rank0"].items()}), flush=True)
PY
done
