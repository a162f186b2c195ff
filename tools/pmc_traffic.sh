#!/bin/bash
# usage: tools/pmc_traffic.sh <tag>  -> FETCH_SIZE / WRITE_SIZE passes (separate runs) over one bench step; per-launch means of the
# to_hidden launches (grid 9000 x 512 threads) of gemm_h3_kernel<..., EpiHidden>
set -e
export TMPDIR=/tmp
tag=$1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmct_${tag}_$c -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmct_${tag}_$c.log 2>&1
done
python3 - "$tag" <<'PY'
import csv, glob, sys, json
tag = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/pmct_{tag}_{c}/**/*counter_collection.csv", recursive=True):
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
             if "EpiHidden" in r["Kernel_Name"] and r["Counter_Name"] == c and int(r["Grid_Size"]) == 9000 * 512]
        out[c] = (sum(v) / len(v), len(v))
    for f in glob.glob(f"gpurun_out/pmct_{tag}_{c}/**/*kernel_trace.csv", recursive=True):
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f))
             if "EpiHidden" in r["Kernel_Name"] and int(r["Grid_Size_X"] if "Grid_Size_X" in r else r["Grid_Size"]) == 9000 * 512]
        if d: out[c + "_us"] = sum(d) / len(d)
print(json.dumps(out))
PY
