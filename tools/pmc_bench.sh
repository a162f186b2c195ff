#!/bin/bash
# usage: tools/pmc_bench.sh <tag> <kernel-name-substring>   -> per-launch PMC means of that kernel over one bench step
set -e
export TMPDIR=/tmp
tag=$1; pat=$2
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmcb_${tag}_$i -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmcb_${tag}_$i.log 2>&1
done
python3 - "$tag" "$pat" <<'PY'
import csv, glob, sys, collections
tag, pat = sys.argv[1], sys.argv[2]
for d in sorted(glob.glob(f"gpurun_out/pmcb_{tag}_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if pat not in r["Kernel_Name"]: continue
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, (s, n) in acc.items():
            print(f"{tag} {k} = {s/n:.4g} per launch ({n} launches)")
    for f in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
        ds=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(f)) if pat in r['Kernel_Name']]
        if ds: print(f"{tag} duration us mean {sum(ds)/len(ds):.0f} ({len(ds)})")
PY
