#!/bin/bash
# usage: tools/pmc_h3.sh <tag> <variant> [M N K]   -> gpurun_out/pmc_<tag>_*.csv summaries
set -e
export TMPDIR=/tmp
tag=$1; shift
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$i -o p -- python3 tools/h3_one.py "$@" > gpurun_out/pmc_${tag}_$i.log 2>&1
done
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
for d in sorted(glob.glob(f"gpurun_out/pmc_{tag}_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "gemm_h3" not in r["Kernel_Name"]: continue
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, (s, n) in acc.items():
            print(f"{tag} {k} = {s/n:.4g} per launch ({n} launches)")
PY
