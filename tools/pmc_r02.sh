#!/bin/bash
# usage: tools/pmc_r02.sh  -> separate rocprofv3 --pmc passes over one step of `bench.py --workload cfg2` (MossFormer2, the kernels of the
# full pipe's dominant stage at M = 255 968 token rows): FETCH_SIZE, WRITE_SIZE (HBM-side traffic) and the SQ set (MFMA pipe busy cycles);
# per-launch means of the dominant kernel (to_hidden+to_qk GEMM) and of the attention GEMM -> gpurun_out/r02_pmc.json
set -e
export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_r02_$i -o p -- python3 bench.py --workload cfg2 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_r02_$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, json, collections
out = {}
for pat, label in (("EpiHiddenSN<2", "to_hidden+to_qk GEMM (gemm_h3_kernel<TWOSEG, EpiHiddenSN<2,true>>)"), ("EpiAttnGatePlOut", "attention GEMM (gemm_h3a_kernel<TWOSEG, EpiAttnGatePlOut>: 128-row tiles, two blocks per CU)"),
                   ("EpiStore", "lin_k^T [v|u] GEMM (gemm_h3a_kernel<one segment, EpiStore>)"), ("EpiHiddenSN<8", "to_out GEMM (gemm_h3_kernel<EpiHiddenSN<8,false>>, 8 scale segments)"),
                   ("conv17_kernel<4", "conv17<4> (v|u depthwise conv -> K-major planes)")):
    acc = collections.defaultdict(lambda: [0.0, 0])
    dur = []
    for f in glob.glob("gpurun_out/pmc_r02_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for f in glob.glob("gpurun_out/pmc_r02_1/**/*kernel_trace.csv", recursive=True):
        dur += [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
    c = {k: s / n for k, (s, n) in acc.items()}
    o = {"counters_mean_per_launch": c, "launches": max(n for _, n in acc.values()) if acc else 0}
    if dur and c:
        us = sum(dur) / len(dur)
        o["duration_us_mean_under_pmc"] = us
        o["clock_GHz_held"] = c["GRBM_GUI_ACTIVE"] / 8 / us / 1e3
        o["mfma_pipe_utilisation"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8)
        o["hbm_side_read_bytes_x2_corrected"] = c["FETCH_SIZE"] * 1024 * 2
        o["hbm_side_write_bytes"] = c["WRITE_SIZE"] * 1024
        o["hbm_side_bytes_per_launch"] = o["hbm_side_read_bytes_x2_corrected"] + o["hbm_side_write_bytes"]
        o["hbm_side_TBps"] = o["hbm_side_bytes_per_launch"] / us / 1e6
    out[label] = o
json.dump(out, open("gpurun_out/r02_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
rm -rf gpurun_out/pmc_r02_*/
