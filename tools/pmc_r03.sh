#!/bin/bash
# usage: tools/pmc_r03.sh  -> separate rocprofv3 --pmc passes over one step of `bench.py --workload cfg2` (MossFormer2 at M = 255 968 token rows):
# FETCH_SIZE, WRITE_SIZE (HBM-side traffic) and the SQ set (MFMA pipe busy cycles); per-launch means of the dominant kernel (to_hidden + to_qk GEMM
# with the fused conv epilogue), the attention GEMM, lin_k^T[v|u], to_out  -> gpurun_out/r03_pmc.json and profiles/traffic.json (with the source hash)
set -e
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_r03_$i -o p -- python3 bench.py --workload cfg2 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_r03_$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, json, collections, sys
sys.path.insert(0, ".")
import bench
out = {}
for pat, label in (("EpiHiddenConv", "to_hidden+to_qk GEMM (gemm_h3_kernel<TWOSEG, EpiHiddenConv>: SiLU + depthwise conv of v|u fused into the epilogue)"),
                   ("EpiAttnGatePlOut", "attention GEMM (gemm_h3a_kernel<TWOSEG, EpiAttnGatePlOut>)"),
                   ("EpiStore", "lin_k^T [v|u] GEMM (gemm_h3a_kernel<one segment, EpiStore>)"), ("EpiHiddenSN<8", "to_out GEMM (8 scale segments)"),
                   ("conv17_kernel<1", "conv17<1> (to_out depthwise conv + residual -> x, x planes)")):
    acc = collections.defaultdict(lambda: [0.0, 0])
    dur = []
    for f in glob.glob("gpurun_out/pmc_r03_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for f in glob.glob("gpurun_out/pmc_r03_1/**/*kernel_trace.csv", recursive=True):
        dur += [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
    c = {k: s / n for k, (s, n) in acc.items()}
    o = {"counters_mean_per_launch": c, "launches": max((n for _, n in acc.values()), default=0)}
    if dur and c:
        us = sum(dur) / len(dur)
        o["duration_us_mean_under_pmc"] = us
        if "GRBM_GUI_ACTIVE" in c:
            o["clock_GHz_held"] = c["GRBM_GUI_ACTIVE"] / 8 / us / 1e3
            o["mfma_pipe_utilisation"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8)
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            o["hbm_side_read_bytes_x2_corrected"] = c["FETCH_SIZE"] * 1024 * 2        # MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128-B request on gfx950
            o["hbm_side_write_bytes"] = c["WRITE_SIZE"] * 1024
            o["hbm_side_bytes_per_launch"] = o["hbm_side_read_bytes_x2_corrected"] + o["hbm_side_write_bytes"]
            o["hbm_side_TBps"] = o["hbm_side_bytes_per_launch"] / us / 1e6
    out[label] = o
json.dump(out, open("gpurun_out/r03_pmc.json", "w"), indent=1)
k = [v for kk, v in out.items() if kk.startswith("to_hidden")][0]
M = 255968
tj = {"kernel": "gemm_h3_kernel<TWOSEG, EpiHiddenConv> (to_hidden+to_qk, token-shifted first K segment, SiLU + depthwise conv of v|u in the epilogue), config-2 shape M = 255 968, one launch",
      "source": "profiles/r03_pmc.json (tools/pmc_r03.sh, separate --pmc passes)", "kernel_source_sha": bench.kernel_source_sha(),
      "fetch_bytes_corrected_x2": k.get("hbm_side_read_bytes_x2_corrected"), "write_bytes_as_counted": k.get("hbm_side_write_bytes"),
      "gemm_to_hidden_hbm_bytes_per_launch": k.get("hbm_side_bytes_per_launch"), "token_rows": M,
      "gemm_to_hidden_hbm_bytes_per_token_row": k.get("hbm_side_bytes_per_launch") / M if k.get("hbm_side_bytes_per_launch") else None,
      "algorithmic_read_bytes": M * 2048 + 2176 * 2048, "algorithmic_write_bytes": M * (8192 + 4096 + 512),
      "mfma_pipe_utilisation": k.get("mfma_pipe_utilisation"), "clock_GHz_held": k.get("clock_GHz_held"),
      "note": "algorithmic bytes per token row: A planes 2 KB read; v|u K-major planes 8 KB + fp32 u 4 KB + to_qk pre-activations 0.5 KB written (the fp32 y of round 2, 8.5 KB written + 8 KB re-read by conv17<4>, is gone)"}
json.dump(tj, open("profiles/traffic.json", "w"), indent=1)
json.dump(tj, open("gpurun_out/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:2500])
PY
rm -rf gpurun_out/pmc_r03_*/
