#!/bin/bash
# A/B of the chunked block -> tile map of the wide x3 kernel (TDX_H3_MC=0: one sweep of the XCD's whole M range per N group): parity subset,
# config 2 timing, HBM-side FETCH_SIZE of the fused to_hidden launch
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_mossformer2.py tests/test_gpu_h3.py tests/test_gpu_eres2net.py -x -q > gpurun_out/r03_mc_tests.log 2>&1 || { tail -40 gpurun_out/r03_mc_tests.log; exit 1; }
tail -2 gpurun_out/r03_mc_tests.log
for f in 0 -1 0 -1; do
  TDX_H3_MC=$f timeout -k 10 300 python bench.py --workload cfg2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 mc=$f', round(d['ms_per_step'],2), 'ms', round(d['roofline']['ms_per_launch'],3), 'ms/launch')" || exit 1
done
export TMPDIR=/tmp
for f in 0 -1; do
  rm -rf gpurun_out/pmc_mc
  TDX_H3_MC=$f rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_mc -o p -- python3 bench.py --workload cfg2 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_mc.log 2>&1
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("gpurun_out/pmc_mc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for pat in ("EpiHiddenConv", "EpiHiddenSN<8", "EpiAttnGate"):
            if pat in k: a = acc[pat]; a[0] += float(r["Counter_Value"]); a[1] += 1
print("mc=$f FETCH_SIZE x2 (GB per launch):", {k: round(v[0] / v[1] * 1024 * 2 / 1e9, 3) for k, v in acc.items()})
PY
done
rm -rf gpurun_out/pmc_mc
