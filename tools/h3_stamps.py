"""per-block phase times of the attention launch from in-kernel wall-clock stamps (diagnostic build of the library with
-DTDX_H3_STAMPS: `python tools/h3_stamps.py build` here, then on the GPU box `TDX_H3A=0|1 python tools/h3_stamps.py run <tag>`)."""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "targetdiarization_amd")
LIBS = os.path.join(PKG, "libtdx_stamps.so")
if sys.argv[1] == "build":
    from targetdiarization_amd import build
    objs = []
    for src in build.SOURCES:
        obj = os.path.join(build.OBJ, os.path.splitext(src)[0] + ("_stamps.o" if src == "mf2.hip" else ".o"))
        if src == "mf2.hip":
            subprocess.run([build.HIPCC] + build.FLAGS + ["-DTDX_H3_STAMPS", "-c", os.path.join(build.CSRC, src), "-o", obj], check=True)
        objs.append(obj)
    subprocess.run([build.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIBS] + objs, check=True)
    print(LIBS); sys.exit(0)
import numpy as np
if sys.argv[1] == "all":
    # every x3 launch of a config-2 forward: TDX_H3_STAMPS_DIR dumps (4th launch of each kernel instantiation)
    d = "gpurun_out/h3_stamps"
    os.makedirs(d, exist_ok=True)
    os.environ["TDX_H3_STAMPS_DIR"] = d
    import torch
    from targetdiarization_amd import _lib
    _lib.LIB_PATH = LIBS
    from targetdiarization_amd.separator import MossFormer2Separator
    from targetdiarization_amd.weights import recipe_state_dict
    sep = MossFormer2Separator(recipe_state_dict(0, 6), device="cuda:0")
    g = torch.Generator().manual_seed(5)
    xb = (torch.randn(30, 64000, generator=g) * 0.1).cuda()
    sep(xb); torch.cuda.synchronize()
    import glob, subprocess
    print(f"{'kernel':60s} {'blocks':>6s} {'span us':>8s} | {'prologue':>8s} {'loop':>8s} {'epilogue':>8s} | per block us")
    for fn in sorted(glob.glob(d + "/*.bin")):
        a = np.fromfile(fn, dtype=np.uint64).reshape(-1, 8)
        a = a[a[:, 0] > 0]
        if not len(a): continue
        t = a[:, :5].astype(np.int64)
        ok = t[:, 4] > 0
        t = t[ok]
        kind, name = os.path.basename(fn)[:-4].split("_", 1)
        mang = name.split("_M")[0]
        dem = subprocess.run(["c++filt", "-t", mang], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
        pro = (t[:, 1] - t[:, 0]).mean() / 100.0; loop = (t[:, 2] - t[:, 1]).mean() / 100.0; epi = (t[:, 4] - t[:, 2]).mean() / 100.0
        print(f"{(kind + ' ' + dem + ' M' + name.split('_M')[1])[:60]:60s} {len(t):6d} {(t[:, 4].max() - t[:, 0].min()) / 100.0:8.1f} | {pro:8.2f} {loop:8.2f} {epi:8.2f} | {(t[:, 4] - t[:, 0]).mean() / 100.0:8.2f}")
        if "EpiHiddenConv" in dem:      # the fused conv epilogue: wave 0's stamps 2 (loop done) -> 3 (half 0 staged) -> 5 (half 0 convolved) -> 6 (half 1 staged) -> 4 (end)
            u = a[ok][:, [2, 3, 5, 6, 4]].astype(np.int64)
            u = u[(u[:, 1] > 0) & (u[:, 2] > 0) & (u[:, 3] > 0)]          # (v|u tiles only: the to_qk tiles do not take the conv path)
            d = np.diff(u, axis=1) / 100.0
            print("    conv epilogue (wave 0), us: stage half 0 %.2f | conv half 0 %.2f | barrier + stage half 1 %.2f | conv half 1 %.2f   (%d tiles)" % (*d.mean(axis=0), len(u)))
    sys.exit(0)
tag = sys.argv[2]
os.makedirs("gpurun_out", exist_ok=True)
fn = f"gpurun_out/h3_stamps_{tag}.bin"
if sys.argv[1] == "run":
    os.environ["TDX_H3_STAMPS"] = fn
    import torch
    from targetdiarization_amd import _lib
    _lib.LIB_PATH = LIBS
    from targetdiarization_amd.separator import MossFormer2Separator
    from targetdiarization_amd.weights import recipe_state_dict
    sep = MossFormer2Separator(recipe_state_dict(0, 6), device="cuda:0")
    g = torch.Generator().manual_seed(5)
    xb = (torch.randn(30, 64000, generator=g) * 0.1).cuda()
    sep(xb); torch.cuda.synchronize()
a = np.fromfile(fn, dtype=np.uint64).reshape(-1, 8)
a = a[a[:, 0] > 0]
t = a[:, :5].astype(np.int64)
t0 = t[:, 0].min()
ph = np.diff(t, axis=1) / 100.0          # us (100 MHz)
names = ["setup+prologue", "main loop", "epilogue phase 1", "epilogue phase 2"]
print(f"{tag}: {len(a)} blocks, launch span {(t[:, 4].max() - t0) / 100.0:.1f} us")
for i, n in enumerate(names):
    print(f"  {n:18s} mean {ph[:, i].mean():7.2f} us  p10 {np.percentile(ph[:, i], 10):7.2f}  p90 {np.percentile(ph[:, i], 90):7.2f}")
print(f"  block total        mean {(t[:, 4] - t[:, 0]).mean() / 100.0:7.2f} us")
hw = a[:, 7]
cu = ((hw >> 32) & 0xf) * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 0xf)      # xcc, se, sh, cu
one = np.where(cu == cu[0])[0]
order = one[np.argsort(t[one, 0])]
print("  timeline of one CU (start, prologue done, loop done, phase 1 done, end; us from launch start):")
for b in order[:12]:
    print("    blk %6d wave-slot %2d: " % (b, hw[b] & 0xf) + "  ".join(f"{(x - t0) / 100.0:8.2f}" for x in t[b]))
gaps = []
for c in np.unique(cu):
    idx = np.where(cu == c)[0]
    st = np.sort(t[idx, 0]); en = np.sort(t[idx, 4])
print(f"  CUs seen: {len(np.unique(cu))}, blocks per CU mean {len(a) / len(np.unique(cu)):.1f}")
