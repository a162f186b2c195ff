"""A/B of the attention launch: TDX_H3A=1 (half-height, two blocks per CU) vs TDX_H3A=0 (wide kernel): run twice, compare the saved outputs bit for bit.
usage: TDX_H3A=0 python tools/h3a_check.py a && TDX_H3A=1 TDX_H3A_SWAP=0 python tools/h3a_check.py b && python tools/h3a_check.py cmp
(TDX_H3A_SWAP=0: the wide kernel's segment order; with the default order the results agree to fp32 rounding, not bit for bit)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
out = "gpurun_out"
os.makedirs(out, exist_ok=True)
if sys.argv[1] == "cmp":
    a, b = np.load(f"{out}/h3a_a.npy"), np.load(f"{out}/h3a_b.npy")
    print("bit-identical:", np.array_equal(a, b), "max abs diff", float(np.abs(a - b).max()), "max abs", float(np.abs(a).max()))
    sys.exit(0 if np.array_equal(a, b) else 1)
import torch
from targetdiarization_amd.separator import MossFormer2Separator
from targetdiarization_amd.weights import recipe_state_dict
sep = MossFormer2Separator(recipe_state_dict(0, 24), device="cuda:0")
g = torch.Generator().manual_seed(5)
x = (torch.randn(3, 138640, generator=g) * 0.1).cuda()        # S = 17 328 + ragged group at the end
y = sep(x)
torch.cuda.synchronize()
np.save(f"{out}/h3a_{sys.argv[1]}.npy", y.cpu().numpy())
xb = (torch.randn(30, 64000, generator=g) * 0.1).cuda()
for _ in range(2): sep(xb)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): sep(xb)
torch.cuda.synchronize(); print(f"TDX_H3A={os.environ.get('TDX_H3A')}: config-2 step {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms", flush=True)
