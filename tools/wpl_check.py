"""large launches: the same 90 windows of 10 s through MossFormer2 as ONE launch sequence and as three of 30 — equal up to the
accumulation order of the split-K lin_k^T[v|u] launch (its chunking depends on the batch)?  (index arithmetic at 1.8 M token rows)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from targetdiarization_amd.separator import MossFormer2Separator
from targetdiarization_amd.weights import recipe_state_dict
sep = MossFormer2Separator(recipe_state_dict(0, 24), device="cuda:0")
g = torch.Generator().manual_seed(11)
x = (torch.randn(90, 160000, generator=g) * 0.1).cuda()
y90 = sep(x)
torch.cuda.synchronize()
y30 = torch.cat([sep(x[i:i + 30]) for i in range(0, 90, 30)])
torch.cuda.synchronize()
d = (y90 - y30).abs().max().item(); m = y30.abs().max().item()
print(f"90 vs 3x30 windows: max abs diff {d:.3e}, max abs {m:.3e}, rel {d / m:.3e}; finite {bool(torch.isfinite(y90).all())}")
t0 = time.perf_counter(); sep(x); torch.cuda.synchronize(); print(f"90 windows: {(time.perf_counter() - t0) * 1e3:.0f} ms")
sys.exit(0 if d / m < 1e-4 else 1)
