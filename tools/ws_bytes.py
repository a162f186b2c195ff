"""peak device memory of a MossFormer2 forward at 1 / 30 / 90 / 180 windows of 10 s per launch sequence (workspace sizing)"""
import sys; sys.path.insert(0, '/root/repo')
import torch
from targetdiarization_amd.separator import MossFormer2Separator
from targetdiarization_amd.weights import recipe_state_dict
sep = MossFormer2Separator(recipe_state_dict(0, 24), device="cuda:0")
for B in (1, 30, 90, 180):
    x = torch.zeros(B, 160000, device="cuda:0")
    torch.cuda.reset_peak_memory_stats()
    y = sep(x); torch.cuda.synchronize()
    print(f"B={B}: peak allocated {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    del x, y
    sep._ws = None; torch.cuda.empty_cache()
