"""per-kernel profile of ONE-window separation at the config-1 size (run under rocprofv3 --kernel-trace --stats)"""
import sys, os, wave as wavmod
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from targetdiarization_amd.separator import MossFormer2Separator
from targetdiarization_amd.weights import recipe_state_dict
with wavmod.open("tests/golden/chat_mix.wav", "rb") as w:
    mix = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float32) / 32768.0
sep = MossFormer2Separator(recipe_state_dict(0, 24), "cuda:0", graph_rows=0)
x = torch.from_numpy(mix)[None].cuda()
for _ in range(5):
    y = sep(x)
torch.cuda.synchronize()
