import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synth_mixtures
from targetdiarization_amd.pipeline import HotPath
from targetdiarization_amd.weights import recipe_state_dict, recipe_eres2netv2_state_dict
from targetdiarization_amd import ops
hp = HotPath(recipe_state_dict(0, 24), recipe_eres2netv2_state_dict(0), None)
utts = [synth_mixtures(1, 480000, seed=5 + i)[0] for i in range(8)]
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for it in range(2):
    t0 = T()
    plans = [hp.ap.window_plan(len(u), 160000) for u in utts]
    wins = [u[s:e].astype(np.float32, copy=True) for u, plan in zip(utts, plans) for (s, e) in plan]
    t1 = T()
    outs = hp.ap.separate_windows_device(wins)
    t2 = T()
    pairs, k = [], 0
    for plan in plans:
        pairs.append(torch.cat(outs[k:k + len(plan)], dim=1)); k += len(plan)
    t3 = T()
    l = ops.loudness(torch.cat(pairs, 0), 16000)
    t4 = T()
    res = hp.ap.louder_first(pairs)
    t5 = T()
    emb = hp.embed_streams([s for pair in res for s in pair])
    t6 = T()
    print(f"plan {1e3*(t1-t0):.0f} sep {1e3*(t2-t1):.0f} cat {1e3*(t3-t2):.0f} loud {1e3*(t4-t3):.0f} louder_first {1e3*(t5-t4):.0f} embed {1e3*(t6-t5):.0f} ms", flush=True)
