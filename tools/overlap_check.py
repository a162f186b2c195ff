"""do two small forwards on different HIP streams overlap? (diagnostic for the one-clip call pattern)"""
import sys, os, time, wave as wavmod
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from targetdiarization_amd.pipeline import HotPath
from targetdiarization_amd.weights import (recipe_state_dict, recipe_eres2netv2_state_dict, recipe_paraformer_state_dict)
def load(fn):
    with wavmod.open(os.path.join("tests/golden", fn), "rb") as w:
        return np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float32) / 32768.0
mix = torch.from_numpy(load("chat_mix.wav")).cuda()
hp = HotPath(recipe_state_dict(0, 24), recipe_eres2netv2_state_dict(0), recipe_paraformer_state_dict(0, 50))
feat = hp.asr.features(mix[None].repeat(2, 1))
def T():
    torch.cuda.synchronize(); return time.perf_counter()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def sep(): return hp.ap.separater(mix[None])
def enc(): return hp.asr.encode(feat)
def emb(): return hp.spk.model(mix[None].repeat(2, 1))
for name, f, g in (("sep|enc", sep, enc), ("sep|emb", sep, emb), ("emb|enc", emb, enc)):
    for _ in range(3): f(); g()
    t0 = T(); f(); t1 = T(); g(); t2 = T()
    with torch.cuda.stream(s1): f()
    with torch.cuda.stream(s2): g()
    t3 = T()
    print(f"{name}: alone {1e3*(t1-t0):.1f} + {1e3*(t2-t1):.1f} ms, on two streams {1e3*(t3-t2):.1f} ms", flush=True)
