"""config 1 (single 8.665 s clip + 1.9 s target clip): latency of the hot-path subset (diagnostic)"""
import sys, os, time, wave as wavmod
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from targetdiarization_amd.pipeline import HotPath
from targetdiarization_amd.weights import recipe_state_dict, recipe_eres2netv2_state_dict, recipe_paraformer_state_dict
def load(fn):
    with wavmod.open(os.path.join("tests/golden", fn), "rb") as w:
        return np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float32) / 32768.0
mix, tgt = load("chat_mix.wav"), load("female_a.wav")
hp = HotPath(recipe_state_dict(0, 24), recipe_eres2netv2_state_dict(0), recipe_paraformer_state_dict(0, 50))
temb = hp.spk.get_speaker_embedding(tgt)
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sep = hp.separate([mix]); torch.cuda.synchronize(); t1 = time.perf_counter()
    out = hp.run([mix], temb); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"iter {i}: separate {1e3*(t1-t0):.1f} ms (RTF {len(mix)/16000/(t1-t0):.0f}); whole subset {1e3*(t2-t1):.1f} ms (RTF {len(mix)/16000/(t2-t1):.0f})", flush=True)
