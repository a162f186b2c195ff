"""host-side timeline of HotPath.run on the config-1 clips (diagnostic): where does the wall time go?"""
import sys, os, time, wave as wavmod
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from targetdiarization_amd import pipeline
from targetdiarization_amd.pipeline import HotPath
from targetdiarization_amd.weights import (recipe_state_dict, recipe_eres2netv2_state_dict, recipe_paraformer_state_dict, recipe_paraformer_decoder_state_dict)
def load(fn):
    with wavmod.open(os.path.join("tests/golden", fn), "rb") as w:
        return np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float32) / 32768.0
mix, tgt = load("chat_mix.wav"), load("female_a.wav")
asr_sd = dict(recipe_paraformer_state_dict(0, 50)); asr_sd.update(recipe_paraformer_decoder_state_dict(0, 16))
hp = HotPath(recipe_state_dict(0, 24), recipe_eres2netv2_state_dict(0), asr_sd)
for _ in range(3): hp.run([mix], target_clip=tgt)
marks = []
def mark(name): marks.append((name, time.perf_counter()))
# wrap the stage methods to log host enter/exit times
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        mark(name + " >"); r = f(*a, **k); mark(name + " <"); return r
    setattr(obj, name, g)
wrap(hp, "separate_device"); wrap(hp.spk, "embed_device"); wrap(hp, "encode_device"); wrap(hp.dec, "decode"); wrap(hp.ap, "louder_first_device")
torch.cuda.synchronize(); t0 = time.perf_counter(); marks.clear()
out = hp.run([mix], target_clip=tgt)
torch.cuda.synchronize(); t1 = time.perf_counter()
for n, t in marks: print(f"{1e3*(t-t0):7.2f} ms  {n}")
print(f"{1e3*(t1-t0):7.2f} ms  end")
