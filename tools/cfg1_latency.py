"""config 1 (single 8.665 s clip + 1.9 s target clip): latency of the hot-path subset, stage by stage (diagnostic)"""
import sys, os, time, wave as wavmod
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from targetdiarization_amd.pipeline import HotPath
from targetdiarization_amd.weights import (recipe_state_dict, recipe_eres2netv2_state_dict, recipe_paraformer_state_dict,
                                           recipe_paraformer_decoder_state_dict)
def load(fn):
    with wavmod.open(os.path.join("tests/golden", fn), "rb") as w:
        return np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float32) / 32768.0
mix, tgt = load("chat_mix.wav"), load("female_a.wav")
asr_sd = dict(recipe_paraformer_state_dict(0, 50)); asr_sd.update(recipe_paraformer_decoder_state_dict(0, 16))
hp = HotPath(recipe_state_dict(0, 24), recipe_eres2netv2_state_dict(0), asr_sd)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for i in range(5):
    t0 = T(); temb = hp.spk.get_speaker_embedding(tgt)
    t1 = T(); pairs = hp.separate_device([mix])
    t2 = T(); flat = [p[k] for p in pairs for k in (0, 1)]; emb = hp.spk.embed_device(flat)
    t3 = T(); enc = hp.encode_device(flat)
    t4 = T(); dec = [hp.dec.decode(e[None]) for e in enc]
    t5 = T(); both = hp.dec.decode(torch.stack(enc))
    t6 = T(); out = hp.run([mix], temb)
    t7 = T()
    print(f"iter {i}: target emb {1e3*(t1-t0):.1f} | separate+loudness {1e3*(t2-t1):.1f} | 2 stream embs {1e3*(t3-t2):.1f} | encoder {1e3*(t4-t3):.1f} | "
          f"decoder 2x1 {1e3*(t5-t4):.1f} / 1x2 {1e3*(t6-t5):.1f} | run() {1e3*(t7-t6):.1f} ms", flush=True)
