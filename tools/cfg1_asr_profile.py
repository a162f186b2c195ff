"""per-kernel profile of the Paraformer encoder + CIF + decoder and of ERes2NetV2 at the config-1 size (2 streams x 8.665 s):
run under rocprofv3 --kernel-trace --stats; also prints wall times of graph replay vs eager launches"""
import sys, os, time, wave as wavmod
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from targetdiarization_amd.pipeline import HotPath
from targetdiarization_amd.weights import (recipe_state_dict, recipe_eres2netv2_state_dict, recipe_paraformer_state_dict, recipe_paraformer_decoder_state_dict)
with wavmod.open("tests/golden/chat_mix.wav", "rb") as w:
    mix = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float32) / 32768.0
asr_sd = dict(recipe_paraformer_state_dict(0, 50)); asr_sd.update(recipe_paraformer_decoder_state_dict(0, 16))
hp = HotPath(recipe_state_dict(0, 2), recipe_eres2netv2_state_dict(0), asr_sd)
x = torch.from_numpy(mix).cuda()
flat = [x, x * 0.5]
which = sys.argv[1] if len(sys.argv) > 1 else "enc"
def T():
    torch.cuda.synchronize(); return time.perf_counter()
if which == "enc":
    if len(sys.argv) > 2: hp.asr.graph_rows = 0       # eager launches: the profiler sees every kernel
    for _ in range(3): enc = hp.encode_device(flat)
    t0 = T()
    for _ in range(10): enc = hp.encode_device(flat)
    print(f"encoder (2 x {enc[0].shape[0]} frames): {(T() - t0) * 100:.2f} ms per call")
elif which == "dec":
    enc = torch.stack(hp.encode_device(flat))
    for _ in range(3): hp.dec.decode(enc)
    t0 = T()
    for _ in range(10): hp.dec.decode(enc)
    print(f"CIF + decoder: {(T() - t0) * 100:.2f} ms per call")
else:
    if len(sys.argv) > 2: hp.spk.model.graph_frames = 0
    for _ in range(3): hp.spk.embed_device(flat)
    t0 = T()
    for _ in range(10): hp.spk.embed_device(flat)
    print(f"2 embeddings: {(T() - t0) * 100:.2f} ms per call")
