"""soak: N calls of TargetDiarization.infer on the config-1 assets (device decoder, MDX body, punctuation); device memory before / after"""
import os, sys, time, wave
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from targetdiarization_amd.target_diarization import TargetDiarization
from targetdiarization_amd.weights import (recipe_state_dict, recipe_eres2netv2_state_dict, recipe_paraformer_state_dict,
                                           recipe_paraformer_decoder_state_dict, recipe_punc_state_dict)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
def load(fn):
    with wave.open(os.path.join("tests/golden", fn), "rb") as w:
        return np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float32) / 32768.0
mix, tgt = load("chat_mix.wav"), load("female_a.wav")
asr_sd = dict(recipe_paraformer_state_dict(0, 4)); asr_sd.update(recipe_paraformer_decoder_state_dict(0, 2))
sd_rows = {"text": [[0.0, 3.0, 0], [2.4, 5.5, 1], [5.5, 8.6, 0]]}
od = [(0.0, 3.0, "SPEAKER_00"), (2.4, 5.5, "SPEAKER_01"), (5.5, 8.6, "SPEAKER_00")]
td = TargetDiarization(cuda_device=0, sep_state_dict=recipe_state_dict(0, 4), spk_state_dict=recipe_eres2netv2_state_dict(0), asr_state_dict=asr_sd,
                       sd_pipeline=lambda a: sd_rows, od_pipeline=lambda a: od, token_list=[chr(0x4e00 + i) for i in range(8404)],
                       punc_state_dict=recipe_punc_state_dict(0, vocab=4096))
free0 = None
for i in range(N):
    n = len(mix) - (i % 7) * 1600          # different lengths: new shapes for the graph runner / workspaces
    t0 = time.perf_counter()
    spk, res, aud = td.infer(mix[:n], tgt)
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if i == 8: free0 = free
    if i % 5 == 0 or i == N - 1:
        print(f"call {i:3d}: {1e3 * (time.perf_counter() - t0):7.1f} ms  torch allocated {torch.cuda.memory_allocated() / 2**20:8.1f} MiB  reserved {torch.cuda.memory_reserved() / 2**20:8.1f} MiB  device used {(total - free) / 2**20:9.1f} MiB", flush=True)
print("device memory growth after call 8:", (free0 - free) / 2**20, "MiB")
