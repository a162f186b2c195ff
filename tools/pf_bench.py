"""Paraformer encoder (50 layers) + CIF + decoder (16 layers) at the benchmark size (120 segments x 500 LFR frames): ms per call, for rocprofv3"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from targetdiarization_amd.paraformer import ParaformerDecoder, ParaformerEncoder
from targetdiarization_amd.weights import recipe_paraformer_decoder_state_dict, recipe_paraformer_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 120
sd = dict(recipe_paraformer_state_dict(0, 50)); sd.update(recipe_paraformer_decoder_state_dict(0, 16))
enc = ParaformerEncoder({k: v for k, v in sd.items() if k.startswith("encoder.")}, torch.device("cuda:0"))
dec = ParaformerDecoder(sd, torch.device("cuda:0"))
x = torch.randn(B, 500, 560, generator=torch.Generator().manual_seed(1)).cuda()
def run():
    e = enc.encode(x)
    return dec.decode(e)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
print(f"B={B} x T=500: encoder + CIF + decoder {e0.elapsed_time(e1):.1f} ms", flush=True)
