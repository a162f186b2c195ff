"""Localise the parity loss under heavy-tailed weights: layer-0 FLASH internals from the fp64 oracle -> device cal_attention."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
from oracle import mossformer2_oracle as orc
from targetdiarization_amd import ops
from targetdiarization_amd.weights import recipe_state_dict, recipe_wave

sd2 = recipe_state_dict(seed=1, num_blocks=2)
dev = torch.device("cuda:0")
x = torch.from_numpy(recipe_wave("heavy", 2, 4803))
g = torch.Generator().manual_seed(11)
sd = {k: v.clone() for k, v in sd2.items()}
p = "mask_net.mdl.intra_mdl.mossformerM.layers.0."
idx_h = torch.randperm(2048, generator=g)[:5]; idx_q = torch.randperm(128, generator=g)[:5]
sd[p + "to_hidden.mdl.1.weight"][idx_h] *= 100.0
sd[p + "to_qk.mdl.1.weight"][idx_q] *= 100.0
sd[p + "to_hidden.mdl.1.bias"][idx_h] *= 30.0
print("outlier v|u channels", sorted(idx_h.tolist()), "qk channels", sorted(idx_q.tolist()))
sd64 = orc.cast_state_dict(sd, torch.float64)
taps = {}
orc.mossformer2_forward(x.double(), sd64, taps=taps)
z = taps["z"].transpose(1, 2) if taps["z"].shape[-1] != 512 else taps["z"]          # [B,S,512]
half = 256
xs = torch.cat((F.pad(z[..., :half], (0, 0, 1, -1)), z[..., half:]), dim=-1)
hid = orc._ffconvm(xs, sd64, p + "to_hidden.", "scale")
v, u = hid.chunk(2, dim=-1)
qk = orc._ffconvm(xs, sd64, p + "to_qk.", "scale")
gam, bet = sd64[p + "qk_offset_scale.gamma"], sd64[p + "qk_offset_scale.beta"]
heads = [qk * gam[h] + bet[h] for h in range(4)]
fr = sd64[p + "rotary_pos_emb.freqs"]
# fp32-rounded inputs: the reference point is fp64 arithmetic ON THESE inputs
h32 = [t.float() for t in heads]; v32, u32 = v.float(), u.float()
av64, au64 = orc.cal_attention(*[t.double() for t in h32], v32.double(), u32.double(), fr)
av32, au32 = orc.cal_attention(*h32, v32, u32, fr.float())
avd, aud = ops.cal_attention(*[t.to(dev) for t in h32], v32.to(dev), u32.to(dev), fr.float().to(dev))
def rel(a, b): return float((a.double().cpu() - b).norm() / b.norm())
print("att_v rel-L2: device %.2e  torch-fp32 %.2e" % (rel(avd, av64), rel(av32, av64)))
print("att_u rel-L2: device %.2e  torch-fp32 %.2e" % (rel(aud, au64), rel(au32, au64)))
# per-channel relative error (channel norms over tokens)
def chan(a, b):
    d = (a.double().cpu() - b); return (d.pow(2).sum((0, 1)).sqrt() / b.pow(2).sum((0, 1)).sqrt())
ce_d, ce_32 = chan(avd, av64), chan(av32, av64)
out_ch = sorted(c for c in idx_h.tolist() if c < 1024)
print("att_v per-channel rel err: device median %.2e max %.2e | fp32 median %.2e max %.2e" % (ce_d.median(), ce_d.max(), ce_32.median(), ce_32.max()))
print("  outlier channels", [(c, "%.1e" % ce_d[c]) for c in out_ch])
# gate
o64 = (au64 * v32.double()) * torch.sigmoid(av64 * u32.double())
od = (aud.double().cpu() * v32.double()) * torch.sigmoid(avd.double().cpu() * u32.double())
o32 = ((au32 * v32) * torch.sigmoid(av32 * u32)).double()
print("gate o rel-L2 (from device att): %.2e   (from fp32 att): %.2e" % (rel(od, o64), rel(o32, o64)))
n64 = orc._scale_norm(o64, sd64[p + "to_out.mdl.0.g"])
nd = orc._scale_norm(od, sd64[p + "to_out.mdl.0.g"]); n32 = orc._scale_norm(o32, sd64[p + "to_out.mdl.0.g"])
print("ScaleNorm(o) rel-L2: %.2e / %.2e" % (rel(nd, n64), rel(n32, n64)))
# magnitudes
print("dyn range: |A| inputs: quad_q max %.2e rms %.2e; v max %.2e rms %.2e; att_v max %.2e rms %.2e" % (h32[0].abs().max(), h32[0].pow(2).mean().sqrt(), v32.abs().max(), v32.pow(2).mean().sqrt(), av64.abs().max(), av64.pow(2).mean().sqrt()))
