"""Which heavy-tailed weight family breaks parity (tests/test_gpu_mossformer2.py::test_static_scales_under_heavy_tailed_weights)?
Runs the 2-block model with each family alone and prints device-vs-fp64 errors at the taps (after_flash0, after_fsmn0, out)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import mossformer2_oracle as orc
from targetdiarization_amd.separator import MossFormer2Separator
from targetdiarization_amd.weights import recipe_state_dict, recipe_wave

sd2 = recipe_state_dict(seed=1, num_blocks=2)
dev = torch.device("cuda:0")
x = torch.from_numpy(recipe_wave("heavy", 2, 4803))


def variant(which):
    g = torch.Generator().manual_seed(11)
    sd = {k: v.clone() for k, v in sd2.items()}
    for l in range(2):
        p = f"mask_net.mdl.intra_mdl.mossformerM.layers.{l}."
        if "rows" in which:
            for name, rows in ((p + "to_hidden.mdl.1.weight", 2048), (p + "to_qk.mdl.1.weight", 128)):
                idx = torch.randperm(rows, generator=g)[:5]
                sd[name][idx] *= 100.0
                sd[name.replace("weight", "bias")][idx] *= 30.0
        idx_h = torch.randperm(2048, generator=g)[:5]; idx_q = torch.randperm(128, generator=g)[:5]
        if "rows_h" in which:
            sd[p + "to_hidden.mdl.1.weight"][idx_h] *= 100.0
        if "rows_q" in which:
            sd[p + "to_qk.mdl.1.weight"][idx_q] *= 100.0
        if "bias_h" in which:
            sd[p + "to_hidden.mdl.1.bias"][idx_h] *= 30.0
        if "bias_q" in which:
            sd[p + "to_qk.mdl.1.bias"][idx_q] *= 30.0
        if "gamma" in which:
            sd[p + "qk_offset_scale.gamma"] = sd[p + "qk_offset_scale.gamma"] * (1.0 + 7.0 * torch.rand(4, 128, generator=g))
        if "taps" in which:
            for cw in ("to_hidden.mdl.3.sequential.1.conv.weight", "to_qk.mdl.3.sequential.1.conv.weight"):
                ch = torch.randperm(sd[p + cw].shape[0], generator=g)[:6]
                sd[p + cw][ch] *= 8.0
        q = f"mask_net.mdl.intra_mdl.mossformerM.fsmn.{l}."
        if "prelu1" in which:
            sd[q + "conv1.1.weight"] = torch.tensor([-1.7 if l == 0 else 1.9])
        if "prelu256" in which:
            for nm in ("prelu1", "prelu2"):
                sd[q + f"gated_fsmn.fsmn.conv.{nm}.weight"] = torch.rand(256, generator=g) * 4.0 - 2.0
        if "inorm" in which:
            for nm in ("norm1", "norm2"):
                sd[q + f"gated_fsmn.fsmn.conv.{nm}.weight"] = sd[q + f"gated_fsmn.fsmn.conv.{nm}.weight"] * 50.0
                sd[q + f"gated_fsmn.fsmn.conv.{nm}.bias"] = sd[q + f"gated_fsmn.fsmn.conv.{nm}.bias"] * 50.0
    return sd


def rel(a, b):
    a = a.double().cpu()
    if a.shape != b.shape:
        b = b.transpose(-1, -2)
    return float((a - b).norm() / b.norm())


for which in (["none"], ["rows_h", "rows_q"], ["rows_h", "rows_q", "bias_h"], ["rows", "gamma", "taps", "prelu1", "prelu256", "inorm"]):
    sd = variant(which)
    sep = MossFormer2Separator(sd, device=dev, graph_rows=0)
    sep.enable_taps(True)
    out = sep(x.to(dev))
    taps = {}
    ref = orc.mossformer2_forward(x.double(), orc.cast_state_dict(sd, torch.float64), taps=taps)
    taps32 = {}
    ref32 = orc.mossformer2_forward(x, sd, taps=taps32)
    line = [f"{'+'.join(which):40s}"]
    for name in ("z", "after_flash0", "after_fsmn0", "after_stack"):
        line.append(f"{name} {rel(sep.tap(name), taps[name]):.1e}/{rel(taps32[name], taps[name]):.1e}")
    line.append(f"out {rel(out, ref):.1e}/{rel(ref32, ref):.1e}")
    print("  ".join(line), flush=True)
