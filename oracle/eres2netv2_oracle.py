"""ORACLE — test infrastructure, not product code.

CPU restatement (functional PyTorch) of ERes2NetV2-w24s4ep4, the speaker-embedding network
behind `self.embedding['eres2netv2_large'](wav, output_emb=True)` (TargetASR.py:155-163;
pipeline built :98-109 from modelscope model iic/speech_eres2netv2w24s4ep4_sv_zh-cn_16k-common).
modelscope / 3D-Speaker are THIRD-PARTY: un-vendored, unpinned (requirements.txt:25), not
installed, no weights in the reference tree => PARITY UNPINNED.  This restates the published
3D-Speaker architecture (speakerlab/models/eres2net/ERes2NetV2.py):
stem conv3x3+BN+ReLU, 4 stages of Res2Net bottlenecks [3,4,6,3] (planes 64/128/256/512,
baseWidth 24, scale 4, expansion 4, strides 1/2/2/2, ReLU20 = clamp(0,20)), AFF fusion inside
stages 3-4, layer3_ds + fuse34, TSTP pooling, Linear 40960 -> 192.  BatchNorm in eval mode.
State-dict names follow 3D-Speaker's module tree.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
NUM_BLOCKS = (3, 4, 6, 3)


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], False, 0.0, BN_EPS)


def _relu20(x):
    return torch.clamp(x, 0.0, 20.0)


def _aff(x, y, sd, p):
    """AFF.forward: att = 1 + tanh(BN(conv(SiLU(BN(conv(cat[x,y])))))); x*att + y*(2-att)."""
    xa = torch.cat((x, y), dim=1)
    t = F.silu(_bn(F.conv2d(xa, sd[p + "local_att.0.weight"], sd[p + "local_att.0.bias"]), sd, p + "local_att.1."))
    a = 1.0 + torch.tanh(_bn(F.conv2d(t, sd[p + "local_att.3.weight"], sd[p + "local_att.3.bias"]), sd, p + "local_att.4."))
    return x * a + y * (2.0 - a)


def _block(x, sd, p, stride, aff):
    """BasicBlockERes2NetV2(AFF).forward"""
    w4 = sd[p + "conv1.weight"].shape[0]
    width = w4 // 4
    out = _relu20(_bn(F.conv2d(x, sd[p + "conv1.weight"], stride=stride), sd, p + "bn1."))
    spx = torch.split(out, width, 1)
    outs = []
    sp = None
    for i in range(4):
        if i == 0:
            sp = spx[0]
        elif aff:
            sp = _aff(sp, spx[i], sd, p + f"fuse_models.{i - 1}.")
        else:
            sp = sp + spx[i]
        sp = _relu20(_bn(F.conv2d(sp, sd[p + f"convs.{i}.weight"], padding=1), sd, p + f"bns.{i}."))
        outs.append(sp)
    out = _bn(F.conv2d(torch.cat(outs, 1), sd[p + "conv3.weight"]), sd, p + "bn3.")
    if (p + "shortcut.0.weight") in sd:
        res = _bn(F.conv2d(x, sd[p + "shortcut.0.weight"], stride=stride), sd, p + "shortcut.1.")
    else:
        res = x
    return _relu20(out + res)


def eres2netv2_forward(feat: torch.Tensor, sd, taps=None):
    """feat [B,F,80] (fbank minus utterance mean) -> embedding [B,192]."""
    x = feat.permute(0, 2, 1).unsqueeze(1)                                   # [B,1,80,F]
    out = F.relu(_bn(F.conv2d(x, sd["conv1.weight"], padding=1), sd, "bn1."))
    feats = []
    for li, (nb, stride) in enumerate(zip(NUM_BLOCKS, (1, 2, 2, 2)), start=1):
        for i in range(nb):
            out = _block(out, sd, f"layer{li}.{i}.", stride if i == 0 else 1, aff=li >= 3)
        feats.append(out)
        if taps is not None:
            taps[f"layer{li}"] = out
    out3_ds = F.conv2d(feats[2], sd["layer3_ds.weight"], stride=2, padding=1)
    fuse = _aff(feats[3], out3_ds, sd, "fuse34.")
    B, C, Fq, T = fuse.shape
    v = fuse.reshape(B, C * Fq, T)
    stats = torch.cat((v.mean(dim=-1), torch.sqrt(v.var(dim=-1) + 1e-7)), dim=-1)   # TSTP (unbiased var)
    if taps is not None:
        taps["stats"] = stats
    return F.linear(stats, sd["seg_1.weight"], sd["seg_1.bias"])
