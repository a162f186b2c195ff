"""ORACLE — test infrastructure, not product code.

CPU restatement (functional PyTorch) of the Paraformer-large SANM encoder that
`funasr.AutoModel(...).generate` runs (reached from ASRProcessor.py:424; model built :222-231).
funasr is a THIRD-PARTY dependency: un-vendored, unpinned (requirements.txt:24), not installed,
and no weights exist in the reference tree => PARITY UNPINNED.  This restates the published
funasr architecture (SANMEncoder / EncoderLayerSANM / MultiHeadedAttentionSANM /
SinusoidalPositionEncoder): 1 + 49 pre-LN layers, d=512, 4 heads,
FFN 2048 ReLU, FSMN memory v + dwconv11(v) added to the attention output, LayerNorm eps 1e-12.
State-dict names follow funasr's (`encoder.encoders0.0.*`, `encoder.encoders.{i}.*`,
`encoder.after_norm.*`).  All sequences of a batch have equal length (no padding mask).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

LN_EPS = 1e-12


def sinusoidal_pe(T: int, depth: int, dtype=torch.float32):
    """SinusoidalPositionEncoder.encode: positions 1..T, [sin | cos] halves."""
    pos = torch.arange(1, T + 1, dtype=dtype)
    inc = math.log(10000.0) / (depth / 2 - 1)
    inv = torch.exp(torch.arange(depth // 2, dtype=dtype) * (-inc))
    st = pos[:, None] * inv[None, :]
    return torch.cat((torch.sin(st), torch.cos(st)), dim=1)          # [T, depth]


def _sanm_attention(x, sd, p, heads=4, ksize=11):
    """MultiHeadedAttentionSANM.forward (no mask): softmax(q k^T / sqrt(dk)) v W_o + (v + dwconv11(v))."""
    B, T, _ = x.shape
    qkv = F.linear(x, sd[p + "linear_q_k_v.weight"], sd[p + "linear_q_k_v.bias"])
    n = qkv.shape[-1] // 3
    q, k, v = torch.split(qkv, n, dim=-1)
    dk = n // heads
    qh = q.reshape(B, T, heads, dk).transpose(1, 2) * dk ** (-0.5)
    kh = k.reshape(B, T, heads, dk).transpose(1, 2)
    vh = v.reshape(B, T, heads, dk).transpose(1, 2)
    att = torch.softmax(torch.matmul(qh, kh.transpose(-2, -1)), dim=-1)
    ctx = torch.matmul(att, vh).transpose(1, 2).reshape(B, T, n)
    out = F.linear(ctx, sd[p + "linear_out.weight"], sd[p + "linear_out.bias"])
    left = (ksize - 1) // 2
    mem = F.conv1d(F.pad(v.transpose(1, 2), (left, ksize - 1 - left)), sd[p + "fsmn_block.weight"], groups=n).transpose(1, 2) + v
    return out + mem


def _layer(x, sd, p):
    """EncoderLayerSANM.forward, normalize_before=True; no residual around attention when the
    input width differs from the model width (first layer, 560 -> 512)."""
    size = sd[p + "norm2.weight"].shape[0]
    in_size = sd[p + "norm1.weight"].shape[0]
    h = F.layer_norm(x, (in_size,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], LN_EPS)
    a = _sanm_attention(h, sd, p + "self_attn.")
    x = x + a if in_size == size else a
    h = F.layer_norm(x, (size,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], LN_EPS)
    h = F.linear(F.relu(F.linear(h, sd[p + "feed_forward.w_1.weight"], sd[p + "feed_forward.w_1.bias"])),
                 sd[p + "feed_forward.w_2.weight"], sd[p + "feed_forward.w_2.bias"])
    return x + h


def sanm_encoder_forward(feats: torch.Tensor, sd, num_blocks: int | None = None):
    """feats [B,T,560] (LFR+CMVN features) -> [B,T,512]."""
    if num_blocks is None:
        num_blocks = 2 + max(int(k.split("encoders.")[1].split(".")[0]) for k in sd if ".encoders." in k)
    B, T, D = feats.shape
    x = feats * (512 ** 0.5) + sinusoidal_pe(T, D, feats.dtype)[None]
    x = _layer(x, sd, "encoder.encoders0.0.")
    for i in range(num_blocks - 1):
        x = _layer(x, sd, f"encoder.encoders.{i}.")
    return F.layer_norm(x, (512,), sd["encoder.after_norm.weight"], sd["encoder.after_norm.bias"], LN_EPS)
