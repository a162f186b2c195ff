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


# ---------------------------------------------------------------------------------------------------------------
# N2: CIF predictor + non-autoregressive SANM decoder + token / timestamp assembly.  THIRD-PARTY (funasr, unpinned,
# absent): restated from the published funasr code (CifPredictorV2.forward / tail_process_fn / cif(), DecoderLayerSANM,
# MultiHeadedAttentionSANMDecoder, MultiHeadedAttentionCrossAtt, PositionwiseFeedForwardDecoderSANM,
# ParaformerSANMDecoder.forward, Paraformer.inference).  PARITY UNPINNED.
# ---------------------------------------------------------------------------------------------------------------
CIF_THRESHOLD, CIF_TAIL = 1.0, 0.45


def cif_alphas(enc: torch.Tensor, sd):
    """CifPredictorV2.forward up to the alphas (smooth_factor 1, noise_threshold 0), full-length sequences (mask = 1), with the
    tail frame of tail_process_fn appended: enc [B,T,512] -> hidden [B,T+1,512] (zero last frame), alphas [B,T+1]."""
    ctx = enc.transpose(1, 2)
    mem = F.conv1d(F.pad(ctx, (1, 1)), sd["predictor.cif_conv1d.weight"], sd["predictor.cif_conv1d.bias"])
    out = F.relu((mem + ctx).transpose(1, 2))
    alphas = torch.sigmoid(F.linear(out, sd["predictor.cif_output.weight"], sd["predictor.cif_output.bias"]))[..., 0]
    alphas = F.relu(alphas * 1.0 - 0.0)
    B, T, Dm = enc.shape
    alphas = torch.cat((alphas, torch.full((B, 1), CIF_TAIL, dtype=enc.dtype)), dim=1)
    hidden = torch.cat((enc, torch.zeros(B, 1, Dm, dtype=enc.dtype)), dim=1)
    return hidden, alphas


def cif(hidden: torch.Tensor, alphas: torch.Tensor, threshold: float = CIF_THRESHOLD):
    """funasr cif(): the sequential integrate-and-fire loop.  Returns (list of [n_b,512] fired frames, fires [B,T'])."""
    B, T, Dm = hidden.shape
    integrate = torch.zeros(B, dtype=hidden.dtype)
    frame = torch.zeros(B, Dm, dtype=hidden.dtype)
    fires, frames = [], []
    for t in range(T):
        alpha = alphas[:, t]
        completion = 1.0 - integrate
        integrate = integrate + alpha
        fires.append(integrate)
        fire = integrate >= threshold
        integrate = torch.where(fire, integrate - 1.0, integrate)
        cur = torch.where(fire, completion, alpha)
        remainds = alpha - cur
        frame = frame + cur[:, None] * hidden[:, t, :]
        frames.append(frame)
        frame = torch.where(fire[:, None], remainds[:, None] * hidden[:, t, :], frame)
    fires = torch.stack(fires, 1)
    frames = torch.stack(frames, 1)
    outs = [frames[b][fires[b] >= threshold] for b in range(B)]
    return outs, fires


def _ff_dec(x, sd, p):
    """PositionwiseFeedForwardDecoderSANM: w_2(LayerNorm(relu(w_1(x)))), w_2 without bias"""
    h = F.relu(F.linear(x, sd[p + "w_1.weight"], sd[p + "w_1.bias"]))
    h = F.layer_norm(h, (h.shape[-1],), sd[p + "norm.weight"], sd[p + "norm.bias"], LN_EPS)
    return F.linear(h, sd[p + "w_2.weight"])


def _dec_layer(x, memory, mask, sd, p, heads=4, ksize=11):
    """DecoderLayerSANM.forward (normalize_before): FFN first, FSMN memory block as "self attention", then cross attention"""
    residual = x
    tgt = _ff_dec(F.layer_norm(x, (512,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], LN_EPS), sd, p + "feed_forward.")
    t2 = F.layer_norm(tgt, (512,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], LN_EPS) * mask
    left = (ksize - 1) // 2
    mem = F.conv1d(F.pad(t2.transpose(1, 2), (left, ksize - 1 - left)), sd[p + "self_attn.fsmn_block.weight"], groups=512).transpose(1, 2)
    x = residual + (mem + t2) * mask
    residual = x
    q_in = F.layer_norm(x, (512,), sd[p + "norm3.weight"], sd[p + "norm3.bias"], LN_EPS)
    B, L, _ = q_in.shape
    T = memory.shape[1]
    dk = 512 // heads
    q = F.linear(q_in, sd[p + "src_attn.linear_q.weight"], sd[p + "src_attn.linear_q.bias"])
    kv = F.linear(memory, sd[p + "src_attn.linear_k_v.weight"], sd[p + "src_attn.linear_k_v.bias"])
    k, v = torch.split(kv, 512, dim=-1)
    qh = q.reshape(B, L, heads, dk).transpose(1, 2) * dk ** (-0.5)
    kh = k.reshape(B, T, heads, dk).transpose(1, 2)
    vh = v.reshape(B, T, heads, dk).transpose(1, 2)
    att = torch.softmax(torch.matmul(qh, kh.transpose(-2, -1)), dim=-1)
    ctx = torch.matmul(att, vh).transpose(1, 2).reshape(B, L, 512)
    return residual + F.linear(ctx, sd[p + "src_attn.linear_out.weight"], sd[p + "src_attn.linear_out.bias"])


def sanm_decoder_forward(embeds: torch.Tensor, lens, memory: torch.Tensor, sd, num_blocks: int | None = None):
    """ParaformerSANMDecoder.forward: embeds [B,L,512] (zero beyond lens[b]), memory = encoder output [B,T,512] -> logits [B,L,V]"""
    if num_blocks is None:
        num_blocks = 1 + max(int(k.split("decoders.")[1].split(".")[0]) for k in sd if "decoder.decoders." in k)
    B, L, _ = embeds.shape
    mask = (torch.arange(L)[None, :] < torch.as_tensor(lens)[:, None]).to(embeds.dtype)[..., None]
    x = embeds
    for i in range(num_blocks):
        x = _dec_layer(x, memory, mask, sd, f"decoder.decoders.{i}.")
    p = "decoder.decoders3.0."
    x = _ff_dec(F.layer_norm(x, (512,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], LN_EPS), sd, p + "feed_forward.")
    x = F.layer_norm(x, (512,), sd["decoder.after_norm.weight"], sd["decoder.after_norm.bias"], LN_EPS)
    return F.linear(x, sd["decoder.output_layer.weight"], sd["decoder.output_layer.bias"])


def paraformer_decode(enc: torch.Tensor, sd, num_blocks: int | None = None):
    """Paraformer.inference after the encoder: predictor -> CIF -> decoder -> argmax.  Returns per utterance
    (token ids, fire frame indices): token count = floor(sum alphas) as in CifPredictorV2 (tail included)."""
    hidden, alphas = cif_alphas(enc, sd)
    fired, fires = cif(hidden, alphas)
    counts = torch.floor(alphas.sum(-1)).long()
    L = int(counts.max())
    B = enc.shape[0]
    if L < 1:
        return [([], [])] * B, None
    emb = torch.zeros(B, L, 512, dtype=enc.dtype)
    for b in range(B):
        n = min(int(counts[b]), fired[b].shape[0])
        emb[b, :n] = fired[b][:n]
    logits = sanm_decoder_forward(emb, counts, enc, sd, num_blocks)
    ids = logits.argmax(-1)
    res = []
    for b in range(B):
        n = int(counts[b])
        peaks = torch.nonzero(fires[b] >= CIF_THRESHOLD)[:, 0][:n].tolist()
        res.append((ids[b, :n].tolist(), peaks))
    return res, logits
