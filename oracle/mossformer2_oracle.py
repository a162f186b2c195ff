"""ORACLE — test infrastructure, not product code.

CPU restatement (functional PyTorch, fp32 or fp64) of the reference's MossFormer2
forward pass.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
import this file; the product path (targetdiarization_amd/) never does and fails loudly
when the HIP library is missing.

Parity status: PINNED for everything in the reference tree — oracle/make_goldens.py runs
the reference's own modules (imported by path in the build container) on the same recipe
weights/inputs and (a) asserts this restatement agrees with them, (b) writes the fixtures
under tests/golden/ that tests/test_oracle_golden.py re-checks everywhere.
UNPINNED for the rotary embedding: `rotary_embedding_torch` is an un-vendored, unpinned
third-party dependency (requirements.txt:28; call sites mossformer_block.py:6,230-233,453);
its published algorithm (lucidrains, RotaryEmbedding(dim=32), theta=10000, interleaved
pairs) is restated in `_rotary` below and in oracle/_load_reference.py's stand-in.

Every function cites the reference lines it follows (paths relative to /root/reference).
The arithmetic is written from the equations in SURVEY.md Appendix A, as plain tensor
ops on a {state_dict key: tensor} mapping — no nn.Module mirrors the reference classes.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

PFX = "mask_net.mdl.intra_mdl.mossformerM."


def _dwconv_tokens(y: torch.Tensor, w: torch.Tensor, dilation: int = 1) -> torch.Tensor:
    """Depthwise 'same' conv along the token axis of y[B,S,C]; w[C,1,k] (cross-correlation,
    zero padding (k-1)/2*dilation, no bias).  conv_module.py:138-177 (DepthwiseConv1d)."""
    k = w.shape[-1]
    pad = (k - 1) // 2 * dilation
    return F.conv1d(y.transpose(1, 2), w.reshape(w.shape[0], 1, k), padding=pad,
                    dilation=dilation, groups=w.shape[0]).transpose(1, 2)


def _scale_norm(x: torch.Tensor, g: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """ScaleNorm: x / clamp(||x||_2 * d^-0.5, eps) * g.  mossformer_block.py:44-54."""
    nrm = torch.linalg.vector_norm(x, dim=-1, keepdim=True) * (x.shape[-1] ** -0.5)
    return x / nrm.clamp(min=eps) * g


def _ffconvm(x, sd, p, norm: str):
    """FFConvM = norm -> Linear -> SiLU -> (y + dwconv17(y)); dropout = identity in eval.
    mossformer_block.py:89-102; ConvModule conv_module.py:180-220."""
    if norm == "scale":
        n = _scale_norm(x, sd[p + "mdl.0.g"])
    else:
        n = F.layer_norm(x, (x.shape[-1],), sd[p + "mdl.0.weight"], sd[p + "mdl.0.bias"], 1e-5)
    y = F.silu(F.linear(n, sd[p + "mdl.1.weight"], sd[p + "mdl.1.bias"]))
    return y + _dwconv_tokens(y, sd[p + "mdl.3.sequential.1.conv.weight"])


def _rotary(t: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    """rotary_embedding_torch.RotaryEmbedding(dim=32).rotate_queries_or_keys on t[B,S,D]:
    interleaved pairs (2i,2i+1) of the first 2*len(freqs) features rotated by angle
    pos*freqs[i], pos = 0..S-1 (absolute), angle product in t's dtype.
    Call site mossformer_block.py:230-233; third-party algorithm, see module docstring."""
    S = t.shape[-2]
    rd = 2 * freqs.shape[0]
    pos = torch.arange(S, device=t.device).to(freqs.dtype)
    ang = pos[:, None] * freqs[None, :]
    c, s = ang.cos(), ang.sin()
    a = t[..., 0:rd:2]
    b = t[..., 1:rd:2]
    ra = a * c - b * s
    rb = b * c + a * s
    rot = torch.stack((ra, rb), dim=-1).reshape(t.shape[:-1] + (rd,))
    return torch.cat((rot, t[..., rd:]), dim=-1)


def cal_attention(quad_q, lin_q, quad_k, lin_k, v, u, freqs, group: int = 256):
    """FLASH_ShareA_FFConvM.cal_attention, non-causal, no mask.  mossformer_block.py:222-294.
    Inputs [B,S,*]; returns att_v, att_u [B,S,E]."""
    B, S, _ = v.shape
    quad_q, lin_q, quad_k, lin_k = (_rotary(t, freqs) for t in (quad_q, lin_q, quad_k, lin_k))
    pad = (-S) % group
    if pad:
        quad_q, quad_k, lin_q, lin_k, v, u = (F.pad(t, (0, 0, 0, pad)) for t in
                                              (quad_q, quad_k, lin_q, lin_k, v, u))
    G = (S + pad) // group
    gq, gk, lq, lk, gv, gu = (t.reshape(B, G, group, t.shape[-1]) for t in
                              (quad_q, quad_k, lin_q, lin_k, v, u))
    sim = torch.matmul(gq, gk.transpose(-1, -2)) / group          # :256
    attn = F.relu(sim) ** 2                                          # :258
    if pad:                                                          # :243-245, :261-262
        keymask = (torch.arange(G * group, device=v.device) < S).reshape(1, G, 1, group)
        attn = attn.masked_fill(~keymask, 0.0)
    quad_v = torch.matmul(attn, gv)                                  # :269
    quad_u = torch.matmul(attn, gu)                                  # :270
    lin_kv = torch.einsum("bgnd,bgne->bde", lk, gv) / S              # :286 (n = unpadded S)
    lin_ku = torch.einsum("bgnd,bgne->bde", lk, gu) / S              # :289
    lin_v = torch.einsum("bgnd,bde->bgne", lq, lin_kv)               # :287
    lin_u = torch.einsum("bgnd,bde->bgne", lq, lin_ku)               # :290
    att_v = (quad_v + lin_v).reshape(B, G * group, -1)[:, :S]        # :293-294
    att_u = (quad_u + lin_u).reshape(B, G * group, -1)[:, :S]
    return att_v, att_u


def flash_layer(x, sd, l: int, taps=None):
    """FLASH_ShareA_FFConvM.forward.  mossformer_block.py:191-220."""
    p = f"{PFX}layers.{l}."
    half = x.shape[-1] // 2
    xs = torch.cat((F.pad(x[..., :half], (0, 0, 1, -1)), x[..., half:]), dim=-1)   # :204-207
    hid = _ffconvm(xs, sd, p + "to_hidden.", "scale")                              # :210
    v, u = hid.chunk(2, dim=-1)
    qk = _ffconvm(xs, sd, p + "to_qk.", "scale")                                   # :211
    gam, bet = sd[p + "qk_offset_scale.gamma"], sd[p + "qk_offset_scale.beta"]     # :76-86
    quad_q, lin_q, quad_k, lin_k = (qk * gam[h] + bet[h] for h in range(4))        # :214
    att_v, att_u = cal_attention(quad_q, lin_q, quad_k, lin_k, v, u,
                                 sd[p + "rotary_pos_emb.freqs"])                   # :215
    out = (att_u * v) * torch.sigmoid(att_v * u)                                   # :217
    if taps is not None and l == 0:
        taps["flash0_v"] = v
        taps["flash0_qk"] = qk
        taps["flash0_att_v"] = att_v
        taps["flash0_gate"] = out
    return x + _ffconvm(out, sd, p + "to_out.", "scale")                           # :219


def dilated_dense_net(p_in, sd, q):
    """DilatedDenseNet(depth=2, lorder=20, in_channels=256) on tokens-major p[B,S,C].
    fsmn.py:76-111.  conv2 is a grouped conv over cat([out1, skip]) with groups=C, so
    output channel j reads concat channels 2j, 2j+1 (fsmn.py:92-98,110)."""
    C = p_in.shape[-1]
    w1 = sd[q + "conv1.weight"].reshape(C, 1, -1)
    w2 = sd[q + "conv2.weight"].reshape(C, 2, -1)
    k = w1.shape[-1]

    def inorm_prelu(t, i):      # InstanceNorm2d(affine) over S per (b,c), biased var, eps 1e-5; PReLU(C)
        mu = t.mean(dim=1, keepdim=True)
        var = t.var(dim=1, unbiased=False, keepdim=True)
        y = (t - mu) / torch.sqrt(var + 1e-5) * sd[q + f"norm{i}.weight"] + sd[q + f"norm{i}.bias"]
        a = sd[q + f"prelu{i}.weight"]
        return torch.where(y >= 0, y, a * y)

    c1 = F.conv1d(p_in.transpose(1, 2), w1, padding=(k - 1) // 2, groups=C).transpose(1, 2)
    o1 = inorm_prelu(c1, 1)
    cat = torch.cat((o1, p_in), dim=-1)                       # channels [o1_0.., p_0..]
    c2 = F.conv1d(cat.transpose(1, 2), w2, padding=(k - 1), dilation=2, groups=C).transpose(1, 2)
    return inorm_prelu(c2, 2)


def fsmn_layer(x, sd, l: int, taps=None):
    """GatedFSMNBlockDilated.forward (mossformer_block.py:419-425) with GatedFSMNDilated
    (:316-325) and UniDeepFsmnDilated (fsmn.py:136-144); CLayerNorm layer_norm.py:9-30."""
    p = f"{PFX}fsmn.{l}."
    C = sd[p + "conv1.0.bias"].shape[0]
    h = F.linear(x, sd[p + "conv1.0.weight"].reshape(C, -1), sd[p + "conv1.0.bias"])
    a = sd[p + "conv1.1.weight"]
    h = torch.where(h >= 0, h, a * h)                                              # PReLU(1)
    h = F.layer_norm(h, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)  # CLayerNorm
    g = p + "gated_fsmn."
    xu = _ffconvm(h, sd, g + "to_u.", "layer")
    xv = _ffconvm(h, sd, g + "to_v.", "layer")
    f1 = F.relu(F.linear(xu, sd[g + "fsmn.linear.weight"], sd[g + "fsmn.linear.bias"]))
    p1 = F.linear(f1, sd[g + "fsmn.project.weight"])
    xu = xu + dilated_dense_net(p1, sd, g + "fsmn.conv.")
    gated = xv * xu + h
    n2 = F.layer_norm(gated, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
    y = F.linear(n2, sd[p + "conv2.weight"].reshape(-1, C), sd[p + "conv2.bias"])
    if taps is not None and l == 0:
        taps["fsmn0_h"] = h
        taps["fsmn0_xu"] = xu
    return y + x


def _group_norm1(x_bcs, w, b, eps=1e-8):
    """nn.GroupNorm(1, C, eps=1e-8) on [B,C,S]: stats over all C*S.  mossformer2.py:143-154."""
    return F.group_norm(x_bcs, 1, w, b, eps)


def masknet(E, sd, num_blocks: int, taps=None):
    """MossFormerMaskNet.forward.  mossformer2.py:470-523 (+ComputationBlock :366-397,
    MossFormerM :309-322, ScaledSinuEmbedding mossformer_block.py:60-73)."""
    B, C, S = E.shape
    z = _group_norm1(E, sd["mask_net.norm.weight"], sd["mask_net.norm.bias"])
    z = F.conv1d(z, sd["mask_net.conv1d_encoder.weight"])
    inv = sd["mask_net.pos_enc.inv_freq"]
    t = torch.arange(S, device=E.device).to(inv.dtype)
    sinu = t[:, None] * inv[None, :]
    emb = torch.cat((sinu.sin(), sinu.cos()), dim=-1) * sd["mask_net.pos_enc.scale"]   # [S,C]
    z = z + emb.transpose(0, 1)
    if taps is not None:
        taps["z"] = z
    h = z.permute(0, 2, 1).contiguous()
    for l in range(num_blocks):
        h = flash_layer(h, sd, l, taps)
        if taps is not None and l == 0:
            taps["after_flash0"] = h
        h = fsmn_layer(h, sd, l, taps)
        if taps is not None and l == 0:
            taps["after_fsmn0"] = h
    if taps is not None:
        taps["after_stack"] = h
    h = F.layer_norm(h, (C,), sd["mask_net.mdl.intra_mdl.norm.weight"],
                     sd["mask_net.mdl.intra_mdl.norm.bias"], 1e-6)
    r = _group_norm1(h.permute(0, 2, 1), sd["mask_net.mdl.intra_norm.weight"],
                     sd["mask_net.mdl.intra_norm.bias"]) + z
    a = sd["mask_net.prelu.weight"]
    r = torch.where(r >= 0, r, a * r)
    r = F.conv1d(r, sd["mask_net.conv1d_out.weight"], sd["mask_net.conv1d_out.bias"])
    nspk = r.shape[1] // C
    r = r.reshape(B * nspk, C, S)
    r = torch.tanh(F.conv1d(r, sd["mask_net.output.0.weight"], sd["mask_net.output.0.bias"])) * \
        torch.sigmoid(F.conv1d(r, sd["mask_net.output_gate.0.weight"], sd["mask_net.output_gate.0.bias"]))
    r = F.relu(F.conv1d(r, sd["mask_net.conv1_decoder.weight"]))
    return r.reshape(B, nspk, C, S).transpose(0, 1)          # [spk,B,C,S]


def mossformer2_forward(wave: torch.Tensor, sd, num_blocks: int | None = None, taps=None):
    """MossFormer2.forward: wave[B,T] -> [B,2,T].  mossformer2.py:563-589
    (Encoder :157-210, Decoder :213-257)."""
    if num_blocks is None:
        num_blocks = 1 + max(int(k.split("layers.")[1].split(".")[0]) for k in sd if ".layers." in k)
    if wave.ndim == 1:
        wave = wave[None]
    if wave.ndim == 3:
        wave = wave.squeeze(1)
    B, T = wave.shape
    k = sd["enc.conv1d.weight"].shape[-1]
    E = F.relu(F.conv1d(wave[:, None, :], sd["enc.conv1d.weight"], stride=k // 2))
    if taps is not None:
        taps["enc"] = E
    M = masknet(E, sd, num_blocks, taps)
    if taps is not None:
        taps["mask"] = M
    outs = []
    for spk in range(M.shape[0]):
        y = F.conv_transpose1d(E * M[spk], sd["dec.weight"], stride=k // 2)[:, 0, :]
        outs.append(y)
    est = torch.stack(outs, dim=1)                            # [B,spk,T_est]
    T_est = est.shape[-1]
    if T > T_est:
        est = F.pad(est, (0, T - T_est))
    else:
        est = est[..., :T]
    return est.contiguous()


def cast_state_dict(sd, dtype):
    return {k: v.to(dtype) for k, v in sd.items()}


# ---------------------------------------------------------------------------------------
# Host-side pieces of the path (python, no arithmetic kernels)
# ---------------------------------------------------------------------------------------
def window_plan(n: int, window: int = 160000, start: int = 0):
    """The [start,end) windows AudioProcessor.separate_speaker feeds the separator, for one
    VAD frame [start, start+n).  AudioProcessor.py:920-935."""
    starts, ends = [], []
    k = n // window
    if k == 0:
        starts.append(start)
        ends.append(start + n)
    else:
        for j in range(k):
            starts.append(start + j * window)
            ends.append(start + (j + 1) * window)
        r = n % window
        if r > 0:
            if r > window / 2:
                starts.append(ends[-1])
                ends.append(start + n)
            else:
                ends[-1] = start + n
    return list(zip(starts, ends))


def cosine_similarity(a, b) -> float:
    """TargetASR.cosine_similarity: 1.0 if either is all-zero, else clip(cos, 0, 1).
    TargetASR.py:144-152."""
    import numpy as np
    a = np.asarray(a)
    b = np.asarray(b)
    if np.all(a == 0.0) or np.all(b == 0.0):
        return 1.0
    s = np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b))
    return float(max(0.0, min(s, 1.0)))
