"""Stand-alone device ops exported by libtdx.so (test hooks for individual reference functions
and the cosine scorer).  All take/return CUDA(HIP) torch tensors; PyTorch only owns memory."""
from __future__ import annotations

import torch

from . import _lib


def _st(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def linear(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
    """C = A W^T (+bias) through the fp32-MFMA GEMM core (tdx_linear)."""
    a = a.contiguous().float(); w = w.contiguous().float()
    M, K = a.shape; N = w.shape[0]
    c = torch.empty(M, N, device=a.device, dtype=torch.float32)
    _lib.check(_lib.lib().tdx_linear(a.data_ptr(), w.data_ptr(), bias.contiguous().data_ptr() if bias is not None else None,
                                     M, N, K, c.data_ptr(), _st(a)))
    return c


def cal_attention(quad_q, lin_q, quad_k, lin_k, v, u, freqs):
    """mossformer_block.py:222-294 on device (tdx_cal_attention)."""
    l = _lib.lib()
    ts = [t.contiguous().float() for t in (quad_q, lin_q, quad_k, lin_k, v, u, freqs)]
    B, S, E = ts[4].shape
    nb = int(l.tdx_cal_attention_workspace_bytes(B, S, E))
    if nb == 0:
        raise _lib.TdxError("cal_attention: bad shape")
    ws = torch.empty(nb, dtype=torch.uint8, device=v.device)
    av = torch.empty(B, S, E, device=v.device); au = torch.empty(B, S, E, device=v.device)
    _lib.check(l.tdx_cal_attention(*[t.data_ptr() for t in ts], B, S, E, av.data_ptr(), au.data_ptr(), ws.data_ptr(), nb, _st(v)))
    return av, au


def dilated_dense_net(p, w1, w2, in_g, in_b, prelu):
    """fsmn.py:103-111 on device (tdx_dilated_dense_net).  p[B,S,256]; w1[256,39]; w2[256,2,39];
    in_g/in_b/prelu [2,256]."""
    l = _lib.lib()
    ts = [t.contiguous().float() for t in (p, w1, w2, in_g, in_b, prelu)]
    B, S, _ = ts[0].shape
    nb = int(l.tdx_dilated_dense_net_workspace_bytes(B, S))
    ws = torch.empty(nb, dtype=torch.uint8, device=p.device)
    out = torch.empty(B, S, 256, device=p.device)
    _lib.check(l.tdx_dilated_dense_net(ts[0].data_ptr(), B, S, *[t.data_ptr() for t in ts[1:]], out.data_ptr(), ws.data_ptr(), nb, _st(p)))
    return out


def cosine_scores(emb: torch.Tensor, ref: torch.Tensor) -> torch.Tensor:
    """TargetASR.cosine_similarity (TargetASR.py:144-152) for N embeddings vs one reference."""
    emb = emb.contiguous().float(); ref = ref.contiguous().float()
    N, D = emb.shape
    out = torch.empty(N, device=emb.device)
    _lib.check(_lib.lib().tdx_cosine_scores(emb.data_ptr(), ref.data_ptr(), N, D, out.data_ptr(), _st(emb)))
    return out


def loudness(wav: torch.Tensor, rate: int = 16000) -> torch.Tensor:
    """AudioProcessor.meter_loudness (AudioProcessor.py:1123-1127) for B device-resident mono clips [B,N]:
    integrated loudness in LUFS as float64 [B] (unrounded; -inf for silence).  N >= 0.4 s."""
    wav = wav.contiguous().float()
    if wav.ndim == 1:
        wav = wav[None]
    B, N = wav.shape
    l = _lib.lib()
    nb = int(l.tdx_loudness_workspace_bytes(B, N, rate))
    if nb == 0:
        raise ValueError("Audio must have length greater than the block size.")
    ws = torch.empty(nb, dtype=torch.uint8, device=wav.device)
    out = torch.empty(B, dtype=torch.float64, device=wav.device)
    _lib.check(l.tdx_loudness(wav.data_ptr(), B, N, rate, out.data_ptr(), ws.data_ptr(), nb, _st(wav)))
    return out


_RESAMPLE_FILTERS = {}


def resample_filter(up: int, down: int):
    """The FIR of scipy.signal.resample_poly(window=("kaiser", 5.0)): 2*half+1 taps, half = 10*max(up, down), a Kaiser-windowed
    sinc with cutoff 1/max(up, down) of Nyquist, unit DC gain, times `up` (restated with numpy; float64 design, float32 taps)."""
    import numpy as np
    key = (up, down)
    if key not in _RESAMPLE_FILTERS:
        max_rate = max(up, down)
        half = 10 * max_rate
        n = 2 * half + 1
        cutoff = 1.0 / max_rate
        m = np.arange(n, dtype=np.float64) - half
        h = cutoff * np.sinc(cutoff * m) * np.kaiser(n, 5.0)
        h = h / h.sum() * up
        _RESAMPLE_FILTERS[key] = (h.astype(np.float32), half)
    return _RESAMPLE_FILTERS[key]


def resample_poly(x: torch.Tensor, orig_sr: int, target_sr: int) -> torch.Tensor:
    """AudioProcessor.audio_resample (AudioProcessor.py:549-569) on the device: x [C,n] or [n] at orig_sr -> target_sr, rational
    polyphase (up/down = target/orig reduced), scipy.signal.resample_poly's default filter.  The reference calls
    librosa.resample (soxr, third-party): parity unpinned."""
    import math
    if orig_sr == target_sr:
        return x
    g = math.gcd(int(orig_sr), int(target_sr))
    up, down = int(target_sr) // g, int(orig_sr) // g
    one_d = x.ndim == 1
    xx = (x[None] if one_d else x).contiguous().float()
    C_, n_in = xx.shape
    n_out = (n_in * up + down - 1) // down
    h, half = resample_filter(up, down)
    hd = torch.from_numpy(h).to(xx.device)
    y = torch.empty(C_, n_out, device=xx.device)
    _lib.check(_lib.lib().tdx_resample_poly(xx.data_ptr(), n_in, C_, up, down, hd.data_ptr(), half, y.data_ptr(), n_out, _st(xx)))
    return y[0] if one_d else y
