"""Spectral front-ends over the C-ABI: Kaldi fbank (speaker / ASR variants), LFR+CMVN and the
MDX block STFT/iSTFT (ConvTDFNet.stft/.istft, AudioProcessor.py:82-120)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _st(dev):
    return torch.cuda.current_stream(dev).cuda_stream


class Fbank:
    """mode "sv": povey window, no scaling, per-utterance mean removed (ERes2NetV2 front-end);
    mode "asr": hamming window, input x32768 (funasr WavFrontend before LFR/CMVN)."""

    def __init__(self, mode: str = "sv", device="cuda:0"):
        self.device = torch.device(device)
        self._l = _lib.lib()
        h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        _lib.check(self._l.tdx_fbank_create({"sv": 0, "asr": 1}[mode], idx, C.byref(h)))
        self._h = h

    def frames(self, N: int) -> int:
        return int(self._l.tdx_fbank_frames(N))

    def __call__(self, wav: torch.Tensor) -> torch.Tensor:
        if wav.ndim == 1:
            wav = wav[None]
        wav = wav.to(self.device, torch.float32).contiguous()
        B, N = wav.shape
        F = self.frames(N)
        if F < 1:
            raise _lib.TdxError("fbank needs >= 400 samples")
        nb = int(self._l.tdx_fbank_workspace_bytes(B, N))
        ws = torch.empty(nb, dtype=torch.uint8, device=self.device)
        out = torch.empty(B, F, 80, device=self.device)
        _lib.check(self._l.tdx_fbank_forward(self._h, wav.data_ptr(), B, N, out.data_ptr(), ws.data_ptr(), nb, _st(self.device)))
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._l.tdx_fbank_destroy(self._h); self._h = None
        except Exception:
            pass


def lfr_cmvn(feat: torch.Tensor, shift: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    feat = feat.contiguous().float()
    B, F, _ = feat.shape
    out = torch.empty(B, (F + 5) // 6, 560, device=feat.device)
    _lib.check(_lib.lib().tdx_lfr_cmvn(feat.data_ptr(), B, F, shift.contiguous().float().data_ptr(),
                                       scale.contiguous().float().data_ptr(), out.data_ptr(), _st(feat.device)))
    return out


class BlockSTFT:
    """ConvTDFNet's stft/istft pair (AudioProcessor.py:65-120): dim_t is the number of frames
    (the reference passes dim_t=8 meaning 2**8)."""

    def __init__(self, n_fft=6144, hop=2048, dim_f=3072, dim_t=256, device="cuda:0"):
        self.device = torch.device(device)
        self._l = _lib.lib()
        h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        _lib.check(self._l.tdx_stft_create(n_fft, hop, dim_f, dim_t, idx, C.byref(h)))
        self._h = h
        self.n_fft, self.hop, self.dim_f, self.dim_t = n_fft, hop, dim_f, dim_t
        self.chunk_size = int(self._l.tdx_stft_chunk_size(h))
        self.dim_c = 4

    def _ws(self, R):
        nb = int(self._l.tdx_stft_workspace_bytes(self._h, R))
        return torch.empty(nb, dtype=torch.uint8, device=self.device), nb

    def stft(self, x: torch.Tensor) -> torch.Tensor:
        """x [..., chunk] (reference: [n_blk, 2, chunk]) -> [n_blk, 4, dim_f, dim_t]"""
        x = x.to(self.device, torch.float32).reshape(-1, self.chunk_size).contiguous()
        R = x.shape[0]
        ws, nb = self._ws(R)
        spec = torch.empty(R, 2, self.dim_f, self.dim_t, device=self.device)
        _lib.check(self._l.tdx_stft_forward(self._h, x.data_ptr(), R, spec.data_ptr(), ws.data_ptr(), nb, _st(self.device)))
        return spec.reshape(-1, 4, self.dim_f, self.dim_t)

    def istft(self, spec: torch.Tensor) -> torch.Tensor:
        """spec [n_blk, 4, dim_f, dim_t] -> [n_blk, 2, chunk]"""
        spec = spec.to(self.device, torch.float32).reshape(-1, 2, self.dim_f, self.dim_t).contiguous()
        R = spec.shape[0]
        ws, nb = self._ws(R)
        y = torch.empty(R, self.chunk_size, device=self.device)
        _lib.check(self._l.tdx_stft_inverse(self._h, spec.data_ptr(), R, y.data_ptr(), ws.data_ptr(), nb, _st(self.device)))
        return y.reshape(-1, 2, self.chunk_size)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._l.tdx_stft_destroy(self._h); self._h = None
        except Exception:
            pass
