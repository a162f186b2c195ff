"""Paraformer-large SANM encoder over the C-ABI (tdx_pfenc_*) plus the ASR feature front-end.
Replaces the neural forward inside funasr's `AutoModel.generate` (ASRProcessor.py:424) up to
the encoder output; CIF predictor / decoder / tokenizer are SURVEY "next" row N2."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .frontend import Fbank, lfr_cmvn
from .weights import pack_blob


class ParaformerEncoder:
    def __init__(self, state_dict, device="cuda:0", num_blocks: int | None = None, cmvn_shift=None, cmvn_scale=None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.TdxError("ParaformerEncoder needs a HIP device")
        if num_blocks is None:
            num_blocks = 2 + max(int(k.split("encoders.")[1].split(".")[0]) for k in state_dict if ".encoders." in k)
        self.num_blocks = num_blocks
        self._l = _lib.lib()
        blob = pack_blob(state_dict)
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        _lib.check(self._l.tdx_pfenc_create(num_blocks, buf, len(blob), idx, C.byref(h)))
        self._h = h
        self._ws = None
        self.fbank = Fbank("asr", self.device)
        # am.mvn vectors (funasr WavFrontend.apply_cmvn): (x + shift) * scale; identity if absent
        self.cmvn_shift = (cmvn_shift if cmvn_shift is not None else torch.zeros(560)).to(self.device).float()
        self.cmvn_scale = (cmvn_scale if cmvn_scale is not None else torch.ones(560)).to(self.device).float()

    def flops(self, B, T):
        return float(self._l.tdx_pfenc_flops(self._h, B, T))

    def features(self, wav: torch.Tensor) -> torch.Tensor:
        """wav [B,N] in [-1,1] -> LFR+CMVN features [B,ceil(F/6),560]"""
        return lfr_cmvn(self.fbank(wav), self.cmvn_shift, self.cmvn_scale)

    def encode(self, feats: torch.Tensor) -> torch.Tensor:
        feats = feats.to(self.device, torch.float32).contiguous()
        B, T, _ = feats.shape
        nb = int(self._l.tdx_pfenc_workspace_bytes(self._h, B, T))
        if self._ws is None or self._ws.numel() < nb:
            self._ws = torch.empty(nb, dtype=torch.uint8, device=self.device)
        out = torch.empty(B, T, 512, device=self.device)
        st = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._l.tdx_pfenc_forward(self._h, feats.data_ptr(), None, B, T, out.data_ptr(), self._ws.data_ptr(), self._ws.numel(), st))
        return out

    def __call__(self, wav: torch.Tensor) -> torch.Tensor:
        return self.encode(self.features(wav))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._l.tdx_pfenc_destroy(self._h); self._h = None
        except Exception:
            pass
