"""Paraformer-large over the C-ABI: SANM encoder (tdx_pfenc_*) plus the ASR feature front-end, and the CIF predictor +
non-autoregressive SANM decoder (tdx_pfdec_*, SURVEY "next" row N2).  Together they replace the neural forward inside
funasr's `AutoModel.generate` (ASRProcessor.py:424); funasr's VAD segmentation, tokenizer files and punctuation stay outside."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .frontend import Fbank, lfr_cmvn
from .weights import pack_blob


class ParaformerEncoder:
    def __init__(self, state_dict, device="cuda:0", num_blocks: int | None = None, cmvn_shift=None, cmvn_scale=None, graph_rows: int = 2048):
        """graph_rows: forwards with B*T <= graph_rows LFR frames are replayed as HIP graphs (_lib.GraphRunner); 0 disables"""
        self.device = torch.device(device)
        self.graph_rows = graph_rows
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
        self._guard = _lib.HandleGuard(self.device)      # calls on this object are serialised (host lock + device event chain)
        self._graphs = _lib.GraphRunner(self.device)
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
        with self._guard.call():
            if self.graph_rows and B * T <= self.graph_rows:
                def launch(si, so, ws, st):
                    _lib.check(self._l.tdx_pfenc_forward(self._h, si.data_ptr(), None, B, T, so.data_ptr(), ws.data_ptr(), ws.numel(), st))
                out = self._graphs((B, T), feats, (B, T, 512), nb, launch)    # None until the shape is seen a second time
                if out is not None:
                    return out
            ws = self._guard.workspace(nb)
            out = torch.empty(B, T, 512, device=self.device)
            st = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(self._l.tdx_pfenc_forward(self._h, feats.data_ptr(), None, B, T, out.data_ptr(), ws.data_ptr(), ws.numel(), st))
            return out

    def __call__(self, wav: torch.Tensor) -> torch.Tensor:
        return self.encode(self.features(wav))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._l.tdx_pfenc_destroy(self._h); self._h = None
        except Exception:
            pass


class ParaformerDecoder:
    """CIF predictor + NAR SANM decoder (funasr CifPredictorV2 + ParaformerSANMDecoder, third-party: parity unpinned).
    `decode(enc)` returns, per utterance, {"token_ids", "scores", "timestamp": [[start_ms, end_ms], ...]}.
    Timestamps: the published Paraformer has none of its own (the reference's "-vad-punc" bundle adds an upsampling
    predictor for them); here token k spans from the encoder frame after the previous integrate-and-fire peak to its own
    peak, in 60 ms LFR frames — an own definition, documented in INTEGRATION.md."""

    FRAME_MS = 60.0

    def __init__(self, state_dict, device="cuda:0", num_blocks: int | None = None, vocab: int | None = None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.TdxError("ParaformerDecoder needs a HIP device")
        if num_blocks is None:
            num_blocks = 1 + max(int(k.split("decoders.")[1].split(".")[0]) for k in state_dict if "decoder.decoders." in k)
        if vocab is None:
            vocab = int(state_dict["decoder.output_layer.bias"].shape[0])
        self.num_blocks, self.vocab = num_blocks, vocab
        self._l = _lib.lib()
        keep = {k: v for k, v in state_dict.items() if k.startswith(("predictor.cif_", "decoder.decoders", "decoder.after_norm", "decoder.output_layer"))}
        blob = pack_blob(keep)
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        _lib.check(self._l.tdx_pfdec_create(num_blocks, vocab, buf, len(blob), idx, C.byref(h)))
        self._h = h
        self._guard = _lib.HandleGuard(self.device)      # calls on this object are serialised (host lock + device event chain)

    def predict(self, enc: torch.Tensor):
        """enc [B,T,512] -> (alphas [B,T+1], embeds [B,T+1,512], counts int32 [B], peaks int32 [B,T+1]) on the device"""
        enc = enc.to(self.device, torch.float32).contiguous()
        B, T, _ = enc.shape
        alphas = torch.empty(B, T + 1, device=self.device)
        emb = torch.empty(B, T + 1, 512, device=self.device)
        counts = torch.empty(B, dtype=torch.int32, device=self.device)
        peaks = torch.empty(B, T + 1, dtype=torch.int32, device=self.device)
        nb = int(self._l.tdx_pfdec_predict_workspace_bytes(self._h, B, T))
        with self._guard.call():
            ws = self._guard.workspace(nb)
            st = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(self._l.tdx_pfdec_predict(self._h, enc.data_ptr(), B, T, alphas.data_ptr(), emb.data_ptr(), counts.data_ptr(), peaks.data_ptr(),
                                                 ws.data_ptr(), ws.numel(), st))
        return alphas, emb, counts, peaks

    def decode_embeds(self, emb: torch.Tensor, counts: torch.Tensor, enc: torch.Tensor, L: int):
        """the decoder alone: emb [B,R,512] (first L rows used), counts int32 [B], enc [B,T,512] -> (ids int32 [B,L], scores [B,L])"""
        enc = enc.to(self.device, torch.float32).contiguous()
        emb = emb.to(self.device, torch.float32).contiguous()
        B, T, _ = enc.shape
        ids = torch.empty(B, L, dtype=torch.int32, device=self.device)
        score = torch.empty(B, L, device=self.device)
        nb = int(self._l.tdx_pfdec_decode_workspace_bytes(self._h, B, L, T))
        with self._guard.call():
            ws = self._guard.workspace(nb)
            st = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(self._l.tdx_pfdec_decode(self._h, emb.data_ptr(), emb.shape[1], counts.data_ptr(), enc.data_ptr(), B, L, T, ids.data_ptr(),
                                                score.data_ptr(), ws.data_ptr(), ws.numel(), st))
        return ids, score

    def decode(self, enc: torch.Tensor):
        """enc [B,T,512] (equal-length segments of a batch) -> list of per-utterance results"""
        alphas, emb, counts, peaks = self.predict(enc)
        cnt = counts.cpu().tolist()                        # the one host read: the decoder's row count is data dependent
        L = max(cnt) if cnt else 0
        B = enc.shape[0]
        if L < 1:
            return [{"token_ids": [], "scores": [], "timestamp": []} for _ in range(B)]
        ids, score = self.decode_embeds(emb, counts, enc, L)
        ids_h, score_h, peaks_h = ids.cpu().numpy(), score.cpu().numpy(), peaks[:, :L].cpu().numpy()
        out = []
        for b in range(B):
            n = cnt[b]
            ts, prev = [], -1
            for k in range(n):
                pk = int(peaks_h[b, k])
                if pk < 0:                                 # fewer fires than floor(sum alphas): the reference pads those frames with zeros
                    pk = prev
                ts.append([int(round((prev + 1) * self.FRAME_MS)), int(round((pk + 1) * self.FRAME_MS))])
                prev = pk
            out.append({"token_ids": [int(t) for t in ids_h[b, :n]], "scores": [float(s) for s in score_h[b, :n]], "timestamp": ts})
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._l.tdx_pfdec_destroy(self._h); self._h = None
        except Exception:
            pass
