"""MDX-Net denoiser body over the C-ABI (tdx_mdx_*): the network the reference runs as an ONNX file between ConvTDFNet.stft and
.istft (`self.mdx_model.run(None, {"input": mix_spec})`, AudioProcessor.py:224-241, :630).  Architecture: KUIELab ConvTDFNet
(TFC-TDF v2), third-party — parity unpinned (oracle/mdx_oracle.py restates the published forward).

    body = ConvTDFNetBody(state_dict, device="cuda:0")          # module names of the PyTorch class, BatchNorm2d (eval)
    ap = AudioProcessor(is_denoise_vocal=True, mdx_model=body, ...)   # the `mdx_model(spec) -> spec` plug-in of denoise_vocal

Real UVR weights live in *.onnx files: export their initializers under the PyTorch names (first_conv.0.weight, ...) to load them;
no checkpoint ships with the reference, tests and benches use recipe weights (weights.recipe_mdx_state_dict)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .weights import pack_blob


class ConvTDFNetBody:
    def __init__(self, state_dict, device="cuda:0", L: int = 11, l: int = 3, g: int = 32, k: int = 3, bn: int = 8, dim_f: int = 3072,
                 dim_t: int = 256, max_blocks_per_launch: int = 8):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.TdxError("ConvTDFNetBody needs a HIP device")
        self._l = _lib.lib()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        self.dim_f, self.dim_t = dim_f, dim_t
        self.max_blocks_per_launch = max_blocks_per_launch
        cfg = _lib.MdxConfig(num_blocks=L, l=l, g=g, k=k, bn=bn, dim_f=dim_f, dim_t=dim_t)
        blob = pack_blob(state_dict)
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        h = C.c_void_p()
        with torch.cuda.device(idx):
            _lib.check(self._l.tdx_mdx_create(C.byref(cfg), buf, len(blob), idx, C.byref(h)))
        self._h = h
        self._guard = _lib.HandleGuard(self.device)

    def flops(self, B: int) -> float:
        return float(self._l.tdx_mdx_flops(self._h, B))

    def __call__(self, spec: torch.Tensor) -> torch.Tensor:
        """spec [B, 4, dim_f, dim_t] -> [B, 4, dim_f, dim_t]"""
        spec = spec.to(self.device, torch.float32).contiguous()
        if spec.ndim != 4 or spec.shape[1] != 4 or spec.shape[2] != self.dim_f or spec.shape[3] != self.dim_t:
            raise _lib.TdxError(f"ConvTDFNetBody: expected [B, 4, {self.dim_f}, {self.dim_t}], got {tuple(spec.shape)}")
        out = torch.empty_like(spec)
        step = max(1, self.max_blocks_per_launch)
        with torch.cuda.device(self.device), self._guard.call():
            for a in range(0, spec.shape[0], step):
                x = spec[a:a + step]
                nb = int(self._l.tdx_mdx_workspace_bytes(self._h, x.shape[0]))
                ws = self._guard.workspace(nb)
                st = torch.cuda.current_stream(self.device).cuda_stream
                _lib.check(self._l.tdx_mdx_forward(self._h, x.data_ptr(), x.shape[0], out[a:a + step].data_ptr(), ws.data_ptr(), ws.numel(), st))
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._l.tdx_mdx_destroy(self._h); self._h = None
        except Exception:
            pass
