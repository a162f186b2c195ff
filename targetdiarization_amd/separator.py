"""MossFormer2 separator: Python host over the C-ABI (tdx_mf2_*).

Mirrors the call surface the reference uses at AudioProcessor.py:268-274, 943:
    sep = MossFormer2Separator.from_pretrain(path)      # BaseModel.from_pretrain base_model.py:52-64
    out = sep(wave_tensor[B,T])                          # -> [B,2,T]   (mossformer2.py:563-589)
PyTorch is used only for device memory and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .weights import pack_blob


class MossFormer2Separator:
    def __init__(self, state_dict, device="cuda:0", num_blocks: int | None = None, graph_rows: int = 40000):
        """graph_rows: forwards with B*S <= graph_rows token rows (the reference's own call pattern: ONE window per call) are
        captured once per (B, T) as a HIP graph and replayed — the ~450 launches of a forward are then one submission (at
        S = 17 000 the kernels are 20-100 us each and the launch gaps are a tenth of the time); 0 disables."""
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.TdxError("MossFormer2Separator needs a HIP device (cuda:N); there is no CPU path")
        if num_blocks is None:
            num_blocks = 1 + max(int(k.split("layers.")[1].split(".")[0]) for k in state_dict if ".layers." in k)
        self.num_blocks = num_blocks
        self._l = _lib.lib()
        blob = pack_blob(state_dict)
        cfg = _lib.Mf2Config(num_blocks=num_blocks, channels=512, kernel_size=16, num_spks=2, group_size=256)
        h = C.c_void_p()
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        with torch.cuda.device(idx):
            _lib.check(self._l.tdx_mf2_create(C.byref(cfg), buf, len(blob), idx, C.byref(h)))
        self._h = h
        self._guard = _lib.HandleGuard(self.device)      # calls on this object are serialised (host lock + device event chain)
        self._ws = None
        self._taps = False
        self._profiling = False
        self.graph_rows = graph_rows
        self._graphs = _lib.GraphRunner(self.device)

    @classmethod
    def from_pretrain(cls, path, device="cuda:0", **model_args):
        """Load a look2hear checkpoint dict {"model_name","state_dict",...} (base_model.py:56-63)."""
        conf = torch.load(path, map_location="cpu", weights_only=True)     # never unpickle code from a checkpoint
        sd = conf["state_dict"] if "state_dict" in conf else conf
        name = conf.get("model_name", "MossFormer2") if isinstance(conf, dict) else "MossFormer2"
        if name != "MossFormer2":
            raise _lib.TdxError(f"checkpoint model_name={name!r}: the MI355X build implements MossFormer2 only")
        # config.yaml's `model:` block = constructor kwargs (AudioProcessor.py:268-271); supported: the
        # constructor defaults (mossformer2.py:532-541) with any num_blocks
        defaults = {"in_channels": 512, "out_channels": 512, "kernel_size": 16, "norm": "ln", "num_spks": 2,
                    "skip_around_intra": True, "use_global_pos_enc": True, "max_length": 20000}
        for k, v in model_args.items():
            if k in ("num_blocks", "_target_"):
                continue
            if k not in defaults:
                raise _lib.TdxError(f"unsupported MossFormer2 constructor argument {k!r}")
            if v != defaults[k] and k != "max_length":
                raise _lib.TdxError(f"unsupported MossFormer2 constructor argument {k}={v!r} (supported: {defaults[k]!r})")
        return cls(sd, device=device, num_blocks=model_args.get("num_blocks"))

    def eval(self):
        return self

    def to(self, device):
        if torch.device(device) != self.device:
            raise _lib.TdxError("weights live on %s; re-create the separator for another device" % self.device)
        return self

    def enable_taps(self, on=True):
        _lib.check(self._l.tdx_mf2_enable_taps(self._h, 1 if on else 0))
        self._taps = on

    def profile_enable(self, max_records: int):
        _lib.check(self._l.tdx_mf2_profile_enable(self._h, max_records))
        self._profiling = max_records > 0

    def profile_collect(self):
        """(total ms, launches) of the dominant GEMM since the last collect; sync first."""
        ms = C.c_double(); n = C.c_int()
        _lib.check(self._l.tdx_mf2_profile_collect(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def workspace_bytes(self, B, T):
        return int(self._l.tdx_mf2_workspace_bytes(self._h, B, T))

    def flops(self, B, T):
        return float(self._l.tdx_mf2_flops(self._h, B, T))

    def __call__(self, wav: torch.Tensor) -> torch.Tensor:
        if wav.ndim == 1:
            wav = wav.unsqueeze(0)
        if wav.ndim == 3:
            wav = wav.squeeze(1)
        wav = wav.to(self.device, torch.float32).contiguous()
        B, T = wav.shape
        S = (T - 16) // 8 + 1
        need = self.workspace_bytes(B, T)
        if need == 0:
            raise _lib.TdxError(f"bad shape B={B} T={T}")
        with self._guard.call():
            if self.graph_rows and B * S <= self.graph_rows and not self._taps and not self._profiling:
                out = self._forward_graph(wav, B, T, need)
                if out is not None:
                    return out
            ws = self._guard.workspace(need)
            out = torch.empty(B, 2, T, dtype=torch.float32, device=self.device)
            st = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(self._l.tdx_mf2_forward(self._h, wav.data_ptr(), B, T, out.data_ptr(), ws.data_ptr(), ws.numel(), st))
            self._last = (B, T)
            self._ws = ws
            return out

    def _forward_graph(self, wav, B, T, need):
        """small forwards: HIP-graph replay (_lib.GraphRunner) from the second sighting of a shape on; None = run eager"""
        def launch(si, so, ws, st):
            _lib.check(self._l.tdx_mf2_forward(self._h, si.data_ptr(), B, T, so.data_ptr(), ws.data_ptr(), ws.numel(), st))
        out = self._graphs((B, T), wav, (B, 2, T), need, launch)
        if out is not None:
            self._last = (B, T)
            self._ws = self._graphs._g[(B, T)][3]
        return out

    forward = __call__

    def tap(self, name: str) -> torch.Tensor:
        B, T = self._last
        S = (T - 16) // 8 + 1
        n = 2 * self.num_blocks if name == "headroom" else (2 if name == "mask" else 1) * B * S * 512
        dst = torch.empty(n, dtype=torch.float32, device=self.device)
        cnt = C.c_size_t()
        with self._guard.call():
            st = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(self._l.tdx_mf2_tap(self._h, name.encode(), B, T, self._ws.data_ptr(), dst.data_ptr(), n, C.byref(cnt), st))
        if name == "headroom":        # [L][2]: largest |f16| written under the static scales of (v|u, lin_k); the f16 range ends at 65504
            return dst.view(self.num_blocks, 2)
        return dst.view(2, B, S, 512) if name == "mask" else dst.view(B, S, 512)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._l.tdx_mf2_destroy(self._h)
                self._h = None
        except Exception:
            pass
