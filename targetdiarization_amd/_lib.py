"""ctypes binding of libtdx.so (include/tdx.h).  Fails loudly when the library is missing:
there is no CPU / PyTorch fallback anywhere in the product path."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libtdx.so")


class TdxError(RuntimeError):
    pass


class Mf2Config(C.Structure):
    _fields_ = [("num_blocks", C.c_int32), ("channels", C.c_int32), ("kernel_size", C.c_int32),
                ("num_spks", C.c_int32), ("group_size", C.c_int32), ("reserved", C.c_int32 * 3)]


_lib = None

# every symbol include/tdx.h declares: (restype, argtypes)
_vp, _sz, _i, _fp = C.c_void_p, C.c_size_t, C.c_int, C.c_void_p
SIGNATURES = {
    "tdx_version": (C.c_char_p, []),
    "tdx_last_error": (C.c_char_p, []),
    "tdx_mf2_create": (_i, [C.POINTER(Mf2Config), _vp, _sz, _i, C.POINTER(_vp)]),
    "tdx_mf2_destroy": (_i, [_vp]),
    "tdx_mf2_enable_taps": (_i, [_vp, _i]),
    "tdx_mf2_profile_enable": (_i, [_vp, _i]),
    "tdx_mf2_profile_collect": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(_i)]),
    "tdx_mf2_workspace_bytes": (_sz, [_vp, _i, _i]),
    "tdx_mf2_forward": (_i, [_vp, _fp, _i, _i, _fp, _vp, _sz, _vp]),
    "tdx_mf2_flops": (C.c_double, [_vp, _i, _i]),
    "tdx_mf2_tap": (_i, [_vp, C.c_char_p, _i, _i, _vp, _fp, _sz, C.POINTER(_sz), _vp]),
    "tdx_cal_attention": (_i, [_fp] * 7 + [_i, _i, _i, _fp, _fp, _vp, _sz, _vp]),
    "tdx_cal_attention_workspace_bytes": (_sz, [_i, _i, _i]),
    "tdx_dilated_dense_net": (_i, [_fp, _i, _i, _fp, _fp, _fp, _fp, _fp, _fp, _vp, _sz, _vp]),
    "tdx_dilated_dense_net_workspace_bytes": (_sz, [_i, _i]),
    "tdx_linear": (_i, [_fp, _fp, _fp, _i, _i, _i, _fp, _vp]),
    "tdx_fbank_create": (_i, [_i, _i, C.POINTER(_vp)]),
    "tdx_fbank_destroy": (_i, [_vp]),
    "tdx_fbank_frames": (_i, [_i]),
    "tdx_fbank_workspace_bytes": (_sz, [_i, _i]),
    "tdx_fbank_forward": (_i, [_vp, _fp, _i, _i, _fp, _vp, _sz, _vp]),
    "tdx_lfr_cmvn": (_i, [_fp, _i, _i, _fp, _fp, _fp, _vp]),
    "tdx_stft_create": (_i, [_i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "tdx_stft_destroy": (_i, [_vp]),
    "tdx_stft_chunk_size": (_i, [_vp]),
    "tdx_stft_workspace_bytes": (_sz, [_vp, _i]),
    "tdx_stft_forward": (_i, [_vp, _fp, _i, _fp, _vp, _sz, _vp]),
    "tdx_stft_inverse": (_i, [_vp, _fp, _i, _fp, _vp, _sz, _vp]),
    "tdx_pfenc_create": (_i, [_i, _vp, _sz, _i, C.POINTER(_vp)]),
    "tdx_pfenc_destroy": (_i, [_vp]),
    "tdx_pfenc_workspace_bytes": (_sz, [_vp, _i, _i]),
    "tdx_pfenc_flops": (C.c_double, [_vp, _i, _i]),
    "tdx_pfenc_forward": (_i, [_vp, _fp, _vp, _i, _i, _fp, _vp, _sz, _vp]),
    "tdx_pfdec_create": (_i, [_i, _i, _vp, _sz, _i, C.POINTER(_vp)]),
    "tdx_pfdec_destroy": (_i, [_vp]),
    "tdx_pfdec_predict_workspace_bytes": (_sz, [_vp, _i, _i]),
    "tdx_pfdec_predict": (_i, [_vp, _fp, _i, _i, _fp, _fp, _vp, _vp, _vp, _sz, _vp]),
    "tdx_pfdec_decode_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "tdx_pfdec_decode": (_i, [_vp, _fp, _i, _vp, _fp, _i, _i, _i, _vp, _fp, _vp, _sz, _vp]),
    "tdx_eres2net_create": (_i, [_vp, _sz, _i, C.POINTER(_vp)]),
    "tdx_eres2net_destroy": (_i, [_vp]),
    "tdx_eres2net_workspace_bytes": (_sz, [_vp, _i, _i]),
    "tdx_eres2net_flops": (C.c_double, [_vp, _i, _i]),
    "tdx_eres2net_forward": (_i, [_vp, _fp, _i, _i, _fp, _vp, _sz, _vp]),
    "tdx_cosine_scores": (_i, [_fp, _fp, _i, _i, _fp, _vp]),
    "tdx_loudness_workspace_bytes": (_sz, [_i, C.c_long, _i]),
    "tdx_loudness": (_i, [_fp, _i, C.c_long, _i, _vp, _vp, _sz, _vp]),
    "tdx_resample_poly": (_i, [_fp, C.c_long, _i, _i, _i, _fp, _i, _fp, C.c_long, _vp]),
}


def lib():
    """Load libtdx.so once; raise TdxError if it is absent (never fall back)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TdxError(f"{LIB_PATH} not found: build it with `python -m targetdiarization_amd.build` "
                           "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        # One HIP runtime per process: libtdx.so NEEDs libamdhip64.so.7 by SONAME.  PyTorch-ROCm
        # ships its own copy with that SONAME; device pointers and streams handed across the
        # C-ABI belong to torch's runtime, so it must be the one libtdx binds to.  Importing
        # torch first guarantees that (loading libtdx first would pull in /opt/rocm's copy and
        # leave the process with two runtimes -> "no ROCm-capable device").
        import torch  # noqa: F401
        tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(tl):
            C.CDLL(tl, mode=C.RTLD_GLOBAL)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


# every symbol include/tdx_test.h declares (libtdx_diag.so: test hooks, never used by the product path)
_l = C.c_long
DIAG_SIGNATURES = {
    "tdx_diag_last_error": (C.c_char_p, []),
    "tdx_h3_split_rows": (_i, [_vp, _l, _vp, _vp, _l, _i, _vp]),
    "tdx_h3_split_kmajor": (_i, [_vp, _l, _vp, _l, _i, C.c_float, _vp]),
    "tdx_h3_gemm_x": (_i, [_i] + [_vp] * 5 + [_i] * 3 + [_vp]),
    "tdx_h3_gemm": (_i, [_vp] * 6 + [_i] * 3 + [_vp]),
    "tdx_h3_gemm_variant": (_i, [_vp] * 6 + [_i] * 4 + [_vp]),
    "tdx_linear_variant": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp]),
    "tdx_fill_bench": (_i, [_i, _vp, _l, _i, _i, _vp, _vp]),
    "tdx_fill_bench2": (_i, [_i, _vp, _i, _i, _i, _vp, _vp]),
    "tdx_fill_bench3": (_i, [_vp, _l, _i, _l, _i, _i, _i, _vp, _vp]),
}
DIAG_PATH = os.path.join(HERE, "libtdx_diag.so")
_diag = None


def diag():
    """Load libtdx_diag.so (tests / tools only)."""
    global _diag
    if _diag is None:
        if not os.path.exists(DIAG_PATH):
            raise TdxError(f"{DIAG_PATH} not found: build it with `python -m targetdiarization_amd.build`")
        lib()                                   # torch's HIP runtime first (see lib())
        l = C.CDLL(DIAG_PATH)
        for name, (res, args) in DIAG_SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _diag = l
    return _diag


def check(status: int):
    if status != 0:
        raise TdxError(f"libtdx status {status}: {lib().tdx_last_error().decode()}")


class GraphRunner:
    """HIP-graph replay of a C-ABI forward for SMALL problems (the reference's own call pattern is one clip per call: a
    forward is then hundreds of 10-100 us kernels and the launch gaps are a sizeable part of the latency).  The forwards
    allocate nothing, never synchronise and issue only kernel launches on the caller's stream (include/tdx.h), so a launch
    sequence is captured once per key (shape) with its own static input / output / workspace and replayed.
    (A hipMemsetAsync inside a captured sequence broke the node order on ROCm 7.2: the forwards contain none.)"""

    def __init__(self, device, max_entries: int = 8):
        self.device = device
        self.max_entries = max_entries
        self._g = {}

    def __call__(self, key, x, out_shape, ws_bytes: int, launch):
        """launch(in_tensor, out_tensor, workspace_tensor, stream_handle) issues the forward; returns a fresh output tensor"""
        import torch
        ent = self._g.get(key)
        if ent is None:
            if len(self._g) >= self.max_entries:
                self._g.pop(next(iter(self._g)))
            si = torch.empty_like(x)
            so = torch.empty(out_shape, dtype=torch.float32, device=self.device)
            ws = torch.empty(max(int(ws_bytes), 16), dtype=torch.uint8, device=self.device)
            si.copy_(x)

            def run():
                launch(si, so, ws, torch.cuda.current_stream(self.device).cuda_stream)
            side = torch.cuda.Stream(self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                run()                                  # warm-up outside the capture (function attributes, first-use state)
            torch.cuda.current_stream(self.device).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                run()
            ent = self._g[key] = (g, si, so, ws)
        g, si, so, ws = ent
        si.copy_(x)
        g.replay()
        return so.clone()
