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


class MdxConfig(C.Structure):
    _fields_ = [("num_blocks", C.c_int32), ("l", C.c_int32), ("g", C.c_int32), ("k", C.c_int32), ("bn", C.c_int32),
                ("dim_f", C.c_int32), ("dim_t", C.c_int32), ("reserved", C.c_int32)]


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
    "tdx_mdx_create": (_i, [C.POINTER(MdxConfig), _vp, _sz, _i, C.POINTER(_vp)]),
    "tdx_mdx_destroy": (_i, [_vp]),
    "tdx_mdx_workspace_bytes": (_sz, [_vp, _i]),
    "tdx_mdx_flops": (C.c_double, [_vp, _i]),
    "tdx_mdx_forward": (_i, [_vp, _fp, _i, _fp, _vp, _sz, _vp]),
    "tdx_punc_create": (_i, [_i, _i, _i, _vp, _sz, _i, C.POINTER(_vp)]),
    "tdx_punc_destroy": (_i, [_vp]),
    "tdx_punc_workspace_bytes": (_sz, [_vp, _i, _i]),
    "tdx_punc_forward": (_i, [_vp, _vp, _vp, _i, _i, _fp, _vp, _sz, _vp]),
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


class HandleGuard:
    """Serialises the calls on ONE model object (one tdx handle, its grow-only workspace and its graph-replay buffers) — the
    threading contract of include/tdx.h seen from the host side.  The C handles are re-entrant per (workspace, stream) pair; the
    Python wrappers share one workspace per object, so they serialise instead:
      * host side: a re-entrant lock held while a forward is enqueued (two threads on one model: one after the other);
      * device side: an event chain — every call makes ITS stream wait for the event recorded behind the previous call,
        whatever stream that one ran on, so the shared workspace / static graph buffers never serve two forwards at a time
        (the reference serves one shared model from a REST handler and a WebSocket worker thread, main.py:42,366-367).
    Calls on DIFFERENT model objects (separator || target-clip embedding, embeddings || Paraformer) still overlap.
    Inside a stream capture nothing is waited for or recorded (an event from outside a capture cannot be waited on there);
    the capturing caller holds the lock for the whole capture."""

    def __init__(self, device):
        import threading
        self.device = device
        self.lock = threading.RLock()
        self._event = None
        self._ws = None

    def call(self):
        return _GuardedCall(self)

    def workspace(self, nbytes: int):
        """the grow-only workspace (use inside `with guard.call():`).  Before a smaller buffer is dropped the host waits for its
        last user: the caching allocator would hand the block back to the stream that allocated it, which has not waited."""
        import torch
        nbytes = max(int(nbytes), 16)
        if self._ws is None or self._ws.numel() < nbytes:
            if self._ws is not None and self._event is not None:
                self._event.synchronize()
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws


class _GuardedCall:
    def __init__(self, g):
        self.g = g

    def __enter__(self):
        import torch
        g = self.g
        g.lock.acquire()
        try:
            self.stream = torch.cuda.current_stream(g.device)
            self.capturing = torch.cuda.is_current_stream_capturing()
            if g._event is not None and not self.capturing:
                self.stream.wait_event(g._event)
        except BaseException:
            g.lock.release()
            raise
        return g

    def __exit__(self, *exc):
        import torch
        g = self.g
        try:
            if not self.capturing:
                if g._event is None:
                    g._event = torch.cuda.Event()
                g._event.record(self.stream)      # (re-recording is safe: a wait binds to the record that preceded it)
        finally:
            g.lock.release()
        return False


class GraphRunner:
    """HIP-graph replay of a C-ABI forward for SMALL problems (the reference's own call pattern is one clip per call: a
    forward is then hundreds of 10-100 us kernels and the launch gaps are a sizeable part of the latency).  The forwards
    allocate nothing, never synchronise and issue only kernel launches on the caller's stream (include/tdx.h), so a launch
    sequence is captured once per key (shape) with its own static input / output / workspace and replayed.
    (A hipMemsetAsync inside a captured sequence broke the node order on ROCm 7.2: the forwards contain none.)

    A key is captured when it is seen the SECOND time (`capture_after`): a capture costs a warm-up forward + the capture + the
    first replay, which a shape that never comes back (streaming buffers: a new T on almost every call) would pay for nothing —
    `__call__` returns None for such a call and the caller runs the plain eager forward.  Entries are evicted least recently
    used first; before an entry's static buffers are dropped the host waits for its last replay (they were allocated on one
    stream and replayed on others).  Callers serialise their use of one runner (HandleGuard)."""

    def __init__(self, device, max_entries: int = 8, capture_after: int = 2):
        from collections import OrderedDict
        self.device = device
        self.max_entries = max_entries
        self.capture_after = capture_after
        self._g = OrderedDict()
        self._seen = OrderedDict()
        self.captures = 0

    def _evict_one(self):
        _, (g, si, so, ws, last) = self._g.popitem(last=False)
        last.synchronize()

    def __call__(self, key, x, out_shape, ws_bytes: int, launch):
        """launch(in_tensor, out_tensor, workspace_tensor, stream_handle) issues the forward.  Returns a fresh output tensor, or
        None when the key has not been seen `capture_after` times yet (run eager)."""
        import torch
        ent = self._g.get(key)
        if ent is None:
            n = self._seen.pop(key, 0) + 1
            if n < self.capture_after:
                self._seen[key] = n
                while len(self._seen) > 256:
                    self._seen.popitem(last=False)
                return None
            while len(self._g) >= self.max_entries:
                self._evict_one()
            si = torch.empty_like(x)
            so = torch.empty(out_shape, dtype=torch.float32, device=self.device)
            ws = torch.empty(max(int(ws_bytes), 16), dtype=torch.uint8, device=self.device)
            si.copy_(x)
            main = torch.cuda.current_stream(self.device)
            side = torch.cuda.Stream(self.device)

            def run():
                launch(si, so, ws, torch.cuda.current_stream(self.device).cuda_stream)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                run()                                  # warm-up outside the capture, ON THE CAPTURE STREAM (function attributes, first-use
                                                       # state, the separator's per-stream fork/join context: csrc/mf2.hip side_for)
            g = torch.cuda.CUDAGraph()
            # thread_local: another thread's GPU work (a second session on another model object) must not fail this capture
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                run()
            main.wait_stream(side)
            ent = self._g[key] = [g, si, so, ws, torch.cuda.Event()]
            self.captures += 1
        else:
            self._g.move_to_end(key)
        g, si, so, ws, last = ent
        si.copy_(x)
        g.replay()
        out = so.clone()
        last.record(torch.cuda.current_stream(self.device))
        return out
