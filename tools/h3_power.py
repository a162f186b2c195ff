"""x3 core: same launch on random operands and on all-zero operands (no MFMA data toggling -> lower power): separates the
power/clock limit from structural stalls (diagnostic)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from targetdiarization_amd import _lib
from tools.h3_test import split, dev
l = _lib.diag()
gv = l.tdx_h3_gemm_variant; gv.restype = C.c_int
gv.argtypes = [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_void_p]
for (m, n, k) in [(65536, 2048, 2048), (255968, 2176, 512)]:
    bias = torch.zeros(n, device=dev); c = torch.empty(m, n, device=dev)
    for label in ("random", "zeros"):
        a = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev)
        pa, sa = split(a); pb, sb = split(w)
        if label == "zeros":
            pa.zero_(); pb.zero_()
        for v in (0, 9, 1, 2, 3):
            for _ in range(2):
                gv(pa.data_ptr(), sa.data_ptr(), pb.data_ptr(), sb.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, v, None)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                gv(pa.data_ptr(), sa.data_ptr(), pb.data_ptr(), sb.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, v, None)
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 10
            print(f"M={m} N={n} K={k} {label:6s} variant {v}: {t*1e3:.1f} us {2.0*m*n*k/t/1e9:.1f} TF", flush=True)
        del a, w, pa, pb
