"""timing-only variants of the x3 core (diagnostic)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from targetdiarization_amd import _lib
from tools.h3_test import split, dev
l = _lib.diag()
gv = l.tdx_h3_gemm_variant; gv.restype = C.c_int
gv.argtypes = [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_void_p]
for (m, n, k) in [(256, 256, 512), (65536, 2048, 2048), (255968, 2176, 512)]:
    a = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev); bias = torch.zeros(n, device=dev); c = torch.empty(m, n, device=dev)
    pa, sa = split(a); pb, sb = split(w)
    for v in (0, 1, 2, 3, 4, 9, 0):
        for _ in range(2):
            gv(pa.data_ptr(), sa.data_ptr(), pb.data_ptr(), sb.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, v, None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            gv(pa.data_ptr(), sa.data_ptr(), pb.data_ptr(), sb.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, v, None)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10
        print(f"M={m} N={n} K={k} variant {v}: {t*1e3:.1f} us {2.0*m*n*k/t/1e9:.1f} TF", flush=True)
