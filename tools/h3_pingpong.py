"""ping-pong main loop of the x3 core (VARIANT 5) against the product loop: correctness and time (diagnostic)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from targetdiarization_amd import _lib
from tools.h3_test import split, dev
l = _lib.diag()
gv = l.tdx_h3_gemm_variant; gv.restype = C.c_int
gv.argtypes = [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_void_p]
for (m, n, k) in [(1000, 300, 64), (65536, 2048, 2048), (255968, 2176, 512), (255968, 512, 1024), (255968, 512, 512), (255968, 256, 256)]:
    a = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev); bias = torch.randn(n, device=dev)
    pa, sa = split(a); pb, sb = split(w)
    outs = {}
    for v in (0, 5, 0, 5):
        c = torch.zeros(m, n, device=dev)
        for _ in range(2):
            gv(pa.data_ptr(), sa.data_ptr(), pb.data_ptr(), sb.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, v, None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            gv(pa.data_ptr(), sa.data_ptr(), pb.data_ptr(), sb.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, v, None)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20
        outs[v] = (c, min(t, outs[v][1]) if v in outs else t)
    d = (outs[5][0] - outs[0][0]).abs().max().item()
    print(f"M={m} N={n} K={k}: product {outs[0][1]*1e3:.1f} us {2.0*m*n*k/outs[0][1]/1e9:.1f} TF | ping-pong {outs[5][1]*1e3:.1f} us {2.0*m*n*k/outs[5][1]/1e9:.1f} TF | max abs diff {d:.3e}", flush=True)
