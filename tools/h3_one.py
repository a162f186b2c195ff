"""one shape / one variant of the x3 core, a few launches (for rocprofv3 --pmc passes)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from targetdiarization_amd import _lib
from tools.h3_test import split, dev
l = _lib.diag()
gv = l.tdx_h3_gemm_variant; gv.restype = C.c_int
gv.argtypes = [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_void_p]
v = int(sys.argv[1]) if len(sys.argv) > 1 else 0
m, n, k = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (65536, 2048, 2048)
a = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev); bias = torch.zeros(n, device=dev); c = torch.empty(m, n, device=dev)
pa, sa = split(a); pb, sb = split(w)
for _ in range(4):
    gv(pa.data_ptr(), sa.data_ptr(), pb.data_ptr(), sb.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, v, None)
torch.cuda.synchronize()
