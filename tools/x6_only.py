import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from targetdiarization_amd import _lib
l = _lib.diag()
f = l.tdx_linear_variant
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
m, n, k = 255968, 2048, 512
a = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev); c = torch.empty(m, n, device=dev)
for _ in range(3):
    f(a.data_ptr(), w.data_ptr(), m, n, k, c.data_ptr(), 6, None)
torch.cuda.synchronize()
