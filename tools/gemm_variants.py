import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from targetdiarization_amd import _lib
l = _lib.diag()
f = l.tdx_linear_variant
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
shapes = [(255968, 2176, 512), (255968, 512, 1024), (255968, 512, 256)]
if len(sys.argv) > 2: shapes = shapes[:int(sys.argv[2])]
for (m, n, k) in shapes:
    a = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev); c = torch.empty(m, n, device=dev)
    for var in (0, 4, 1):
        f(a.data_ptr(), w.data_ptr(), m, n, k, c.data_ptr(), var, None); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f(a.data_ptr(), w.data_ptr(), m, n, k, c.data_ptr(), var, None)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"M={m} N={n} K={k} variant={var}: {ms:.3f} ms {2.0*m*n*k/ms/1e9:.1f} TF", flush=True)
