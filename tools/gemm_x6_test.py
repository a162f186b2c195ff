import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from targetdiarization_amd import _lib
l = _lib.diag()
f = l.tdx_linear_variant
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
def rel(a, b): return float((a.double() - b).norm() / b.norm())
# accuracy
for (m, n, k) in [(300, 256, 512), (1000, 512, 1024), (77, 256, 16), (513, 768, 256)]:
    g = torch.Generator().manual_seed(m)
    a = torch.randn(m, k, generator=g); w = torch.randn(n, k, generator=g) / k ** 0.5
    ref = a.double() @ w.double().T
    out = {}
    ad, wd = a.to(dev), w.to(dev)
    for var in (0, 6):
        c = torch.zeros(m, n, device=dev)
        rc = f(ad.data_ptr(), wd.data_ptr(), m, n, k, c.data_ptr(), var, None); torch.cuda.synchronize()
        assert rc == 0, l.tdx_last_error()
        out[var] = rel(c.cpu(), ref)
    print(f"acc M={m} N={n} K={k}: fp32-mfma {out[0]:.3e}  bf16x6 {out[6]:.3e}", flush=True)
# asymmetric identity check
a = torch.eye(256); w = torch.arange(256 * 256, dtype=torch.float32).reshape(256, 256) / 1000.0
c = torch.zeros(256, 256, device=dev)
ad, wd = a.to(dev), w.to(dev)
rc = f(ad.data_ptr(), wd.data_ptr(), 256, 256, 256, c.data_ptr(), 6, None); torch.cuda.synchronize()
print("rc", rc, l.tdx_last_error())
print("identity max err", float((c.cpu() - w.T).abs().max()))
# speed
for (m, n, k) in [(255968, 2048, 512), (255968, 512, 1024), (255968, 512, 256), (255968, 256, 512), (255968, 256, 256)]:
    a = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev); c = torch.empty(m, n, device=dev)
    for var in (0, 6):
        f(a.data_ptr(), w.data_ptr(), m, n, k, c.data_ptr(), var, None); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f(a.data_ptr(), w.data_ptr(), m, n, k, c.data_ptr(), var, None)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"M={m} N={n} K={k} variant={var}: {ms:.3f} ms {2.0*m*n*k/ms/1e9:.1f} TF", flush=True)
