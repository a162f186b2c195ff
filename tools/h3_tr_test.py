"""x3 GEMM with K-major (token-major) operands: accuracy vs fp64 and throughput (diagnostic, GPU box)."""
import sys, os, math, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from targetdiarization_amd import _lib
from tools.h3_test import split, dev
l = _lib.diag()
sk = l.tdx_h3_split_kmajor; sk.restype = C.c_int
sk.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_float, C.c_void_p]
gx = l.tdx_h3_gemm_x; gx.restype = C.c_int
gx.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_int] * 3 + [C.c_void_p]


def split_k(x):           # x [K][N] -> K-major planes + one scale
    k, n = x.shape
    e = math.floor(math.log2(float(x.abs().max()) * 4.0))      # a loose bound (x4) like the static bounds of the model
    s = 2.0 ** (14 - e)
    planes = torch.empty(k, n * 4, dtype=torch.uint8, device=dev)
    assert sk(x.data_ptr(), n, planes.data_ptr(), k, n, s, None) == 0, _lib.last_error()
    return planes, torch.full((1,), 1.0 / s, device=dev)


def run(mode, m, n, k, iters=0):
    torch.manual_seed(m + n + k + mode)
    a = torch.randn(m, k, device=dev); b = torch.randn(n, k, device=dev) / k ** 0.5
    pa, sa = split_k(a.t().contiguous()) if mode & 1 else split(a)
    pb, sb = split_k(b.t().contiguous()) if mode & 2 else split(b)
    c = torch.empty(m, n, device=dev)
    assert gx(mode, pa.data_ptr(), sa.data_ptr(), pb.data_ptr(), sb.data_ptr(), c.data_ptr(), m, n, k, None) == 0, _lib.last_error()
    torch.cuda.synchronize()
    mm = min(m, 2048)
    ref = a[:mm].double() @ b.double().T
    err = ((c[:mm].double() - ref).norm() / ref.norm()).item()
    e32 = (((a[:mm] @ b.T).double() - ref).norm() / ref.norm()).item()
    msg = f"mode {mode} M={m} N={n} K={k}: rel_l2 {err:.2e} (torch fp32 {e32:.2e})"
    if iters:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2):
            gx(mode, pa.data_ptr(), sa.data_ptr(), pb.data_ptr(), sb.data_ptr(), c.data_ptr(), m, n, k, None)
        e0.record()
        for _ in range(iters):
            gx(mode, pa.data_ptr(), sa.data_ptr(), pb.data_ptr(), sb.data_ptr(), c.data_ptr(), m, n, k, None)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / iters
        msg += f" | {t*1e3:.0f} us {2.0*m*n*k/t/1e9:.1f} TF"
    print(msg, flush=True)


if __name__ == "__main__":
    for mode in (2, 1, 3):
        for shp in [(256, 256, 16), (256, 256, 64), (256, 128, 128), (512, 384, 96), (2048, 128, 512), (300, 256, 256)]:
            if mode & 1 and shp[0] % 128:
                continue
            run(mode, *shp)
    for mode in (0, 2, 1, 3):
        run(mode, 65536, 2048, 256, iters=10)
        run(mode, 8192, 2048, 2048, iters=10)
    run(3, 2048, 128, 8192, iters=10)
