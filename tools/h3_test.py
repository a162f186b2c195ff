"""split-f16 x3 GEMM core: accuracy against fp64 and throughput (diagnostic, GPU box)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from targetdiarization_amd import _lib
l = _lib.diag()
sp = l.tdx_h3_split_rows; sp.restype = C.c_int
sp.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p]
gm = l.tdx_h3_gemm; gm.restype = C.c_int
gm.argtypes = [C.c_void_p] * 6 + [C.c_int] * 3 + [C.c_void_p]
x6 = l.tdx_linear_variant; x6.restype = C.c_int
x6.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")


def split(x):
    r, k = x.shape
    kp = (k + 15) // 16 * 16
    planes = torch.empty(r, kp * 4, dtype=torch.uint8, device=dev)
    sc = torch.empty(r, device=dev)
    assert sp(x.data_ptr(), k, planes.data_ptr(), sc.data_ptr(), r, k, None) == 0, _lib.last_error()
    return planes, sc


def run(m, n, k, dist="n", check=True, iters=10):
    torch.manual_seed(m + n + k)
    a = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) / k ** 0.5
    if dist == "wide":
        a = a * torch.exp(torch.randn(m, k, device=dev) * 3); w = w * torch.exp(torch.randn(n, k, device=dev) * 3)
    if dist == "rows":
        a = a * torch.exp(torch.randn(m, 1, device=dev) * 8); w = w * torch.exp(torch.randn(n, 1, device=dev) * 8)
    bias = torch.randn(n, device=dev)
    c = torch.empty(m, n, device=dev)
    pa, sa = split(a); pb, sb = split(w)
    assert gm(pa.data_ptr(), sa.data_ptr(), pb.data_ptr(), sb.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, None) == 0, _lib.last_error()
    torch.cuda.synchronize()
    msg = f"M={m} N={n} K={k} {dist}:"
    if check:
        mm = min(m, 2048)
        ref = a[:mm].double() @ w.double().T + bias.double()
        e3 = ((c[:mm].double() - ref).norm() / ref.norm()).item()
        e32 = (((a[:mm] @ w.T + bias).double() - ref).norm() / ref.norm()).item()
        mx = ((c[:mm].double() - ref).abs().max() / ref.abs().max()).item()
        if m > mm:
            ref2 = a[-mm:].double() @ w.double().T + bias.double()
            e3 = max(e3, ((c[-mm:].double() - ref2).norm() / ref2.norm()).item())
        msg += f" rel_l2 x3 {e3:.2e} (torch fp32 {e32:.2e}) max {mx:.2e}"
    if iters:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2):
            gm(pa.data_ptr(), sa.data_ptr(), pb.data_ptr(), sb.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, None)
        ev0.record()
        for _ in range(iters):
            gm(pa.data_ptr(), sa.data_ptr(), pb.data_ptr(), sb.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, None)
        ev1.record(); torch.cuda.synchronize()
        t = ev0.elapsed_time(ev1) / iters
        msg += f" | x3 {t*1e3:.0f} us {2.0*m*n*k/t/1e9:.1f} TF"
        ev0.record()
        for _ in range(iters):
            sp(a.data_ptr(), k, pa.data_ptr(), sa.data_ptr(), m, k, None)
        ev1.record(); torch.cuda.synchronize()
        ts = ev0.elapsed_time(ev1) / iters
        msg += f" | split {ts*1e3:.0f} us {m*k*8/ts/1e6:.0f} GB/s"
        if n % 256 == 0:
            ev0.record()
            for _ in range(iters):
                x6(a.data_ptr(), w.data_ptr(), m, n, k, c.data_ptr(), 6, None)
            ev1.record(); torch.cuda.synchronize()
            t6 = ev0.elapsed_time(ev1) / iters
            msg += f" | x6 {t6*1e3:.0f} us {2.0*m*n*k/t6/1e9:.1f} TF"
    print(msg, flush=True)


if __name__ == "__main__":
    for shp in [(256, 256, 16), (256, 256, 32), (100, 256, 48), (300, 256, 64), (1000, 384, 96), (257, 130, 512), (4803, 2176, 512), (5000, 512, 1024)]:
        for d in ("n", "wide", "rows"):
            run(*shp, dist=d, iters=0)
    for shp in [(255968, 2176, 512), (255968, 512, 1024), (255968, 256, 512), (255968, 512, 256), (255968, 256, 256), (65536, 2048, 2048)]:
        run(*shp)
