"""x3 core: the wide kernel (256x256 tile, one block per CU) against the narrow kernel (256x128, two blocks per CU) and
the 16x16x32 MFMA shape as a timing variant of both, on random and on all-zero operands (diagnostic, GPU box).
variant 0 wide product | 6 wide 16x16x32 (timing only) | 10 narrow product | 16 narrow 16x16x32 (timing only)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from targetdiarization_amd import _lib
from tools.h3_test import split, dev
l = _lib.diag()
gv = l.tdx_h3_gemm_variant
shapes = [(255968, 2176, 512), (65536, 2048, 2048), (255968, 512, 1024), (255968, 256, 512), (255968, 512, 256), (17328, 2176, 512), (17328, 512, 256)]
if len(sys.argv) > 1:
    shapes = shapes[: int(sys.argv[1])]
for (m, n, k) in shapes:
    bias = torch.randn(n, device=dev); c = torch.empty(m, n, device=dev)
    a = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev)
    pa, sa = split(a); pb, sb = split(w)
    # correctness of the narrow kernel: bit-identical accumulation order per element -> equal to the wide kernel
    c0 = torch.empty(m, n, device=dev); c1 = torch.empty(m, n, device=dev)
    assert gv(pa.data_ptr(), sa.data_ptr(), pb.data_ptr(), sb.data_ptr(), bias.data_ptr(), c0.data_ptr(), m, n, k, 0, None) == 0
    assert gv(pa.data_ptr(), sa.data_ptr(), pb.data_ptr(), sb.data_ptr(), bias.data_ptr(), c1.data_ptr(), m, n, k, 10, None) == 0
    c2 = torch.empty(m, n, device=dev)
    assert gv(pa.data_ptr(), sa.data_ptr(), pb.data_ptr(), sb.data_ptr(), bias.data_ptr(), c2.data_ptr(), m, n, k, 20, None) == 0
    torch.cuda.synchronize()
    mm = min(m, 4096)
    ref = a[:mm].double() @ w.double().T + bias.double()
    e_w = float((c0[:mm].double() - ref).norm() / ref.norm()); e_n = float((c1[:mm].double() - ref).norm() / ref.norm())
    e_p = float((c2[:mm].double() - ref).norm() / ref.norm())
    print(f"M={m} N={n} K={k}: wide vs fp64 {e_w:.2e}, narrow vs fp64 {e_n:.2e}, narrow == wide: {bool(torch.equal(c0, c1))}, pair-stage vs fp64 {e_p:.2e}, "
          f"pair-stage vs wide (all rows) {float((c2.double() - c0.double()).norm() / c0.double().norm()):.2e}", flush=True)
    del c0, c1, c2, ref
    for label in ("random", "zeros"):
        if label == "zeros":
            pa.zero_(); pb.zero_()
        for v in (0, 6, 20, 10):
            for _ in range(2):
                gv(pa.data_ptr(), sa.data_ptr(), pb.data_ptr(), sb.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, v, None)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                gv(pa.data_ptr(), sa.data_ptr(), pb.data_ptr(), sb.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, v, None)
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 10
            print(f"  {label:6s} variant {v:2d}: {t*1e3:8.1f} us {2.0*m*n*k/t/1e9:7.1f} TF", flush=True)
    del a, w, pa, pb
