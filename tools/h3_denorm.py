"""Does the x3 GEMM keep its precision when a K-major operand sits far below its (static) bound, i.e. when the lo plane is made of
f16 SUBNORMALS?  rel-L2 vs fp64 as a function of the headroom 2^h between the bound and the largest element."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_h3 import split_kmajor, split_rows, rel, dev
from targetdiarization_amd import _lib
h3 = _lib.diag()
g = torch.Generator().manual_seed(0)
m, n, k = 512, 256, 512
a = torch.randn(m, k, generator=g).to(dev); b = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev)
ref = a.double() @ b.double().T
for h in (2, 8, 12, 14, 16, 18, 20, 22, 24):
    out = []
    for mode in (2, 1):
        pa, sa = (split_kmajor(h3, a.t().contiguous(), 2.0 ** h) if mode & 1 else split_rows(h3, a))
        pb, sb = (split_kmajor(h3, b.t().contiguous(), 2.0 ** h) if mode & 2 else split_rows(h3, b))
        c = torch.empty(m, n, device=dev)
        assert h3.tdx_h3_gemm_x(mode, pa.data_ptr(), sa.data_ptr(), pb.data_ptr(), sb.data_ptr(), c.data_ptr(), m, n, k, None) == 0
        out.append(rel(c, ref))
    # representation error of the planes themselves (what a perfect product of them would give)
    pb, sb = split_kmajor(h3, b.t().contiguous(), 2.0 ** h)
    p = pb.view(torch.float16).reshape(k, n // 32, 2, 32).float()
    rec = (p[:, :, 0] + p[:, :, 1]).reshape(k, n) * sb
    print(f"headroom 2^{h:2d}: B K-major {out[0]:.2e}   A K-major {out[1]:.2e}   planes representation {rel(rec, b.t().double()):.2e}", flush=True)
