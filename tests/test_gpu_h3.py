"""-m gpu: the split-f16 x3 GEMM core (csrc/gemm_h3.hpp) on its own: operand formats, every operand
mode against an fp64 product, range robustness of the exponent-aligned split.  Tolerance: the
result must be at least as accurate as an fp32 GEMM (rel-L2 vs fp64 below 2e-6 at K <= 2048)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def h3():
    from targetdiarization_amd import _lib
    return _lib.diag()          # libtdx_diag.so: the hooks of include/tdx_test.h


dev = torch.device("cuda:0")


def split_rows(l, x):
    r, k = x.shape
    kp = (k + 15) // 16 * 16
    planes = torch.zeros(r, kp * 4, dtype=torch.uint8, device=dev)
    sc = torch.empty(r, device=dev)
    assert l.tdx_h3_split_rows(x.data_ptr(), k, planes.data_ptr(), sc.data_ptr(), r, k, None) == 0
    return planes, sc


def split_kmajor(l, x, loose=4.0):
    k, n = x.shape
    e = math.floor(math.log2(max(float(x.abs().max()), 1e-30) * loose))
    s = 2.0 ** (14 - e)
    planes = torch.zeros(k, n * 4, dtype=torch.uint8, device=dev)
    assert l.tdx_h3_split_kmajor(x.data_ptr(), n, planes.data_ptr(), k, n, s, None) == 0
    return planes, torch.full((1,), 1.0 / s, device=dev)


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


def test_split_rows_format_and_precision(h3):
    torch.manual_seed(0)
    x = torch.randn(37, 96, device=dev) * torch.exp(torch.randn(37, 1, device=dev) * 6)
    x[5] = 0.0
    x[7, 3:] = x[7, 3:] * 1e-7                      # one dominant element: the small ones keep an ABSOLUTE error bound
    planes, sc = split_rows(h3, x)
    p = planes.view(torch.float16).reshape(37, 96 // 8, 2, 8).float()
    rec = (p[:, :, 0] + p[:, :, 1]).reshape(37, 96) * sc[:, None]
    mu = x.abs().amax(dim=1, keepdim=True)
    err = (rec - x).abs()
    assert bool((err <= torch.maximum(x.abs() * 2.0 ** -21, mu * 2.0 ** -38)).all())
    assert float(p[5].abs().max()) == 0.0 and float(sc[5]) == 1.0
    lg = torch.log2(sc[sc != 1.0])
    assert torch.equal(lg, lg.round())             # scales are powers of two
    assert float((p[:, :, 0].abs().amax(dim=(1, 2))[mu[:, 0] > 0]).min()) >= 2.0 ** 14      # row max lands in [2^14, 2^15)
    assert float(p[:, :, 0].abs().max()) < 2.0 ** 15 + 16


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_gemm_modes_vs_fp64(h3, mode):
    """mode bit 0: A in K-major planes, bit 1: B in K-major planes (ds_read_b64_tr_b16 gathers)."""
    g = torch.Generator(device="cpu").manual_seed(mode)
    shapes = [(256, 256, 16), (256, 128, 128), (512, 384, 96), (2048, 128, 512), (384, 256, 2048)]
    if not mode & 1:
        shapes += [(300, 256, 256), (1, 256, 64), (257, 130 if not mode & 2 else 128, 512)]
    for (m, n, k) in shapes:
        a = torch.randn(m, k, generator=g).to(dev); b = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev)
        if m > 4:
            a[3] *= 1e4; a[4] *= 1e-6                  # per-row dynamic range (row-major planes only use it)
        pa, sa = split_kmajor(h3, a.t().contiguous(), 1.0) if mode & 1 else split_rows(h3, a)
        pb, sb = split_kmajor(h3, b.t().contiguous()) if mode & 2 else split_rows(h3, b)
        c = torch.full((m, n), float("nan"), device=dev)
        assert h3.tdx_h3_gemm_x(mode, pa.data_ptr(), sa.data_ptr(), pb.data_ptr(), sb.data_ptr(), c.data_ptr(), m, n, k, None) == 0
        ref = a.double() @ b.double().T
        rows = torch.ones(m, dtype=torch.bool, device=dev)
        if mode & 1 and m > 4:
            rows[4] = False                          # 1e-10 of the tensor max: below the uniform scale's precision window
        assert rel(c[rows], ref[rows]) < 2e-6, (mode, m, n, k)
        assert bool(torch.isfinite(c).all())


def test_gemm_extreme_rows_and_columns(h3):
    """the row exponents carry the dynamic range: rows / weight rows spanning 60 binades stay fp32-accurate"""
    torch.manual_seed(1)
    m, n, k = 512, 256, 512
    a = torch.randn(m, k, device=dev) * torch.exp2(torch.randint(-30, 30, (m, 1), device=dev).float())
    b = torch.randn(n, k, device=dev) * torch.exp2(torch.randint(-30, 30, (n, 1), device=dev).float())
    pa, sa = split_rows(h3, a); pb, sb = split_rows(h3, b)
    c = torch.empty(m, n, device=dev)
    assert h3.tdx_h3_gemm_x(0, pa.data_ptr(), sa.data_ptr(), pb.data_ptr(), sb.data_ptr(), c.data_ptr(), m, n, k, None) == 0
    ref = a.double() @ b.double().T
    assert float(((c.double() - ref).abs() / ref.abs().clamp_min(1e-300)).median()) < 1e-6
    assert float(((c.double() - ref).norm(dim=1) / ref.norm(dim=1)).max()) < 2e-6


@pytest.mark.parametrize("shape", [(1000, 300, 64), (4096, 512, 48), (3000, 2176, 512)])
def test_pingpong_loop_is_bit_identical(h3, shape):
    """the ping-pong main loop (diagnostic variant 5: MFMA bursts and load phases alternate between the two waves of a
    SIMD) accumulates in the same order as the product loop: identical bits, partial edge tiles and short K included"""
    torch.manual_seed(3)
    m, n, k = shape
    gv = h3.tdx_h3_gemm_variant
    gv.restype = C.c_int
    gv.argtypes = [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_void_p]
    a = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev); bias = torch.randn(n, device=dev)
    pa, sa = split_rows(h3, a); pb, sb = split_rows(h3, w)
    out = []
    for v in (0, 5):
        c = torch.zeros(m, n, device=dev)
        assert gv(pa.data_ptr(), sa.data_ptr(), pb.data_ptr(), sb.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, v, None) == 0
        out.append(c)
    torch.cuda.synchronize()
    assert torch.equal(out[0], out[1])
    ref = a.double() @ w.double().T + bias.double()
    assert rel(out[1], ref) < 1e-6
