"""streaming a 2 GB operand as column strips (K^T[V|U] pattern) against a contiguous stream (diagnostic)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from targetdiarization_amd import _lib
l = _lib.diag()
g = l.tdx_fill_bench3; g.restype = C.c_int
g.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0")
sink = torch.zeros(4, device=dev)
rows = 262144                      # 32 windows x 8192 tokens, 8 KB per token row = 2 GB
src = torch.empty(rows * 8192, dtype=torch.uint8, device=dev).random_(0, 255)
for (label, pitch, seg, strips) in [("8 KB rows, 1 KB strips (x3 tile 256 columns)", 8192, 1024, 8), ("8 KB rows, 2 KB strips (512 columns)", 8192, 2048, 4),
                                    ("8 KB rows, 4 KB strips", 8192, 4096, 2), ("8 KB rows, whole rows", 8192, 8192, 1),
                                    ("strip-major 2 KB rows (contiguous per block)", 2048, 2048, 1)]:
    blocks = 2048
    nrows = rows if pitch == 8192 else rows * 4
    rpb = nrows // (blocks // strips)
    iters = int(rpb * seg // 32768)
    for rep in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g(src.data_ptr(), pitch, seg, rpb, strips, blocks, iters, sink.data_ptr(), None)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3
    gb = blocks * iters * 32768 / 1e9
    print(f"{label}: {gb:.2f} GB in {t*1e6:.0f} us = {gb / t / 1e3:.2f} TB/s", flush=True)
