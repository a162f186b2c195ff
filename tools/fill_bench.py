"""per-CU operand fill rate: LDS-DMA vs register staging, L2-resident and HBM-streaming sources (diagnostic)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from targetdiarization_amd import _lib
l = _lib.diag()
f = l.tdx_fill_bench; f.restype = C.c_int
f.argtypes = [C.c_int, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0")
sink = torch.zeros(4, device=dev)
for (label, per_block) in [("L2-resident 64 KB/block", 65536), ("1 MB/block (L2+MALL)", 1 << 20), ("HBM stream 32 MB/block", 32 << 20)]:
    src = torch.empty(256 * per_block, dtype=torch.uint8, device=dev).random_(0, 255)
    for mode in (0, 1):
        iters = 4096 if per_block >= (1 << 20) else 8192
        if per_block >= (32 << 20):
            iters = 1024
        for _ in range(2):
            f(mode, src.data_ptr(), per_block, 256, iters, sink.data_ptr(), None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        f(mode, src.data_ptr(), per_block, 256, iters, sink.data_ptr(), None)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3
        gb = 256 * iters * 32768 / 1e9
        print(f"{label}: mode {'LDS-DMA' if mode == 0 else 'regs+ds_write'}: {gb / t / 1e3:.2f} TB/s chip = {gb / t / 256:.1f} GB/s per CU ({t / iters * 1e6:.2f} us per 32 KB)", flush=True)
    del src

# GEMM-shaped slabs: row-major planes pattern (16 rows x 64 B per wave instruction) against contiguous tiles of the same bytes
g = l.tdx_fill_bench2; g.restype = C.c_int
g.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
for stride in (2048, 8192):
    src = torch.empty(34 * 256 * stride, dtype=torch.uint8, device=dev).random_(0, 255)
    for mode in (0, 1, 0, 1):
        iters = 8192
        for _ in range(2):
            g(mode, src.data_ptr(), stride, 256, iters, sink.data_ptr(), None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g(mode, src.data_ptr(), stride, 256, iters, sink.data_ptr(), None)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3
        gb = 256 * iters * 32768 / 1e9
        print(f"GEMM slabs, row stride {stride}: {'contiguous 16 KB tiles' if mode == 0 else '16 rows x 64 B per instr'}: "
              f"{gb / t / 256:.1f} GB/s per CU ({t / iters * 1e6:.2f} us per 32 KB)", flush=True)
    del src
