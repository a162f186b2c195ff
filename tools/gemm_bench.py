"""Micro-benchmark of the fp32-MFMA GEMM core through tdx_linear on the path's shapes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from targetdiarization_amd import ops

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 255968
shapes = [(M, 2176, 512), (M, 512, 1024), (M, 512, 256), (M, 256, 512), (M, 256, 256)]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for (m, n, k) in shapes:
    a = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev); b = torch.randn(n, device=dev)
    c = ops.linear(a, w, b)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        c = ops.linear(a, w, b)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"M={m} N={n} K={k}: {ms:.3f} ms  {2.0*m*n*k/ms/1e9:.1f} TFLOP/s", flush=True)
