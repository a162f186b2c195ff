"""MDX-Net body (csrc/mdx.hip) at the reference's geometry: L = 11, l = 3, g = 32, bn = 8, dim_f = 3072, dim_t = 256 (one block = 11.8 s of
44.1 kHz stereo at quality 3), recipe weights: ms per launch of 8 blocks, algorithmic TFLOP/s, audio-seconds per second of the body alone."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from targetdiarization_amd.mdx import ConvTDFNetBody
from targetdiarization_amd.weights import recipe_mdx_state_dict
dev = torch.device("cuda:0")
body = ConvTDFNetBody(recipe_mdx_state_dict(0), dev, max_blocks_per_launch=8)
x = torch.randn(8, 4, 3072, 256, device=dev)
for _ in range(2):
    y = body(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 5
for _ in range(n):
    y = body(x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
fl = body.flops(8)
print(f"8 blocks: {dt * 1e3:.1f} ms, {fl / dt / 1e12:.1f} TFLOP/s algorithmic (fp32 MFMA core, peak 157.3), {8 * 516096 / 44100 / dt:.0f} audio-s/s (new samples per block = 516 096), finite = {bool(torch.isfinite(y).all())}")
print(f"workspace {body._l.tdx_mdx_workspace_bytes(body._h, 8) / 2**30:.2f} GiB")
