"""Does MossFormer2 gain from two half-batches in flight on two streams (two model objects = two handles and workspaces)?
Kernels of different character then overlap (MFMA-bound GEMMs of one half with the HBM-bound streaming kernels of the other).
usage: python tools/two_stream_sep.py [windows=32] [samples=64000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from targetdiarization_amd.separator import MossFormer2Separator
from targetdiarization_amd.weights import recipe_state_dict
W = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64000
sd = recipe_state_dict(0, 24)
a = MossFormer2Separator(sd, device="cuda:0", graph_rows=0)
b = MossFormer2Separator(sd, device="cuda:0", graph_rows=0)
x = (torch.randn(W, N, generator=torch.Generator().manual_seed(1)) * 0.1).cuda()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def one():
    return a(x)
def two():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): ya = a(x[: W // 2])
    with torch.cuda.stream(s2): yb = b(x[W // 2:])
    cur.wait_stream(s1); cur.wait_stream(s2)
    return ya, yb
def seq():
    return a(x[: W // 2]), a(x[W // 2:])
for name, fn in (("one call of %d windows" % W, one), ("two halves back to back", seq), ("two halves on two streams", two), ("one call of %d windows" % W, one), ("two halves on two streams", two)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    print(f"{name:32s} {(time.perf_counter() - t0) / 5 * 1e3:8.2f} ms", flush=True)
ya, yb = two(); y = one(); torch.cuda.synchronize()
print("max |diff| halves vs whole:", float((torch.cat([ya, yb]) - y).abs().max()))
