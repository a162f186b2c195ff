"""HIP-graph replay of small MossFormer2 forwards against direct launches (diagnostic)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from targetdiarization_amd import _lib
from targetdiarization_amd.separator import MossFormer2Separator
from targetdiarization_amd.weights import recipe_state_dict, recipe_wave
sd = recipe_state_dict(1, 2)
a = MossFormer2Separator(sd, "cuda:0", graph_rows=0)
B, T = 1, 50000
x = torch.from_numpy(recipe_wave("g0", 1, T)).cuda()
ref = a(x).clone()
l = _lib.lib()
need = a.workspace_bytes(B, T)
si = x.clone(); so = torch.zeros(B, 2, T, device="cuda"); ws = torch.empty(need, dtype=torch.uint8, device="cuda")
def launch():
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(l.tdx_mf2_forward(a._h, si.data_ptr(), B, T, so.data_ptr(), ws.data_ptr(), ws.numel(), st))
launch(); torch.cuda.synchronize(); print("direct", float((so - ref).abs().max()))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    launch()
for i in range(3):
    so.zero_(); torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    print("replay", i, float(so.abs().max()), float((so - ref).abs().max()), flush=True)
# which stage dies?  taps after a replay (enc / z / after_stack live in the workspace)
S = (T - 16) // 8 + 1
for name in ("enc", "z", "after_stack"):
    dst = torch.empty(B * S * 512, device="cuda"); cnt = C.c_size_t()
    _lib.check(l.tdx_mf2_tap(a._h, name.encode(), B, T, ws.data_ptr(), dst.data_ptr(), dst.numel(), C.byref(cnt), None))
    torch.cuda.synchronize(); print(name, float(dst.abs().max()), bool(torch.isfinite(dst).all()))
