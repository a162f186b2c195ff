"""ERes2NetV2 at the benchmark size (B x F = 998 frames): ms per forward (child process; tools/eres_prof.sh profiles the same child).
usage: python tools/eres_bench.py [B=60]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    from targetdiarization_amd.speaker import ERes2NetV2
    from targetdiarization_amd.weights import recipe_eres2netv2_state_dict
    B = int(sys.argv[2])
    m = ERes2NetV2(recipe_eres2netv2_state_dict(0), device="cuda:0")
    g = torch.Generator().manual_seed(3)
    feat = (torch.randn(B, 998, 80, generator=g)).cuda()
    out = m.embed_features(feat)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        out = m.embed_features(feat)
    e1.record(); torch.cuda.synchronize()
    np.save(sys.argv[3], out.cpu().numpy())
    print(f"B={B} x F=998: {e0.elapsed_time(e1) / 3:.2f} ms per forward = {m.flops(B, 998) / (e0.elapsed_time(e1) / 3) / 1e9:.1f} TFLOP/s algorithmic", flush=True)
    sys.exit(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 60
subprocess.run([sys.executable, __file__, "child", str(B), "/tmp/eres.npy"], check=True)
