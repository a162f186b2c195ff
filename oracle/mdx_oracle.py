"""MDX-Net denoiser body = KUIELab ConvTDFNet (TFC-TDF v2), CPU restatement (test infrastructure; checker only).

The reference runs this network as an OPAQUE ONNX file through onnxruntime (`self.mdx_model.run(None, {"input": mix_spec})`,
AudioProcessor.py:224-241, :630; "UVR-MDX-NET-Inst_HQ_3.onnx"); neither the file nor onnxruntime nor the PyTorch source is in
the reference tree.  PARITY UNPINNED: this restates the published KUIELab mdx-net `ConvTDFNet.forward` [upstream-recall]:
    first_conv (1x1, dim_c -> g) + BN + ReLU;  x = x.transpose(-1, -2)                         # [B, g, T, F]
    n = L // 2 encoder levels:  x = TFC_TDF(x); skip.append(x); x = ReLU(BN(Conv2d(c -> c+g, 2x2, stride 2)(x)))
    bottleneck TFC_TDF
    n decoder levels:           x = ReLU(BN(ConvTranspose2d(c -> c-g, 2x2, stride 2)(x))); x = x * skip.pop(); x = TFC_TDF(x)
    x = x.transpose(-1, -2);  final_conv (1x1, g -> dim_c)
    TFC_TDF(x): x = l x [Conv2d(c, c, k, padding k//2) + BN + ReLU];  return x + TDF(x)
    TDF(x):     ReLU(BN(Linear(f -> f/bn)(x))) -> ReLU(BN(Linear(f/bn -> f)(.)))   (Linear over the LAST dim = frequency, BN over channels)
BatchNorm2d in eval mode (running statistics).  Input / output [B, dim_c, dim_f, dim_t] like the ONNX graph."""
import torch
import torch.nn.functional as F


def _bn(x, sd, p, eps=1e-5):
    w, b, m, v = (sd[p + k] for k in ("weight", "bias", "running_mean", "running_var"))
    sh = (1, -1, 1, 1)
    return (x - m.view(sh)) / torch.sqrt(v.view(sh) + eps) * w.view(sh) + b.view(sh)


def _tfc_tdf(x, sd, p, l, k):
    for j in range(l):
        x = F.relu(_bn(F.conv2d(x, sd[f"{p}tfc.H.{j}.0.weight"], sd[f"{p}tfc.H.{j}.0.bias"], padding=k // 2), sd, f"{p}tfc.H.{j}.1."))
    t = F.relu(_bn(F.linear(x, sd[p + "tdf.0.weight"], sd.get(p + "tdf.0.bias")), sd, p + "tdf.1."))
    t = F.relu(_bn(F.linear(t, sd[p + "tdf.3.weight"], sd.get(p + "tdf.3.bias")), sd, p + "tdf.4."))
    return x + t


def conv_tdf_net_forward(spec, sd, L: int = 11, l: int = 3, k: int = 3):
    """spec [B, dim_c, dim_f, dim_t] -> [B, dim_c, dim_f, dim_t]"""
    n = L // 2
    x = F.relu(_bn(F.conv2d(spec, sd["first_conv.0.weight"], sd["first_conv.0.bias"]), sd, "first_conv.1."))
    x = x.transpose(-1, -2)
    skips = []
    for i in range(n):
        x = _tfc_tdf(x, sd, f"encoding_blocks.{i}.", l, k)
        skips.append(x)
        x = F.relu(_bn(F.conv2d(x, sd[f"ds.{i}.0.weight"], sd[f"ds.{i}.0.bias"], stride=2), sd, f"ds.{i}.1."))
    x = _tfc_tdf(x, sd, "bottleneck_block.", l, k)
    for i in range(n):
        x = F.relu(_bn(F.conv_transpose2d(x, sd[f"us.{i}.0.weight"], sd[f"us.{i}.0.bias"], stride=2), sd, f"us.{i}.1."))
        x = x * skips[-i - 1]
        x = _tfc_tdf(x, sd, f"decoding_blocks.{i}.", l, k)
    x = x.transpose(-1, -2)
    return F.conv2d(x, sd["final_conv.0.weight"], sd["final_conv.0.bias"])
