"""-m gpu: the MDX-Net denoiser body (csrc/mdx.hip) vs its oracle (oracle/mdx_oracle.py: the published KUIELab ConvTDFNet
forward restated; third-party, the reference only holds an .onnx path — parity unpinned), recipe weights.  Tolerance 1e-4 rel-L2."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
dev = torch.device("cuda:0")


def rel_l2(a, b):
    a = torch.as_tensor(a).double().cpu().reshape(-1); b = torch.as_tensor(b).double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize("cfg", [dict(L=5, l=2, g=32, bn=8, dim_f=256, dim_t=32), dict(L=3, l=1, g=32, bn=4, dim_f=128, dim_t=8),
                                 dict(L=7, l=3, g=32, bn=8, dim_f=768, dim_t=64), dict(L=5, l=2, g=64, bn=8, dim_f=256, dim_t=16)])
def test_body_vs_oracle(cfg):
    from oracle import mdx_oracle as mo
    from targetdiarization_amd.mdx import ConvTDFNetBody
    from targetdiarization_amd.weights import recipe_mdx_state_dict
    sd = recipe_mdx_state_dict(0, L=cfg["L"], l=cfg["l"], g=cfg["g"], bn=cfg["bn"], dim_f=cfg["dim_f"])
    body = ConvTDFNetBody(sd, dev, L=cfg["L"], l=cfg["l"], g=cfg["g"], bn=cfg["bn"], dim_f=cfg["dim_f"], dim_t=cfg["dim_t"], max_blocks_per_launch=2)
    g = torch.Generator().manual_seed(cfg["dim_f"])
    x = torch.randn(3, 4, cfg["dim_f"], cfg["dim_t"], generator=g)
    ref = mo.conv_tdf_net_forward(x.double(), {k: v.double() for k, v in sd.items()}, L=cfg["L"], l=cfg["l"])
    out = body(x.to(dev))                      # 3 blocks in launches of 2 + 1
    assert out.shape == x.shape and bool(torch.isfinite(out).all())
    e = rel_l2(out, ref)
    assert e < 1e-4, (cfg, e)
    one = body(x[2:3].to(dev))                 # batch independence
    assert rel_l2(one, out[2:3]) < 1e-6
    assert body.flops(1) > 0


def test_tdf_bias_and_strict_loading():
    from oracle import mdx_oracle as mo
    from targetdiarization_amd import _lib
    from targetdiarization_amd.mdx import ConvTDFNetBody
    from targetdiarization_amd.weights import recipe_mdx_state_dict
    sd = recipe_mdx_state_dict(1, L=3, l=2, g=32, bn=8, dim_f=128, bias=True)
    body = ConvTDFNetBody(sd, dev, L=3, l=2, g=32, bn=8, dim_f=128, dim_t=16)
    x = torch.randn(1, 4, 128, 16, generator=torch.Generator().manual_seed(3))
    ref = mo.conv_tdf_net_forward(x.double(), {k: v.double() for k, v in sd.items()}, L=3, l=2)
    assert rel_l2(body(x.to(dev)), ref) < 1e-4
    bad = dict(sd); bad.pop("ds.0.0.bias")
    with pytest.raises(_lib.TdxError, match="ds.0.0.bias"):
        ConvTDFNetBody(bad, dev, L=3, l=2, g=32, bn=8, dim_f=128, dim_t=16)
    extra = dict(sd); extra["not_a_tensor"] = torch.zeros(3)
    with pytest.raises(_lib.TdxError, match="unexpected"):
        ConvTDFNetBody(extra, dev, L=3, l=2, g=32, bn=8, dim_f=128, dim_t=16)
    with pytest.raises(_lib.TdxError, match="unsupported config"):
        ConvTDFNetBody(sd, dev, L=3, l=2, g=32, bn=8, dim_f=100, dim_t=16)


def test_denoise_vocal_with_the_device_body(gold):
    """AudioProcessor.denoise_vocal end to end with the device body as `mdx_model` (resample -> block STFT -> ConvTDFNet -> iSTFT ->
    the inst-model subtraction): finite, right length, equal to the same chain with the oracle body (small geometry: quality 3 blocks are
    3072 x 256 — the oracle needs minutes there)"""
    from oracle import mdx_oracle as mo
    from targetdiarization_amd.audio_processor import AudioProcessor
    from targetdiarization_amd.mdx import ConvTDFNetBody
    from targetdiarization_amd.weights import recipe_mdx_state_dict
    sd = recipe_mdx_state_dict(2, L=3, l=1, g=32, bn=8, dim_f=3072)
    body = ConvTDFNetBody(sd, dev, L=3, l=1, g=32, bn=8, dim_f=3072, dim_t=256, max_blocks_per_launch=2)
    sd64 = {k: v.double() for k, v in sd.items()}
    ap_dev = AudioProcessor(is_denoise_vocal=True, mdx_weights_file="mdx/weights/UVR-MDX-NET-Inst_HQ_3.onnx", cuda_device=0, quality=3,
                            verbose_log=False, mdx_model=body)
    ap_ref = AudioProcessor(is_denoise_vocal=True, mdx_weights_file="mdx/weights/UVR-MDX-NET-Inst_HQ_3.onnx", cuda_device=0, quality=3,
                            verbose_log=False, mdx_model=lambda spec: mo.conv_tdf_net_forward(spec.double().cpu(), sd64, L=3, l=1).float().to(spec.device))
    rng = np.random.default_rng(4)
    x = (0.1 * rng.standard_normal((3 * 44100, 2))).astype(np.float32)
    y = ap_dev.denoise_vocal(x, 44100)
    y0 = ap_ref.denoise_vocal(x, 44100)
    assert y.shape == y0.shape == x.shape and np.isfinite(y).all()
    assert np.linalg.norm(y.astype(np.float64) - y0) / max(np.linalg.norm(y0), 1e-30) < 1e-4
