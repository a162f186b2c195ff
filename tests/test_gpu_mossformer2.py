"""-m gpu parity tests: HIP path (through the C-ABI) vs the oracle and the golden fixtures
minted from the reference (oracle/make_goldens.py).  Tolerance (BASELINE.json north_star):
1e-4 relative (rel-L2) and 1e-3 cosine; the fp32 noise floor of the reference itself is 7e-6."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4


def rel_l2(a, b):
    a = a.detach().double().cpu().reshape(-1)
    b = torch.as_tensor(b).double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm())


def cos(a, b):
    a = a.detach().double().cpu().reshape(-1)
    b = torch.as_tensor(b).double().cpu().reshape(-1)
    return float(torch.dot(a, b) / (a.norm() * b.norm()))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def sep24(sd24, dev):
    from targetdiarization_amd.separator import MossFormer2Separator
    return MossFormer2Separator(sd24, device=dev)


def test_linear_vs_fp64(dev):
    from targetdiarization_amd import ops
    g = torch.Generator().manual_seed(0)
    for (M, N, K) in [(300, 256, 512), (128, 128, 32), (1000, 2176, 512), (77, 512, 1024)]:
        a = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g); b = torch.randn(N, generator=g)
        ref = a.double() @ w.double().T + b.double()
        out = ops.linear(a.to(dev), w.to(dev), b.to(dev))
        assert rel_l2(out, ref) < 2e-6, (M, N, K)
    # asymmetric identity check (catches transposed C writes)
    a = torch.eye(128, 128); w = torch.arange(128 * 128, dtype=torch.float32).reshape(128, 128) / 1000.0
    out = ops.linear(a.to(dev), w.to(dev))
    assert torch.equal(out.cpu(), w.T.contiguous())


def test_cal_attention_golden(gold, sd24, dev):
    """G2: reference cal_attention on [2,600,*] (600 -> 768 group padding)."""
    from oracle import mossformer2_oracle as orc
    from targetdiarization_amd import ops
    from targetdiarization_amd.weights import philox_uniform
    fx = np.load(os.path.join(gold, "g2_cal_attention_2x600.npz"))
    B, S, E = 2, 600, 64
    u_ = lambda nm, *shape: torch.from_numpy(philox_uniform("g2:" + nm, int(np.prod(shape)))).reshape(shape)
    qq, lq, qk, lk = (u_(n, B, S, 128) for n in ("qq", "lq", "qk", "lk"))
    v, u = u_("v", B, S, E), u_("u", B, S, E)
    freqs = sd24[orc.PFX + "layers.0.rotary_pos_emb.freqs"]
    av, au = ops.cal_attention(*(t.to(dev) for t in (qq, lq, qk, lk, v, u, freqs)))
    assert rel_l2(av, fx["att_v"]) < 1e-5
    assert rel_l2(au, fx["att_u"]) < 1e-5
    # ragged / edge sizes vs the oracle: S=1, S=255, S=256, S=257, S=1000; E=128 goes through the
    # split-bf16 x6 attention GEMM (as the model's E=1024 does), E=64 through the fp32-MFMA one
    g = torch.Generator().manual_seed(3)
    for S2, E2 in ((1, 128), (255, 128), (256, 64), (257, 128), (1000, 256)):
        ts = [torch.rand(1, S2, 128, generator=g) - 0.5 for _ in range(4)] + [torch.rand(1, S2, E2, generator=g) - 0.5 for _ in range(2)]
        ov, ou = orc.cal_attention(*[t.double() for t in ts], freqs.double())
        av, au = ops.cal_attention(*(t.to(dev) for t in ts), freqs.to(dev))
        assert rel_l2(av, ov) < 1e-5 and rel_l2(au, ou) < 1e-5, S2


def test_dilated_dense_net_golden(gold, sd24, dev):
    """G3: reference DilatedDenseNet (pins the concat-grouping rule fsmn.py:92-98,110)."""
    from oracle import mossformer2_oracle as orc
    from targetdiarization_amd import ops
    from targetdiarization_amd.weights import philox_uniform
    fx = np.load(os.path.join(gold, "g3_dilated_dense_net_2x150.npz"))
    B, S, C = 2, 150, 256
    p = torch.from_numpy(philox_uniform("g2:ddn", B * S * C)).reshape(B, S, C)
    q = orc.PFX + "fsmn.0.gated_fsmn.fsmn.conv."
    args = (sd24[q + "conv1.weight"].reshape(256, 39), sd24[q + "conv2.weight"].reshape(256, 2, 39),
            torch.stack((sd24[q + "norm1.weight"], sd24[q + "norm2.weight"])),
            torch.stack((sd24[q + "norm1.bias"], sd24[q + "norm2.bias"])),
            torch.stack((sd24[q + "prelu1.weight"], sd24[q + "prelu2.weight"])))
    out = ops.dilated_dense_net(p.to(dev), *(a.to(dev) for a in args))
    assert rel_l2(out, fx["out"]) < 1e-5
    for S2 in (1, 37, 513, 1031):    # ragged lengths incl. shorter than the 39-tap window
        p2 = torch.rand(1, S2, C, generator=torch.Generator().manual_seed(S2)) - 0.5
        ref = orc.dilated_dense_net(p2.double(), {k: v.double() for k, v in sd24.items() if k.startswith(q)}, q) if S2 > 1 else None
        out = ops.dilated_dense_net(p2.to(dev), *(a.to(dev) for a in args))
        if ref is not None:
            assert rel_l2(out, ref) < 1e-5, S2
        assert torch.isfinite(out).all()


def test_taps_vs_reference_golden(gold, sep24, dev):
    """G1 taps on [1,4000]: every stage against values produced by the reference's modules."""
    from targetdiarization_amd.weights import recipe_wave
    fx = np.load(os.path.join(gold, "g1_taps_1x4000_seed0.npz"))
    x = torch.from_numpy(recipe_wave("g1:taps", 1, 4000)).to(dev)
    sep24.enable_taps(True)
    out = sep24(x)
    torch.cuda.synchronize()
    idx = torch.as_tensor(fx["tokens"]).to(dev)
    errs = {}
    for k in ("enc", "after_flash0", "after_fsmn0", "after_stack", "mask"):
        t = sep24.tap(k)
        sub = t.index_select(-2, idx)
        errs[k] = rel_l2(sub, fx[k])
        errs[k + "_sum"] = abs(float(t.double().sum().cpu()) - float(fx[k + "_sum"])) / float(fx[k + "_abssum"])
    errs["out"] = rel_l2(out, fx["out"])
    sep24.enable_taps(False)
    print(errs)
    for k, e in errs.items():
        assert e < REL_TOL, (k, e, errs)


@pytest.mark.parametrize("tag,B,T", [("1x16000", 1, 16000), ("2x8000", 2, 8000), ("1x64000", 1, 64000)])
def test_full_model_vs_reference_golden(gold, sep24, dev, tag, B, T):
    """G1: 24-block model outputs produced by the reference (recipe weights seed 0)."""
    from targetdiarization_amd.weights import recipe_wave
    fx = np.load(os.path.join(gold, "g1_mossformer2_24blk_seed0.npz"))
    x = torch.from_numpy(recipe_wave(f"g1:{tag}", B, T)).to(dev)
    out = sep24(x)
    e, c = rel_l2(out, fx[tag]), cos(out, fx[tag])
    print(tag, "rel_l2", e, "cos", c)
    assert e < REL_TOL and (1 - c) < 1e-3


def test_ragged_T_and_batch_vs_oracle(sd2, dev):
    """2-block model, T not a multiple of 8, S < 256, S crossing group edges, B>1; fp64 oracle."""
    from oracle import mossformer2_oracle as orc
    from targetdiarization_amd.separator import MossFormer2Separator
    from targetdiarization_amd.weights import recipe_wave
    sep = MossFormer2Separator(sd2, device=dev)
    sd64 = orc.cast_state_dict(sd2, torch.float64)
    for (B, T) in [(1, 16), (1, 23), (1, 2063), (2, 2300), (3, 4803), (1, 4104)]:
        x = torch.from_numpy(recipe_wave(f"rag{B}x{T}", B, T))
        ref = orc.mossformer2_forward(x.double(), sd64)
        out = sep(x.to(dev))
        assert out.shape == (B, 2, T)
        e = rel_l2(out, ref)
        assert e < REL_TOL, (B, T, e)


def test_signal_statistics_vs_oracle(sd2, dev):
    """Inputs far from the recipe noise: the split-f16 planes of v|u and lin_k use STATIC per-layer scales
    (bounds from the weights), so silence, DC, clipping, impulses and a 60 dB level change must neither
    overflow f16 nor lose accuracy against the fp64 oracle."""
    from oracle import mossformer2_oracle as orc
    from targetdiarization_amd.separator import MossFormer2Separator
    sep = MossFormer2Separator(sd2, device=dev)
    sd64 = orc.cast_state_dict(sd2, torch.float64)
    T = 4104
    t = torch.arange(T, dtype=torch.float32)
    g = torch.Generator().manual_seed(5)
    cases = {
        "silence": torch.zeros(T),
        "dc": torch.ones(T),
        "square_clipped": torch.sign(torch.sin(2 * np.pi * 220.0 * t / 16000.0)),
        "impulses": (torch.rand(T, generator=g) > 0.995).float() * 2 - (torch.rand(T, generator=g) > 0.995).float(),
        "quiet_noise": torch.randn(T, generator=g) * 1e-4,
        "loud_noise": torch.randn(T, generator=g).clamp(-1, 1),
        "step_60dB": torch.cat([torch.randn(T // 2, generator=g) * 1e-3, torch.randn(T - T // 2, generator=g)]),
    }
    x = torch.stack(list(cases.values()))
    ref = orc.mossformer2_forward(x.double(), sd64)
    out = sep(x.to(dev))
    assert torch.isfinite(out).all()
    for i, name in enumerate(cases):
        den = float(ref[i].norm())
        err = float((out[i].double().cpu() - ref[i]).norm())
        assert err <= REL_TOL * max(den, 1e-6), (name, err, den)


def test_static_scales_under_heavy_tailed_weights(sd2, dev):
    """The K-major planes of v|u and lin_k use per-layer STATIC scales derived from the weights (a true bound).  Recipe
    weights leave ~2^7 of headroom; trained checkpoints can have outlier channels that push the bound far above the typical value,
    where the f16 split keeps fewer than 22 bits.  Weights with heavy tails — outlier rows x100 in to_hidden / to_qk, OffsetScale
    gamma up to 20, PReLU slopes in [-2, 2], InstanceNorm affine x50, depthwise taps x8 on some channels — through the 2-block
    model against the fp64 oracle at the north-star tolerance; no inf / NaN; the observed headroom (f16 range / largest scaled
    value, tap "headroom") is printed per layer and must stay inside the 2^18 window in which the split keeps 22 bits."""
    from oracle import mossformer2_oracle as orc
    from targetdiarization_amd.separator import MossFormer2Separator
    from targetdiarization_amd.weights import recipe_wave
    g = torch.Generator().manual_seed(11)
    sd = {k: v.clone() for k, v in sd2.items()}
    for l in range(2):
        p = f"mask_net.mdl.intra_mdl.mossformerM.layers.{l}."
        for name, rows in ((p + "to_hidden.mdl.1.weight", 2048), (p + "to_qk.mdl.1.weight", 128)):
            idx = torch.randperm(rows, generator=g)[:5]
            sd[name][idx] *= 100.0
            sd[name.replace("weight", "bias")][idx] *= 30.0
        sd[p + "qk_offset_scale.gamma"] = sd[p + "qk_offset_scale.gamma"] * (1.0 + 7.0 * torch.rand(4, 128, generator=g))       # up to ~20
        for cw in ("to_hidden.mdl.3.sequential.1.conv.weight", "to_qk.mdl.3.sequential.1.conv.weight"):
            ch = torch.randperm(sd[p + cw].shape[0], generator=g)[:6]
            sd[p + cw][ch] *= 8.0
        q = f"mask_net.mdl.intra_mdl.mossformerM.fsmn.{l}."
        sd[q + "conv1.1.weight"] = torch.tensor([-1.7 if l == 0 else 1.9])
        for nm in ("prelu1", "prelu2"):
            sd[q + f"gated_fsmn.fsmn.conv.{nm}.weight"] = torch.rand(256, generator=g) * 4.0 - 2.0
        for nm in ("norm1", "norm2"):
            sd[q + f"gated_fsmn.fsmn.conv.{nm}.weight"] = sd[q + f"gated_fsmn.fsmn.conv.{nm}.weight"] * 50.0
            sd[q + f"gated_fsmn.fsmn.conv.{nm}.bias"] = sd[q + f"gated_fsmn.fsmn.conv.{nm}.bias"] * 50.0
    sep = MossFormer2Separator(sd, device=dev, graph_rows=0)
    sep.enable_taps(True)
    x = torch.from_numpy(recipe_wave("heavy", 2, 4803))
    out = sep(x.to(dev))
    assert torch.isfinite(out).all()
    ref64 = orc.mossformer2_forward(x.double(), orc.cast_state_dict(sd, torch.float64))
    ref32 = orc.mossformer2_forward(x, sd)
    err = float((out.double().cpu() - ref64).norm() / ref64.norm())
    err32 = float((ref32.double() - ref64).norm() / ref64.norm())
    hd = sep.tap("headroom").cpu().numpy()
    base = MossFormer2Separator(sd2, device=dev, graph_rows=0); base.enable_taps(True); base(x.to(dev))
    hd0 = base.tap("headroom").cpu().numpy()
    for l in range(2):
        print(f"layer {l}: headroom (f16 range 2^15 / largest scaled value) v|u 2^{np.log2(32768.0 / hd[l, 0]):.1f} lin_k 2^{np.log2(32768.0 / hd[l, 1]):.1f}"
              f"   (recipe weights: 2^{np.log2(32768.0 / hd0[l, 0]):.1f}, 2^{np.log2(32768.0 / hd0[l, 1]):.1f})")
    print(f"rel-L2 vs fp64 oracle: device {err:.3e}, torch fp32 restatement {err32:.3e}")
    assert (hd > 0).all() and (hd < 32768.0).all()                       # a true bound: nothing reached the f16 range
    assert (32768.0 / hd).max() < 2.0 ** 18                             # and the largest value sits inside the 22-bit window
    # the north-star tolerance, or — this model is badly conditioned by construction: the reference's own fp32 arithmetic is
    # 4e-5 from fp64 — three times the fp32 restatement's distance from fp64, whichever is larger.  (Before the gate read u in fp32
    # the device was at 1.1e-2 here: the planes' absolute precision inside the sigmoid, see EpiAttnGatePlOut.)
    assert err < max(REL_TOL, 3.0 * err32), (err, err32)


def test_batch_independence_and_determinism(sep24, dev):
    """row b of a batch == the same window alone (reference: 1.1e-6); bit-identical reruns."""
    from targetdiarization_amd.weights import recipe_wave
    x = torch.from_numpy(recipe_wave("bi", 3, 8000)).to(dev)
    full = sep24(x).clone()
    again = sep24(x).clone()
    assert torch.equal(full, again)
    single = sep24(x[1:2]).clone()
    assert rel_l2(full[1:2], single) < 1e-5


def test_full_size_properties(sep24, dev):
    """BASELINE config-2 shape (32 x 4 s): finite, batch rows independent of their neighbours,
    and the decoder's pad/trim rule (mossformer2.py:583-588) holds (last T-T_est samples zero)."""
    from targetdiarization_amd.weights import recipe_wave
    B, T = 32, 64000
    x = torch.from_numpy(recipe_wave("cfg2", B, T)).to(dev)
    out = sep24(x)
    assert out.shape == (B, 2, T) and torch.isfinite(out).all()
    one = sep24(x[5:6])
    assert rel_l2(out[5:6], one) < 1e-5
    T2 = 64003   # T_est = (S-1)*8+16 = 64000 < T -> right zero padding
    y = sep24(torch.nn.functional.pad(x[:2], (0, 3)))
    assert torch.count_nonzero(y[..., 64000:]) == 0


def test_max_window_and_properties(sep24, dev):
    """largest window the reference's plan can produce (240 000 samples -> S = 29 999, 118 groups):
    finite, deterministic, linear in nothing but consistent with the 10 s prefix property NOT holding
    (the model is sequence-global per window: SURVEY §5) and with batch independence."""
    from targetdiarization_amd.weights import recipe_wave
    x = torch.from_numpy(recipe_wave("maxwin", 1, 240000)).to(dev)
    y = sep24(x)
    assert y.shape == (1, 2, 240000) and torch.isfinite(y).all()
    y2 = sep24(torch.cat([x, x.flip(-1)], 0))
    assert rel_l2(y2[0:1], y) < 1e-5
    # re-chunking changes the result (global norms / linear attention): guard against silently windowing
    y10 = sep24(x[:, :160000])
    assert rel_l2(y[..., :160000], y10) > 1e-3


def test_cosine_scores(dev):
    from oracle import mossformer2_oracle as orc
    from targetdiarization_amd import ops
    g = torch.Generator().manual_seed(1)
    emb = torch.randn(9, 192, generator=g)
    emb[3] = 0.0
    emb[4] = -emb[0]
    ref = emb[0].clone()
    out = ops.cosine_scores(emb.to(dev), ref.to(dev)).cpu()
    for i in range(9):
        assert abs(float(out[i]) - orc.cosine_similarity(emb[i].numpy(), ref.numpy())) < 1e-6
    z = ops.cosine_scores(emb.to(dev), torch.zeros(192, device=dev)).cpu()
    assert torch.all(z == 1.0)
