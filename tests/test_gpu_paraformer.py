"""-m gpu: Paraformer SANM encoder vs its oracle (parity unpinned: third-party
architecture restated from upstream, recipe weights).  Tolerance 1e-4 rel-L2 / 1e-3 cosine."""
import pytest
import torch

pytestmark = pytest.mark.gpu
dev = torch.device("cuda:0")


def rel_l2(a, b):
    a = a.detach().double().cpu().reshape(-1); b = torch.as_tensor(b).double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm())


def test_encoder_vs_oracle_small_and_full():
    from oracle import paraformer_oracle as po
    from targetdiarization_amd.paraformer import ParaformerEncoder
    from targetdiarization_amd.weights import recipe_paraformer_state_dict
    for nb, shapes in ((3, [(1, 1), (2, 37), (1, 128), (3, 129), (1, 500)]), (50, [(2, 167)])):
        sd = recipe_paraformer_state_dict(0, nb)
        enc = ParaformerEncoder(sd, dev)
        sd64 = {k: v.double() for k, v in sd.items()}
        for (B, T) in shapes:
            x = torch.randn(B, T, 560, generator=torch.Generator().manual_seed(T))
            ref = po.sanm_encoder_forward(x.double(), sd64)
            out = enc.encode(x.to(dev))
            assert out.shape == (B, T, 512)
            e = rel_l2(out, ref)
            assert e < 1e-4, (nb, B, T, e)
        del enc


def test_wave_to_encoder_end_to_end():
    """wav -> fbank(hamming, x32768) -> LFR/CMVN -> encoder, vs the chained oracles."""
    from oracle import frontend_oracle as fo
    from oracle import paraformer_oracle as po
    from targetdiarization_amd.paraformer import ParaformerEncoder
    from targetdiarization_amd.weights import recipe_paraformer_state_dict, recipe_wave
    sd = recipe_paraformer_state_dict(0, 2)
    g = torch.Generator().manual_seed(5)
    shift = -8.0 + torch.randn(560, generator=g); scale = 0.2 + 0.1 * torch.rand(560, generator=g)
    enc = ParaformerEncoder(sd, dev, cmvn_shift=shift, cmvn_scale=scale)
    wav = torch.from_numpy(recipe_wave("pfwav", 2, 48000))
    out = enc(wav.to(dev))
    sd64 = {k: v.double() for k, v in sd.items()}
    for b in range(2):
        feats = fo.asr_features(wav[b].double(), shift.double(), scale.double())
        ref = po.sanm_encoder_forward(feats[None], sd64)[0]
        assert rel_l2(out[b], ref) < 1e-4
