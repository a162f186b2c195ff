"""-m gpu: Paraformer SANM encoder vs its oracle (parity unpinned: third-party
architecture restated from upstream, recipe weights).  Tolerance 1e-4 rel-L2 / 1e-3 cosine."""
import numpy as np
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
    # (5, 301): > 512 rows = the large-launch path (planes-out FFN with segmented row scales, no split-K), a ragged last tile
    for nb, shapes in ((3, [(1, 1), (2, 37), (1, 128), (3, 129), (1, 500), (5, 301)]), (50, [(2, 167)])):
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


# ---- N2: CIF predictor + NAR SANM decoder (funasr, third-party: parity unpinned; oracle restates the published code) ----
def _dec_setup(B, T, blocks, seed=0):
    from targetdiarization_amd.paraformer import ParaformerDecoder
    from targetdiarization_amd.weights import recipe_paraformer_decoder_state_dict
    sd = recipe_paraformer_decoder_state_dict(0, blocks)
    g = torch.Generator().manual_seed(seed + B * 1000 + T)
    enc = torch.randn(B, T, 512, generator=g)
    return sd, enc, ParaformerDecoder(sd, device="cuda:0")


@pytest.mark.parametrize("B,T", [(1, 7), (2, 50), (3, 167), (1, 500)])
def test_cif_predictor_vs_oracle(B, T):
    """alphas, token counts, firing frames and the fired (acoustic embedding) frames against the sequential reference loop"""
    from oracle import paraformer_oracle as po
    sd, enc, dec = _dec_setup(B, T, 1)
    alphas, emb, counts, peaks = dec.predict(enc.cuda())
    hidden, a_ref = po.cif_alphas(enc.double(), {k: v.double() for k, v in sd.items()})
    assert float((alphas.double().cpu() - a_ref).abs().max()) < 2e-5
    # the integrate-and-fire loop itself on the DEVICE's alphas (fp32, the reference's operation order): exact decisions
    fired, fires = po.cif(torch.cat((enc, torch.zeros(B, 1, 512)), 1), alphas.cpu())
    for b in range(B):
        n = fired[b].shape[0]
        pk = torch.nonzero(fires[b] >= 1.0)[:, 0].tolist()
        assert peaks[b, :n].cpu().tolist() == pk and int(peaks[b, n]) == -1 if n < T + 1 else True
        assert int(counts[b]) == int(torch.floor(alphas[b].cpu().sum()))
        e = emb[b, :n].cpu()
        assert float((e - fired[b]).norm() / fired[b].norm()) < 1e-5
        assert float(emb[b, n:].abs().max()) == 0.0 if n < T + 1 else True


@pytest.mark.parametrize("B,T,blocks", [(2, 50, 2), (3, 167, 2), (2, 120, 16)])
def test_nar_decoder_vs_oracle(B, T, blocks):
    """decoder logits path: token ids equal the fp64 oracle's argmax wherever its top-2 margin is not razor thin, scores close"""
    from oracle import paraformer_oracle as po
    sd, enc, dec = _dec_setup(B, T, blocks, seed=3)
    res = dec.decode(enc.cuda())
    sd64 = {k: v.double() for k, v in sd.items()}
    ref, logits = po.paraformer_decode(enc.double(), sd64, blocks)
    lp = torch.log_softmax(logits, -1)
    for b in range(B):
        ids_ref, peaks_ref = ref[b]
        assert len(res[b]["token_ids"]) == len(ids_ref) and len(ids_ref) > 0
        top2 = lp[b, : len(ids_ref)].topk(2, dim=-1).values
        margin = (top2[:, 0] - top2[:, 1]).numpy()
        agree = np.array(res[b]["token_ids"]) == np.array(ids_ref)
        assert agree[margin > 1e-3].all(), (b, np.where(~agree)[0][:5], margin[~agree][:5])
        assert agree.mean() > 0.98
        sc = np.array(res[b]["scores"]); sr = top2[:, 0].numpy()
        assert np.abs(sc[agree] - sr[agree]).max() < 1e-3
        ts = res[b]["timestamp"]
        assert len(ts) == len(ids_ref) and all(s <= e for s, e in ts) and all(ts[i][1] <= ts[i + 1][0] + 1e-9 for i in range(len(ts) - 1))
        assert [int(round((p + 1) * 60.0)) for p in peaks_ref] == [e for _, e in ts][: len(peaks_ref)]


def test_asr_processor_text_through_device_decoder():
    """ASRProcessor.asr_detection (ASRProcessor.py:373-442) end to end on the device: wav -> fbank/LFR -> encoder -> CIF -> decoder ->
    tokens + timestamps in the reference's result format (timestamps in seconds, paired with the tokens)"""
    from targetdiarization_amd.asr_processor import ASRProcessor
    from targetdiarization_amd.weights import recipe_paraformer_decoder_state_dict, recipe_paraformer_state_dict, recipe_wave
    sd = dict(recipe_paraformer_state_dict(0, 2)); sd.update(recipe_paraformer_decoder_state_dict(0, 2))
    asr = ASRProcessor(is_asr=True, asr_state_dict=sd, cuda_device=0, verbose_log=False, token_list=[f"t{i}" for i in range(8404)])
    wav = recipe_wave("asr", 1, 48000)[0]
    res = asr.asr_detection(wav, asr_engine="paraformer")
    assert isinstance(res, list) and len(res) == 1
    r = res[0]
    assert r["text"] and r["language"] in ("zh", "en")
    assert len(r["timestamp"]) == len(r["text"].split(" "))
    tok, (s, e) = r["timestamp"][0]
    assert tok.startswith("t") and 0.0 <= s <= e
    assert asr.asr_detection(wav, output_text_only=True) == r["text"]
