"""CPU tests of the round-2 host logic and oracles: the MDX outer chunker (AudioProcessor.py:666-706) as index arithmetic, the
polyphase filter design against scipy, the CIF oracle's invariants, strict checkpoint/config handling."""
import numpy as np
import pytest
import torch


def _reference_outer_chunker(total, chunk, margin):
    """AudioProcessor.py:662-706 restated literally on indices: which samples of the input end up where in the output"""
    if total <= chunk:
        return [(0, total)]
    margin = min(margin, chunk)
    segs, cursor, k = {}, 0, 0
    while cursor < total:
        start = max(0, cursor - (0 if k == 0 else margin))
        chunk_end = cursor + chunk
        last = chunk_end >= total
        end = total if last else min(chunk_end + margin, total)
        segs[cursor] = (start, end)
        k += 1
        cursor += chunk
        if last:
            break
    kept, keys = [], list(segs)
    for i, key in enumerate(keys):
        a, b = segs[key]
        n = b - a
        t0 = 0 if i == 0 else min(margin, n // 2)
        t1 = None if i == len(keys) - 1 else -min(margin, n // 2)
        idx = list(range(a, b))[t0:t1]
        kept.append((idx[0], idx[-1] + 1) if idx else (a, a))
    return kept


@pytest.mark.parametrize("total", [1000, 661500, 661501, 882000, 1323000, 1367100, 2000000, 705600])
def test_mdx_segments_tile_the_audio_like_the_reference(total):
    from targetdiarization_amd.audio_processor import AudioProcessor
    segs = AudioProcessor.mdx_segments(total, 661500, 44100)
    kept = [(a + t0, (b - t1) if t1 is not None else b) for (a, b, t0, t1) in segs]
    assert kept == _reference_outer_chunker(total, 661500, 44100)
    assert kept[0][0] == 0 and kept[-1][1] == total
    tiles = all(kept[i][1] == kept[i + 1][0] for i in range(len(kept) - 1))
    last_len = segs[-1][1] - segs[-1][0]
    # reference quirk kept: the trims are clamped to HALF the segment, so a last segment shorter than two margins leaves a hole
    # between the last two pieces (the output is then shorter than the input); otherwise the trimmed pieces tile [0, total)
    assert tiles == (len(segs) == 1 or last_len >= 2 * 44100)
    if total == 661501:
        assert kept == [(0, 617401), (639450, 661501)]


@pytest.mark.parametrize("up,down", [(441, 160), (160, 441), (1, 2), (147, 160)])
def test_resample_filter_is_scipys_default(up, down):
    from scipy.signal import firwin
    from targetdiarization_amd.ops import resample_filter
    h, half = resample_filter(up, down)
    ref = firwin(2 * half + 1, 1.0 / max(up, down), window=("kaiser", 5.0)) * up
    assert half == 10 * max(up, down) and np.abs(h - ref).max() <= 1e-6 * np.abs(ref).max()


def test_cif_oracle_invariants():
    """funasr cif(): one fire per crossing of the threshold; the fired frames' weights are a partition of the alphas"""
    from oracle import paraformer_oracle as po
    from targetdiarization_amd.weights import recipe_paraformer_decoder_state_dict
    sd = {k: v.double() for k, v in recipe_paraformer_decoder_state_dict(0, 1).items()}
    g = torch.Generator().manual_seed(1)
    enc = torch.randn(3, 80, 512, generator=g, dtype=torch.float64)
    hidden, alphas = po.cif_alphas(enc, sd)
    assert hidden.shape == (3, 81, 512) and alphas.shape == (3, 81) and torch.all(alphas[:, -1] == 0.45) and torch.all(hidden[:, -1] == 0)
    fired, fires = po.cif(hidden, alphas)
    for b in range(3):
        n = fired[b].shape[0]
        assert n == int(torch.floor(alphas[b].sum() + 1e-9)) and n == int((fires[b] >= 1.0).sum())
        # with constant hidden = 1 every fired frame integrates exactly one unit of alpha
        ones, _ = po.cif(torch.ones(1, 81, 4, dtype=torch.float64), alphas[b:b + 1])
        assert torch.allclose(ones[0], torch.ones_like(ones[0]), atol=1e-12)
    res, logits = po.paraformer_decode(enc, sd, 1)
    assert logits.shape[0] == 3 and all(len(ids) == len(pk) for ids, pk in res)


def test_from_pretrain_checks_model_name_and_args(tmp_path):
    from targetdiarization_amd import _lib
    from targetdiarization_amd.separator import MossFormer2Separator
    p = tmp_path / "best_model.pth"
    torch.save({"model_name": "ConvTasNet", "state_dict": {}}, p)
    with pytest.raises(_lib.TdxError, match="MossFormer2 only"):
        MossFormer2Separator.from_pretrain(str(p), device="cuda:0")
    torch.save({"model_name": "MossFormer2", "state_dict": {}}, p)
    with pytest.raises(_lib.TdxError, match="num_spks"):
        MossFormer2Separator.from_pretrain(str(p), device="cuda:0", num_spks=3)
    with pytest.raises(_lib.TdxError, match="unsupported MossFormer2 constructor argument 'causal'"):
        MossFormer2Separator.from_pretrain(str(p), device="cuda:0", causal=True)


def test_get_speaker_embedding_input_types_without_gpu(tmp_path):
    """str / list inputs resolve to waveforms on the host side (16 kHz mono PCM16 .wav); anything else is a loud error"""
    import wave
    from targetdiarization_amd import _lib
    from targetdiarization_amd.speaker import SpeakerEmbedder
    p = tmp_path / "a.wav"
    with wave.open(str(p), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes((np.arange(1600) % 100).astype(np.int16).tobytes())
    x = SpeakerEmbedder._read_wav(str(p))
    assert x.dtype == np.float32 and x.shape == (1600,) and abs(x[1] - 1 / 32768.0) < 1e-9
    q = tmp_path / "b.wav"
    with wave.open(str(q), "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(44100)
        w.writeframes(np.zeros(200, np.int16).tobytes())
    with pytest.raises(_lib.TdxError, match="16 kHz mono"):
        SpeakerEmbedder._read_wav(str(q))
