"""CPU tests of the rest of N1 against goldens minted by EXECUTING the reference's own methods (oracle/make_goldens_n1b.py):
  * TargetASR.get_target_embedding (TargetASR.py:166-258)            -> target_asr.target_embedding_from_audio
  * TargetDiarization.sd_result_to_target_embedding (:551-578)      -> target_diarization.TargetDiarization
  * TargetDiarization.sd_result_to_asr_audio (:716-820), both the timestamp and the no-timestamp branch
The stand-ins for the neighbouring objects (embedding, VAD, loudness control, recogniser, separation) are the ones the golden
script used, restated here; the HDBSCAN labels come from the product's own implementation (targetdiarization_amd/clustering.py;
the goldens used scikit-learn's: same published algorithm, "parity unpinned" against the absent hdbscan package)."""
import json
import os

import numpy as np
import pytest

import targetdiarization_amd.target_diarization as td_mod
from targetdiarization_amd.clustering import hdbscan_labels
from targetdiarization_amd.target_asr import target_embedding_from_audio

SR = 16000


def embed(audio):
    a = np.asarray(audio, dtype=np.float64).reshape(-1)
    e = np.zeros(192, dtype=np.float32)
    e[0] = np.sqrt(np.mean(a * a)) if a.size else 0.0
    e[1] = a.size / SR / 100.0
    e[2] = 1.0
    if a.size and a[0] == 7.0:
        e[5] = np.nan
    return e


def vad(audio):
    a = np.abs(np.asarray(audio, dtype=np.float32).reshape(-1))
    n = a.shape[0] // 160
    act = a[: n * 160].reshape(n, 160).max(axis=1) > 1e-4 if n else np.zeros(0, bool)
    out, start = [], None
    for i, v in enumerate(list(act) + [False]):
        if v and start is None:
            start = i
        if not v and start is not None:
            out.append([round(start * 0.01, 3), round(i * 0.01, 3)]); start = None
    return out


def loudness_control(audio, sampling_rate=SR):
    if audio.shape[0] / sampling_rate < 0.4:
        return audio
    r = float(np.sqrt(np.mean(audio.astype(np.float64) ** 2)))
    return audio if r == 0.0 else (audio * (0.05 / r)).astype(np.float32)


def synth(seed, segments):
    rng = np.random.default_rng(seed)
    return np.concatenate([(rng.uniform(-1, 1, int(s * SR)) * lv).astype(np.float32) for s, lv in segments])


def asr_stub(audio, with_timestamp, language):
    n = int(audio.shape[0] / SR / 0.5)
    stamps = [(f"t{k}", [round(0.5 * k, 3), round(0.5 * k + 0.3, 3)]) for k in range(n)]
    return " ".join(t for t, _ in stamps), (stamps if with_timestamp else []), language


def separate_stub(clip):
    d = round(clip.shape[0] / SR, 3)
    k = int(clip.shape[0]) % 3
    if k == 0:
        return []
    res = [([0.1, round(d - 0.1, 3)], (clip * 0.5).astype(np.float32))]
    if k == 2:
        res.append(([0.2, d], (clip * 0.25).astype(np.float32)))
    return res


def fingerprint(a):
    a = np.asarray(a, dtype=np.float64)
    return [int(a.shape[0]), float(a.sum()), float((a * a).sum()), float(a[0]) if a.size else 0.0, float(a[-1]) if a.size else 0.0]


@pytest.fixture(scope="module")
def gold_te(gold):
    return json.load(open(os.path.join(gold, "n1_target_embedding.json")))


def test_get_target_embedding_vs_reference(gold_te):
    cases = gold_te["get_target_embedding"]
    assert len(cases) == 256
    for c in cases:
        audios = [synth(100 + c["spec"] * 10 + j, [tuple(x) for x in segs]) for j, segs in enumerate(c["clips"])]
        if c["nan_clip"] >= 0:
            audios[c["nan_clip"]][0] = 7.0
        if c["as_list"]:
            store = {f"f{j}": a for j, a in enumerate(audios)}
            arg, reader = list(store), (lambda p, _s=store: _s[p].copy())
        else:
            arg, reader = audios[0], None
        got = target_embedding_from_audio(arg, lambda clips: np.stack([embed(x) for x in clips]), vad, loudness_control, read_audio=reader,
                                          is_preprocess=c["pre"], is_cluster=c["cluster"], audio_input_type=c["type"],
                                          output_embedding_list=c["list"])
        want = c["out"]
        if c["list"] and isinstance(want, list) and (not want or isinstance(want[0], list)):
            assert isinstance(got, list) and len(got) == len(want), c
            for g, w in zip(got, want):
                assert np.allclose(np.asarray(g, np.float64)[:3], w, rtol=1e-6, atol=1e-9), c
        else:                                             # a single vector (the mean, the one embedding, or the zero fallback: also returned when `list` is set)
            assert not isinstance(got, list) and np.asarray(got).shape == (192,), c
            assert np.allclose(np.asarray(got, np.float64)[:3], want, rtol=1e-6, atol=1e-9), c


def test_hdbscan_vs_sklearn_where_importable():
    skl = pytest.importorskip("sklearn.cluster")
    if not hasattr(skl, "HDBSCAN"):
        pytest.skip("scikit-learn without HDBSCAN")
    rng = np.random.default_rng(2)

    def canon(labels):
        m, out = {}, []
        for x in labels:
            out.append(-1 if x == -1 else m.setdefault(int(x), len(m)))
        return out
    for trial in range(120):
        n, d, k = int(rng.integers(3, 40)), int(rng.choice([2, 8, 192])), int(rng.integers(1, 5))
        cent = rng.normal(0, rng.uniform(0.5, 4), (k, d))
        X = cent[rng.integers(0, k, n)] + rng.normal(0, rng.uniform(0.05, 1.0), (n, d))
        want = skl.HDBSCAN(min_cluster_size=2, metric="euclidean", copy=True).fit_predict(X)      # the reference's setting (TargetASR.py:238)
        assert canon(hdbscan_labels(X, 2)) == canon(want), trial
    assert hdbscan_labels(np.zeros((0, 4))).shape == (0,) and list(hdbscan_labels(np.ones((2, 4)))) == [-1, -1]
    assert set(hdbscan_labels(np.ones((5, 4)))) <= {-1, 0}            # identical points: one cluster or noise, never an error


class _Spk:
    def get_speaker_embeddings(self, clips):
        return np.stack([embed(c) for c in clips]) if clips else np.zeros((0, 192), np.float32)

    def get_speaker_embedding(self, a):
        return embed(a)


class _HotPath:
    def __init__(self, *a, **k):
        self.spk, self.asr, self.dec, self.ap = _Spk(), object(), None, None

    def encode_streams(self, lines):
        return list(lines)


@pytest.fixture
def make_td(monkeypatch):
    monkeypatch.setattr(td_mod, "HotPath", _HotPath)

    def factory(**kw):
        td = td_mod.TargetDiarization(vad=vad, **kw)
        td.audio_loudness_control = lambda a, sampling_rate=SR, target_loudness=-23.0: loudness_control(a, sampling_rate)
        return td
    return factory


def test_sd_result_to_target_embedding_vs_reference(gold_te, make_td):
    audio = synth(7, [(2.0, 0.1), (1.0, 0), (3.0, 0.3), (0.5, 0), (2.5, 0.2), (1.0, 0.05)])
    td = make_td()
    assert len(gold_te["sd_to_target_embedding"]) == 9
    for c in gold_te["sd_to_target_embedding"]:
        sd = {k: [tuple(x) for x in v] for k, v in c["sd"].items()}
        omap = [[tuple(x) for x in m] for m in c["omap"]]
        spk, emb = td.sd_result_to_target_embedding(audio.copy(), sd, omap, target_spk=c["target_spk"])
        assert spk == c["spk"], c
        assert np.allclose(np.asarray(emb, np.float64)[:3], c["emb"], rtol=1e-6, atol=1e-9), c


def test_sd_result_to_asr_audio_vs_reference_both_branches(gold, make_td):
    cases = json.load(open(os.path.join(gold, "n1_asr_audio.json")))["asr_audio"]
    assert len(cases) == 72 and any(not c["with_ts"] for c in cases)
    audio = synth(11, [(12.0, 0.1)])
    for c in cases:
        td = make_td(decoder=lambda enc, _c=c: asr_stub(enc, _c["with_ts"], _c["lang"]), punctuation=lambda t: t + "." if t else t)
        td._separate_overlaps = lambda a, ranges, emb: [separate_stub(td.split_audio_by_time(a, s, e)) for (s, e) in ranges]
        sd = {k: [tuple(x) for x in v] for k, v in c["sd"].items()}
        omap = [[tuple(x) for x in m] for m in c["omap"]]
        emb = embed(audio) if c["have_emb"] else None
        got = td.sd_result_to_asr_audio(audio.copy(), sd, omap, c["target_spk"], emb)
        key = lambda r: (r["timerange"][0], r["speaker"], r["timerange"][1])       # (the reference walks list(set(speakers)): order of equal starts is hash order)
        got = sorted(got, key=key)
        want = sorted(c["out"], key=key)
        assert len(got) == len(want), c
        for g, w in zip(got, want):
            assert g["speaker"] == w["speaker"] and g["type"] == w["type"] and g["text"] == w["text"], (c["with_ts"], c["lang"], g["text"], w)
            assert np.allclose(g["timerange"], w["timerange"], atol=1e-9)
            fp = fingerprint(g["audio"])
            assert fp[0] == w["audio"][0] and np.allclose(fp[1:], w["audio"][1:], rtol=1e-5, atol=1e-6), (c, fp, w["audio"])
