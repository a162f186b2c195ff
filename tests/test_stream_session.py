"""CPU tests of the streaming session (N4, targetdiarization_amd/target_diarization_stream.py): the reference's buffer rules
(TargetDiarizationStream.py:81-171) and per-buffer flow (:174-258) with a FAKE hot path (two "speakers" = two sine frequencies;
embedding = band energies; separation = band split; ASR text = one letter per 0.1 s) — host logic only, no kernels."""
import numpy as np
import pytest

import targetdiarization_amd.target_diarization as td_mod

SR = 16000


def tone(f, seconds, amp=0.1):
    t = np.arange(int(seconds * SR)) / SR
    return (amp * np.sin(2 * np.pi * f * t)).astype(np.float32)


def band(audio, lo, hi):
    spec = np.fft.rfft(audio)
    fr = np.fft.rfftfreq(audio.shape[0], 1.0 / SR)
    spec[(fr < lo) | (fr > hi)] = 0
    return np.fft.irfft(spec, audio.shape[0]).astype(np.float32)


class FakeSpk:
    def get_speaker_embedding(self, audio):
        e = np.zeros(192, np.float32)
        e[0] = float(np.sqrt(np.mean(band(audio, 100, 300) ** 2))); e[1] = float(np.sqrt(np.mean(band(audio, 400, 600) ** 2)))
        return e

    def get_speaker_embeddings(self, clips):
        return np.stack([self.get_speaker_embedding(c) for c in clips])

    def cosine_scores(self, embs, ref):
        embs = np.asarray(embs, np.float64); ref = np.asarray(ref, np.float64)
        n = np.linalg.norm(embs, axis=1) * np.linalg.norm(ref)
        return np.where(n == 0, 1.0, np.clip(embs @ ref / np.maximum(n, 1e-30), 0, 1))


class FakeAP:
    is_denoise_vocal = False

    def separate_speaker(self, audio, sr):
        a, b = band(audio, 100, 300), band(audio, 400, 600)
        return (a, b) if np.abs(a).sum() >= np.abs(b).sum() else (b, a)

    def audio_resample(self, audio, orig, target, output_audio_only=False):
        return (audio, target)


class FakeHotPath:
    def __init__(self, *a, **k):
        self.spk, self.ap, self.asr, self.dec = FakeSpk(), FakeAP(), object(), None

    def separate(self, clips):
        return [(band(c, 100, 300), band(c, 400, 600)) for c in clips]

    def encode_streams(self, audios):
        return list(audios)


def energy_vad(audio):
    """speech = 20 ms frames above 1e-3 RMS, merged"""
    n = audio.shape[0] // 320
    if n == 0:
        return []
    act = np.sqrt((audio[: n * 320].reshape(n, 320) ** 2).mean(axis=1)) > 1e-3
    out, start = [], None
    for i, a in enumerate(list(act) + [False]):
        if a and start is None:
            start = i
        if not a and start is not None:
            out.append([round(start * 0.02, 3), round(i * 0.02, 3)]); start = None
    return out


@pytest.fixture
def make(monkeypatch):
    monkeypatch.setattr(td_mod, "HotPath", FakeHotPath)
    from targetdiarization_amd.target_diarization_stream import TargetDiarizationStream

    def factory(**kw):
        kw.setdefault("decoder", lambda enc: ("x" * max(1, int(np.abs(enc).sum() > 0) * (enc.shape[0] // 1600)), []))
        return TargetDiarizationStream(vad=energy_vad, stream_vad=energy_vad, od_pipeline=lambda a: [], **kw)
    return factory


def collect(s, chunks, **kw):
    return list(s.infer_stream(iter(chunks), **kw))


def test_bootstrap_labels_and_clock(make):
    s = make(max_buffer_duration=30.0)
    a, b = tone(200, 1.0), tone(500, 1.0)
    sil = np.zeros(SR // 2, np.float32)
    # speaker A, pause (releases by rule 3/"speech appears complete"), speaker B, pause
    out = collect(s, [np.concatenate([a, sil]), np.concatenate([b, sil])])
    assert [r[0] for r in out] == ["1", "1"]
    segs = [r[1][0] for r in out]
    assert [x["speaker"] for x in segs] == ["1", "0"]              # the first released buffer defined the target
    assert all(x["type"] == "single" and x["text"] for x in segs) and all("audio" not in x for x in segs)
    assert segs[0]["timerange"][0] == pytest.approx(1.5, abs=0.05)  # the reference's clock: ranges start at the END of their buffer
    assert segs[1]["timerange"][0] == pytest.approx(3.0, abs=0.05)
    assert s.prev_asr_text == segs[1]["text"] and not s.vad_buffer


def test_rule1_max_buffer_and_rule5_wait(make):
    s = make(max_buffer_duration=2.0)
    chunks = [tone(200, 0.5) for _ in range(6)]                    # continuous speech of one speaker: nothing releases it but rule 1
    out = collect(s, chunks)
    # released at 2.0 s (four chunks), the remaining 1.0 s at the end of the stream
    assert len(out) == 2 and all(r[1][0]["speaker"] == "1" for r in out)
    assert len(out[0][1][0]["text"]) == 20 and len(out[1][1][0]["text"]) == 10


def test_rule4_speaker_change_releases_the_buffer(make):
    s = make(max_buffer_duration=30.0, similarity_threshold=0.4)
    out = collect(s, [tone(200, 0.6), tone(200, 0.6), tone(500, 0.6)])
    # the third chunk is another speaker: the buffer (all three chunks, like the reference) is released at that point, nothing is left for the end
    assert len(out) == 1 and len(out[0][1][0]["text"]) == 18


def test_target_clip_sets_loudness_gate_and_silence_rule(make):
    s = make(max_buffer_duration=30.0, loudness_diff_threshold=12.0)
    target = tone(200, 5.0)
    quiet = tone(200, 0.5, amp=0.0005)                             # > 12 LU below the target: treated as silence
    out = collect(s, [tone(200, 1.0), quiet, tone(500, 1.0), quiet], target_file=target, output_target_audio=True)
    assert s.system_loudness_diff != 0.0
    assert [r[1][0]["speaker"] for r in out] == ["1", "0"]
    assert out[0][2] is not None and out[0][2].dtype == np.float32   # target audio of the target's segment
    assert out[1][2] is None or not np.any(out[1][2])                # the other speaker contributes silence only


def test_overlap_buffer_goes_through_separation(make):
    s = make(max_buffer_duration=30.0)
    s.od_pipeline = lambda a: [(0.0, 1.0, "SPEAKER_00"), (0.2, 1.0, "SPEAKER_01")] if a.shape[0] > SR else []
    s.target_embedding = FakeSpk().get_speaker_embedding(tone(200, 1.0))
    s.system_loudness_diff = 0.0
    mix = tone(200, 1.2) + tone(500, 1.2, amp=0.05)
    out = collect(s, [np.concatenate([mix, np.zeros(SR // 2, np.float32)])])
    seg = out[0][1][0]
    assert seg["type"] == "overlap" and seg["speaker"] == "1"


def test_without_buffering_every_chunk_is_processed(make):
    s = make(is_vad_buffer=False)
    out = collect(s, [tone(200, 0.3), tone(200, 0.5), tone(200, 0.5)])     # 0.3 s < 0.4 s: dropped (:203-204)
    assert len(out) == 2
    # once the loudness gate is armed a chunk shorter than the meter's 0.4 s block raises, as pyloudnorm does in the reference (:87)
    with pytest.raises(ValueError):
        collect(s, [tone(200, 0.5), tone(200, 0.3)])
