"""CPU tests of the drop-in host classes that mirror the reference's method surface:
TargetASR.cosine_similarity edge cases (G5), ASRProcessor result post-processing and the
print-and-degrade convention."""
import numpy as np


def test_cosine_similarity_edge_cases():
    from targetdiarization_amd.target_asr import TargetASR
    z = np.zeros(192, dtype=np.float32); a = np.ones(192, dtype=np.float32)
    assert TargetASR.cosine_similarity(z, a) == 1.0 and TargetASR.cosine_similarity(a, z) == 1.0      # TargetASR.py:145-146
    assert TargetASR.cosine_similarity(a, -a) == 0.0                                                  # clipped to [0, 1]
    assert abs(TargetASR.cosine_similarity(a, 3 * a) - 1.0) < 1e-6
    b = np.zeros(192, dtype=np.float32); b[0] = 1.0
    assert abs(TargetASR.cosine_similarity(a, b) - 1.0 / np.sqrt(192)) < 1e-6
    assert isinstance(TargetASR.cosine_similarity(a, b), float)


def test_asr_processor_degrades_and_postprocesses(capsys):
    from targetdiarization_amd.asr_processor import ASRProcessor
    ap = ASRProcessor(is_asr=False)
    assert ap.asr_detection(np.zeros(16000, dtype=np.float32)) == []
    assert ap.asr_detection(np.zeros(16000, dtype=np.float32), output_text_only=True) == ""
    assert "haven't been loaded" in capsys.readouterr().out
    res = [{"key": "k", "text": "ni hao ma", "timestamp": [[0, 240], [240, 505], [505, 1000], [1000, 1234]]}]
    out = ASRProcessor.paraformer_postprocess(res, no_punc=False, punctuation_restore=lambda t: t + "?", detect_language=lambda t: "zh")
    assert out[0]["timestamp"] == [("ni", [0.0, 0.24]), ("hao", [0.24, 0.505]), ("ma", [0.505, 1.0]), ("", [1.0, 1.234])]
    assert out[0]["text"] == "ni hao ma?" and out[0]["language"] == "zh"
    out2 = ASRProcessor.paraformer_postprocess([{"text": "hello"}], True, lambda t: t, ASRProcessor.detect_language)
    assert out2[0]["language"] == "en" and "timestamp" not in out2[0]
    assert ASRProcessor.detect_language("\u4f60\u597d ok") == "zh" and ASRProcessor.detect_language("") == "zh"


def test_target_asr_without_models_raises_keyerror_like_reference():
    import pytest
    from targetdiarization_amd.target_asr import TargetASR
    t = TargetASR(cuda_device=0)
    with pytest.raises(KeyError):
        t.get_speaker_embedding(np.zeros(16000, dtype=np.float32))


def test_read_audio_and_is_same_person():
    """host pieces of the drop-in surface: TargetDiarization.read_audio (PCM16 .wav by path / bytes / file object, arrays pass
    through) and TargetASR.is_same_person (mean of the known embeddings, threshold, verbose form: TargetASR.py:491-505)"""
    import io
    import os
    from targetdiarization_amd.target_asr import TargetASR
    from targetdiarization_amd.target_diarization import TargetDiarization
    path = os.path.join(os.path.dirname(__file__), "golden", "chat_mix.wav")
    x, sr = TargetDiarization.read_audio(path)
    assert sr == 16000 and x.dtype == np.int16 and x.shape == (138634,)
    raw = open(path, "rb").read()
    for src in (raw, io.BytesIO(raw)):
        y, sr2 = TargetDiarization.read_audio(src)
        assert sr2 == 16000 and np.array_equal(x, y)
    arr = np.zeros(10, np.float32)
    assert TargetDiarization.read_audio(arr, 8000) == (arr, 8000) or TargetDiarization.read_audio(arr, 8000)[1] == 8000
    t = TargetASR()                                   # no models: the embedding arithmetic is host-side
    a, b = np.array([1.0, 0.0, 0.0]), np.array([0.0, 1.0, 0.0])
    assert t.is_same_person(a, a) is True and t.is_same_person(a, b) is False
    assert t.is_same_person([a, b], a, threshold=0.7) is True          # mean (0.5, 0.5, 0) vs a: cos = 0.707
    assert t.is_same_person([a, b], a, threshold=0.71) is False
    assert t.is_same_person(a, b, verbose_result=True) == {"is_same": False, "score": 0.0}
    assert t.is_same_person(np.zeros(3), b) is True                     # an all-zero embedding scores 1.0 (:145-146)
