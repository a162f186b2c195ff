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
