"""-m gpu: the AudioProcessor.separate_speaker drop-in (AudioProcessor.py:885-956) vs the
oracle driving the same window plan with batch = 1 (what the reference does)."""
import os
import wave as wavmod

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _oracle_separate(audio, sd):
    from oracle import mossformer2_oracle as orc
    from targetdiarization_amd.loudness import integrated_loudness
    s1, s2 = [], []
    for a, b in orc.window_plan(len(audio)):
        y = orc.mossformer2_forward(torch.from_numpy(audio[a:b].copy())[None], sd)[0].numpy()
        s1.append(y[0]); s2.append(y[1])
    s1, s2 = np.concatenate(s1), np.concatenate(s2)
    if round(integrated_loudness(s1, 16000), 1) < round(integrated_loudness(s2, 16000), 1):
        s1, s2 = s2, s1
    return s1, s2


def test_separate_speaker_on_reference_asset(gold, sd2):
    """config-1 input (assets/chat_mix.wav, 138 634 samples -> ONE 8.665 s window, S=17 328)
    through a 2-block model (CPU oracle stays fast)."""
    from targetdiarization_amd.audio_processor import AudioProcessor
    with wavmod.open(os.path.join(gold, "chat_mix.wav"), "rb") as w:
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)
    audio = (pcm.astype(np.float32) / 32768.0)
    ap = AudioProcessor(is_separate_audio=True, separater_state_dict=sd2, cuda_device=0, verbose_log=False)
    assert ap.is_separate_audio
    a, b = ap.separate_speaker(audio, 16000)
    assert a.shape == b.shape == audio.shape and a.dtype == np.float32
    ra, rb = _oracle_separate(audio, sd2)
    for x, r in ((a, ra), (b, rb)):
        assert np.linalg.norm(x - r) / np.linalg.norm(r) < 1e-4


def test_window_plan_batching_matches_per_window(sd2):
    """400 001 samples -> windows 160000,160000,80001: the two equal windows go through ONE
    batched launch; result must equal the per-window oracle."""
    from targetdiarization_amd.audio_processor import AudioProcessor
    from targetdiarization_amd.weights import recipe_wave
    audio = recipe_wave("ap", 1, 400001)[0]
    ap = AudioProcessor(is_separate_audio=True, separater_state_dict=sd2, cuda_device=0, verbose_log=False)
    a, b = ap.separate_speaker(audio)
    ra, rb = _oracle_separate(audio, sd2)
    assert np.linalg.norm(a - ra) / np.linalg.norm(ra) < 1e-4
    assert np.linalg.norm(b - rb) / np.linalg.norm(rb) < 1e-4


def test_degrade_semantics(sd2):
    from targetdiarization_amd.audio_processor import AudioProcessor
    ap = AudioProcessor(is_separate_audio=False, verbose_log=False)
    x = np.zeros(100, dtype=np.float32)
    a, b = ap.separate_speaker(x)                    # disabled separator returns (audio, audio)  :886-888
    assert a is x and b is x
    ap2 = AudioProcessor(is_separate_audio=True, separater_weights_folder="/nonexistent", verbose_log=False)
    assert ap2.is_separate_audio is False            # init failure flips the flag, never raises   :189-193
