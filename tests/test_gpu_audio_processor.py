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


def test_low_gpu_ram_vad_plan_matches_reference_loop(sd2):
    """low_gpu_ram (AudioProcessor.py:892-917): 1 s windows inside silero-VAD frames (a plug-in here), zeros before / between the
    frames, nothing after the last one.  Oracle = the reference's loop restated with batch 1 per window."""
    from oracle import mossformer2_oracle as orc
    from targetdiarization_amd.audio_processor import AudioProcessor
    from targetdiarization_amd.loudness import integrated_loudness
    from targetdiarization_amd.weights import recipe_wave
    audio = recipe_wave("lowram", 1, 120000)[0]
    frames = [[3000, 40000], [52000, 100001]]
    ap = AudioProcessor(is_separate_audio=True, separater_state_dict=sd2, cuda_device=0, verbose_log=False, silero_vad=lambda a: frames)
    a, b = ap.separate_speaker(audio, 16000, low_gpu_ram=True)
    s1, s2 = np.array([], np.float32), np.array([], np.float32)
    for i, (f0, f1) in enumerate(frames):                          # :908-948
        if f0 > s1.shape[0]:
            gap = np.zeros(f0 if i == 0 else f0 - frames[i - 1][1], np.float32)
            s1, s2 = np.concatenate([s1, gap]), np.concatenate([s2, gap])
        for (w0, w1) in orc.window_plan(f1 - f0, 16000, f0):
            y = orc.mossformer2_forward(torch.from_numpy(audio[w0:w1].copy())[None], sd2)[0].numpy()
            s1, s2 = np.concatenate([s1, y[0]]), np.concatenate([s2, y[1]])
    if round(integrated_loudness(s1, 16000), 1) < round(integrated_loudness(s2, 16000), 1):
        s1, s2 = s2, s1
    assert a.shape == s1.shape == (100001,)                        # the tail after the last VAD frame is dropped, like the reference
    assert np.all(a[:3000] == 0) and np.all(a[40000:52000] == 0)
    assert np.linalg.norm(a - s1) / np.linalg.norm(s1) < 1e-4 and np.linalg.norm(b - s2) / np.linalg.norm(s2) < 1e-4


def test_separate_speaker_resamples_in_and_out(sd2):
    """a rate other than 16 kHz is resampled in and back out (:889-891, :953-955)"""
    from targetdiarization_amd.audio_processor import AudioProcessor
    from targetdiarization_amd.weights import recipe_wave
    audio8k = recipe_wave("sr8k", 1, 20000)[0]
    ap = AudioProcessor(is_separate_audio=True, separater_state_dict=sd2, cuda_device=0, verbose_log=False)
    a, b = ap.separate_speaker(audio8k, 8000)
    x16 = ap.audio_resample(audio8k, 8000, 16000, output_audio_only=True)
    r1, r2 = ap.separate_speaker(x16, 16000)
    e1 = ap.audio_resample(r1, 16000, 8000, output_audio_only=True)
    assert a.shape == e1.shape == (20000,) and a.dtype == np.float32
    assert np.linalg.norm(a - e1) / np.linalg.norm(e1) < 1e-5
