"""ORACLE — test infrastructure, not product code.

CPU restatement (PyTorch, fp32/fp64) of the Kaldi-compatible fbank front-ends the path uses
(SURVEY.md §8 row a2, Appendix B.2).  The arithmetic lives in THIRD-PARTY packages that are
neither vendored nor pinned nor installed here:
  * torchaudio.compliance.kaldi.fbank  (transitive dep of modelscope / funasr; unpinned)
    — call sites: modelscope ERes2NetV2 pipeline reached from TargetASR.py:161, and
      funasr.frontends.wav_frontend.WavFrontend reached from ASRProcessor.py:424
  * funasr WavFrontend.apply_lfr / apply_cmvn (requirements.txt:24, unpinned)
=> PARITY UNPINNED: this file restates their published algorithms; nothing in the reference
tree pins the numbers.  Only tests/, smoke() and bench.py's cpu_baseline may import it.
"""
from __future__ import annotations

import math

import torch

EPS = 1.1920928955078125e-07


def mel_banks(num_bins=80, n_fft=512, sample_rate=16000.0, low=20.0, high=0.0, dtype=torch.float64):
    """torchaudio.compliance.kaldi.get_mel_banks (no VTLN): [num_bins, n_fft/2] triangles on the
    mel scale 1127 ln(1+f/700), evaluated at FFT bin centre frequencies."""
    nyq = 0.5 * sample_rate
    if high <= 0.0:
        high += nyq
    num_fft_bins = n_fft // 2
    bin_w = sample_rate / n_fft
    mel = lambda f: 1127.0 * torch.log(1.0 + f / 700.0)
    mlow, mhigh = mel(torch.tensor(low, dtype=dtype)), mel(torch.tensor(high, dtype=dtype))
    delta = (mhigh - mlow) / (num_bins + 1)
    b = torch.arange(num_bins, dtype=dtype)[:, None]
    left, center, right = mlow + b * delta, mlow + (b + 1) * delta, mlow + (b + 2) * delta
    m = mel(bin_w * torch.arange(num_fft_bins, dtype=dtype))[None, :]
    up = (m - left) / (center - left)
    down = (right - m) / (right - center)
    return torch.clamp(torch.minimum(up, down), min=0.0)          # [num_bins, n_fft/2]


def window_fn(kind: str, n=400, dtype=torch.float64):
    if kind == "povey":
        return torch.hann_window(n, periodic=False, dtype=dtype).pow(0.85)
    if kind == "hamming":
        return torch.hamming_window(n, periodic=False, alpha=0.54, beta=0.46, dtype=dtype)
    raise ValueError(kind)


def kaldi_fbank(wave: torch.Tensor, window="povey", scale=1.0, num_mel_bins=80):
    """kaldi.fbank(wave[1,N]*scale, num_mel_bins=80, frame_length=25, frame_shift=10, dither=0,
    window_type=window, snip_edges=True) -> [F,80], F = 1 + (N-400)//160.  1-D wave in."""
    dt = wave.dtype
    x = wave * scale
    N = x.shape[0]
    if N < 400:
        return torch.empty(0, num_mel_bins, dtype=dt)
    F = 1 + (N - 400) // 160
    fr = x.unfold(0, 400, 160)[:F].clone()                         # strided frames
    fr = fr - fr.mean(dim=1, keepdim=True)                         # remove_dc_offset
    prev = torch.cat((fr[:, :1], fr[:, :-1]), dim=1)               # replicate-left
    fr = fr - 0.97 * prev                                          # preemphasis
    fr = fr * window_fn(window, 400, dt)
    fr = torch.nn.functional.pad(fr, (0, 112))                     # round_to_power_of_two -> 512
    spec = torch.fft.rfft(fr).abs().pow(2.0)                       # [F,257] power
    mel = torch.nn.functional.pad(mel_banks(num_mel_bins, dtype=dt), (0, 1))   # [80,257]
    e = spec @ mel.T
    return torch.clamp(e, min=EPS).log()


def sv_features(wave: torch.Tensor):
    """ERes2NetV2 front-end [upstream-recall 3D-Speaker / modelscope]: fbank80 (povey, input in
    [-1,1]) minus the per-utterance mean over frames."""
    f = kaldi_fbank(wave, "povey", 1.0)
    return f - f.mean(dim=0, keepdim=True)


def apply_lfr(mat: torch.Tensor, m=7, n=6):
    """funasr WavFrontend.apply_lfr: left-pad (m-1)//2 copies of frame 0, stack m frames every n,
    right-pad by repeating the last frame -> [ceil(T/n), m*D]."""
    T = mat.shape[0]
    T_lfr = int(math.ceil(T / n))
    mat = torch.cat((mat[:1].repeat((m - 1) // 2, 1), mat), dim=0)
    T = T + (m - 1) // 2
    rows = []
    for i in range(T_lfr):
        if m <= T - i * n:
            rows.append(mat[i * n:i * n + m].reshape(1, -1))
        else:
            fr = mat[i * n:].reshape(-1)
            for _ in range(m - (T - i * n)):
                fr = torch.hstack((fr, mat[-1]))
            rows.append(fr.reshape(1, -1))
    return torch.vstack(rows)


def asr_features(wave: torch.Tensor, cmvn_shift: torch.Tensor, cmvn_scale: torch.Tensor):
    """funasr WavFrontend (dither 0): fbank80(hamming, wave*32768) -> LFR 7/6 -> (x+shift)*scale."""
    f = kaldi_fbank(wave, "hamming", 32768.0)
    l = apply_lfr(f, 7, 6)
    return (l + cmvn_shift) * cmvn_scale
