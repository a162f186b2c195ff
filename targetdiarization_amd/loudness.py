"""ITU-R BS.1770-4 integrated loudness (the meter behind AudioProcessor.meter_loudness,
AudioProcessor.py:1123-1127, which calls the third-party `pyloudnorm.Meter(rate)
.integrated_loudness`).  pyloudnorm is un-vendored/unpinned (requirements.txt) and absent
here; this restates its published algorithm: K-weighting = high-shelf (+4 dB, Q=1/sqrt2,
1500 Hz) then high-pass (Q=0.5, 38 Hz) biquads, 400 ms blocks with 75 % overlap, absolute
gate -70 LUFS, relative gate -10 LU.  Host-side numpy; only decides which separated stream
is reported first.  Parity vs pyloudnorm: unpinned."""
from __future__ import annotations

import numpy as np
from scipy.signal import lfilter


def _biquad(kind: str, G: float, Q: float, fc: float, rate: float):
    A = 10 ** (G / 40.0)
    w0 = 2.0 * np.pi * (fc / rate)
    alpha = np.sin(w0) / (2.0 * Q)
    c = np.cos(w0)
    if kind == "high_shelf":
        b0 = A * ((A + 1) + (A - 1) * c + 2 * np.sqrt(A) * alpha)
        b1 = -2 * A * ((A - 1) + (A + 1) * c)
        b2 = A * ((A + 1) + (A - 1) * c - 2 * np.sqrt(A) * alpha)
        a0 = (A + 1) - (A - 1) * c + 2 * np.sqrt(A) * alpha
        a1 = 2 * ((A - 1) - (A + 1) * c)
        a2 = (A + 1) - (A - 1) * c - 2 * np.sqrt(A) * alpha
    else:  # high_pass
        b0 = (1 + c) / 2
        b1 = -(1 + c)
        b2 = (1 + c) / 2
        a0 = 1 + alpha
        a1 = -2 * c
        a2 = 1 - alpha
    return np.array([b0, b1, b2]) / a0, np.array([a0, a1, a2]) / a0


def integrated_loudness(data: np.ndarray, rate: int, block_size: float = 0.400) -> float:
    x = np.asarray(data, dtype=np.float64)
    if x.ndim == 1:
        x = x[:, None]
    n, ch = x.shape
    if n < block_size * rate:
        raise ValueError("Audio must have length greater than the block size.")
    for kind, G, Q, fc in (("high_shelf", 4.0, 1 / np.sqrt(2), 1500.0), ("high_pass", 0.0, 0.5, 38.0)):
        b, a = _biquad(kind, G, Q, fc, rate)
        x = lfilter(b, a, x, axis=0)
    G_ch = [1.0, 1.0, 1.0, 1.41, 1.41][:ch]
    T_g, step = block_size, 0.25
    T = n / rate
    nblk = int(np.round(((T - T_g) / (T_g * step))) + 1)
    z = np.zeros((ch, nblk))
    for j in range(nblk):
        lo = int(T_g * (j * step) * rate)
        hi = int(T_g * (j * step + 1) * rate)
        z[:, j] = (1.0 / (T_g * rate)) * np.sum(np.square(x[lo:hi, :]), axis=0)
    with np.errstate(divide="ignore"):
        l = -0.691 + 10.0 * np.log10(np.sum(np.array(G_ch)[:, None] * z, axis=0))
    sel = l >= -70.0
    if not np.any(sel):
        return float("-inf")
    z_avg = z[:, sel].mean(axis=1)
    gamma_r = -0.691 + 10.0 * np.log10(np.sum(np.array(G_ch) * z_avg)) - 10.0
    sel = (l > gamma_r) & (l > -70.0)
    if not np.any(sel):
        return float("-inf")
    z_avg = np.nan_to_num(z[:, sel].mean(axis=1))
    with np.errstate(divide="ignore"):
        return float(-0.691 + 10.0 * np.log10(np.sum(np.array(G_ch) * z_avg)))
