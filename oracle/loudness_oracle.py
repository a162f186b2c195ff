"""ITU-R BS.1770-4 integrated loudness, restated from the RECOMMENDATION (test infrastructure; checker only).

Independent of targetdiarization_amd/loudness.py (which restates pyloudnorm's code path, the meter behind
AudioProcessor.meter_loudness, AudioProcessor.py:1123-1127):
  * K-weighting from the standard's own coefficient table (Annex 1, Tables 1 and 2, 48 kHz).  For other sample rates
    the two biquads are re-derived the way the standard's note asks ("coefficients for other sampling rates should be
    chosen to give the same frequency response"): the analog prototype that the 48 kHz table is the bilinear image of —
    shelf f0 = 1681.9744509555 Hz, G = 3.99984385397 dB, Q = 0.70717523695; high-pass f0 = 38.1354708760 Hz,
    Q = 0.50032703732 (the closed form used by libebur128 / ffmpeg ebur128) — is transformed at the new rate.
    pyloudnorm instead designs RBJ-cookbook filters (1500 Hz, +4 dB, Q 1/sqrt2; 38 Hz, Q 0.5): a DIFFERENT derivation
    of the same curve, so agreement between the two is a real check (they differ by a few 0.01 LU).
  * the biquads run as an explicit direct-form-I recurrence in float64 (no scipy);
  * gating per Annex 1: 400 ms blocks, 75 % overlap, absolute gate -70 LKFS, relative gate -10 LU below the mean of
    the absolutely gated blocks.  Block count = floor((T - 0.4) / 0.1) + 1 as in the Recommendation (pyloudnorm rounds;
    clips in the tests are chosen so that both give the same blocks, except where a test says otherwise).
Pinned by the Recommendation's own known answer: a 997 Hz sine at 0 dB FS in one channel reads -3.01 LKFS.
"""
import math

import numpy as np

TABLE_48K = (
    ((1.53512485958697, -2.69169618940638, 1.19839281085285), (1.0, -1.69065929318241, 0.73248077421585)),   # stage 1: shelving
    ((1.0, -2.0, 1.0), (1.0, -1.99004745483398, 0.99007225036621)),                                          # stage 2: RLB high-pass
)


def k_weighting(rate: float):
    if rate == 48000:
        return TABLE_48K
    f0, G, Q = 1681.974450955533, 3.999843853973347, 0.7071752369554196
    K = math.tan(math.pi * f0 / rate)
    Vh = 10.0 ** (G / 20.0)
    Vb = Vh ** 0.4996667741545416
    a0 = 1.0 + K / Q + K * K
    s1 = (((Vh + Vb * K / Q + K * K) / a0, 2.0 * (K * K - Vh) / a0, (Vh - Vb * K / Q + K * K) / a0),
          (1.0, 2.0 * (K * K - 1.0) / a0, (1.0 - K / Q + K * K) / a0))
    f0, Q = 38.13547087602444, 0.5003270373238773
    K = math.tan(math.pi * f0 / rate)
    a0 = 1.0 + K / Q + K * K
    s2 = ((1.0, -2.0, 1.0), (1.0, 2.0 * (K * K - 1.0) / a0, (1.0 - K / Q + K * K) / a0))
    return s1, s2


def biquad(x, b, a):
    y = np.zeros_like(x)
    x1 = x2 = y1 = y2 = 0.0
    b0, b1, b2 = b
    _, a1, a2 = a
    for n in range(x.shape[0]):
        xn = x[n]
        yn = b0 * xn + b1 * x1 + b2 * x2 - a1 * y1 - a2 * y2
        x2, x1, y2, y1 = x1, xn, y1, yn
        y[n] = yn
    return y


def integrated_loudness(x, rate: int) -> float:
    """mono float signal -> LKFS (-inf when nothing passes the gates); ValueError below one 400 ms block"""
    y = np.asarray(x, dtype=np.float64).reshape(-1)
    nb = int(round(0.4 * rate))
    if y.shape[0] < nb:
        raise ValueError("shorter than one gating block")
    for b, a in k_weighting(rate):
        y = biquad(y, b, a)
    hop = int(round(0.1 * rate))
    count = (y.shape[0] - nb) // hop + 1
    z = np.array([np.mean(y[j * hop: j * hop + nb] ** 2) for j in range(count)])
    with np.errstate(divide="ignore"):
        l = -0.691 + 10.0 * np.log10(z)
    keep = l > -70.0
    if not keep.any():
        return float("-inf")
    gamma_r = -0.691 + 10.0 * math.log10(z[keep].mean()) - 10.0
    keep &= l > gamma_r
    if not keep.any():
        return float("-inf")
    return -0.691 + 10.0 * math.log10(z[keep].mean())
