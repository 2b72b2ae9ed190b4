"""Independent float64 numpy implementation of the DSP stages (TEST
INFRASTRUCTURE).  Written separately from oracle/dsp_ref.py -- explicit
framing, numpy.fft, explicit overlap-add -- so the two can be checked against
each other (double-entry bookkeeping; the reference pins nothing here, see
oracle/__init__.py).  Stage semantics: SURVEY.md Appendix B; call sites
app3.py:135-153,191-193,210,213.
"""
from __future__ import annotations

import numpy as np


def hann(n: int) -> np.ndarray:
    k = np.arange(n, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)


def reflect_pad(x: np.ndarray, p: int) -> np.ndarray:
    """(B, L) -> (B, L+2p), mirror without repeating the edge sample."""
    left = x[:, 1:p + 1][:, ::-1]
    right = x[:, -p - 1:-1][:, ::-1]
    return np.concatenate([left, x, right], axis=1)


def stft(x: np.ndarray, n_fft: int, hop: int) -> np.ndarray:
    """(B, L) -> (B, K, T) complex128; centred, reflect padded, Hann, one-sided."""
    x = np.asarray(x, dtype=np.float64)
    xp = reflect_pad(x, n_fft // 2)
    n_cols = 1 + (xp.shape[1] - n_fft) // hop
    w = hann(n_fft)
    cols = [np.fft.rfft(xp[:, t * hop:t * hop + n_fft] * w, axis=1) for t in range(n_cols)]
    return np.stack(cols, axis=2)


def istft(spec: np.ndarray, n_fft: int, hop: int) -> np.ndarray:
    """(B, K, T) -> (B, hop*(T-1)); irfft, window, overlap-add, divide by the
    window-square envelope, trim n_fft/2 from both ends."""
    spec = np.asarray(spec, dtype=np.complex128)
    b, _, n_cols = spec.shape
    w = hann(n_fft)
    total = n_fft + hop * (n_cols - 1)
    acc = np.zeros((b, total))
    env = np.zeros(total)
    for t in range(n_cols):
        col = spec[:, :, t].copy()
        col[:, 0] = col[:, 0].real          # C2R transforms ignore Im of DC/Nyquist
        col[:, -1] = col[:, -1].real
        acc[:, t * hop:t * hop + n_fft] += np.fft.irfft(col, n=n_fft, axis=1) * w
        env[t * hop:t * hop + n_fft] += w * w
    lo, hi = n_fft // 2, total - n_fft // 2
    return acc[:, lo:hi] / env[lo:hi]


def mel_fbanks(n_freqs: int, n_mels: int, sample_rate: int) -> np.ndarray:
    """HTK triangles, unnormalised, f in [0, sr//2].  (n_freqs, n_mels) float64."""
    f_max = float(sample_rate // 2)
    freqs = np.linspace(0.0, f_max, n_freqs)
    mel_max = 2595.0 * np.log10(1.0 + f_max / 700.0)
    edges = 700.0 * (10.0 ** (np.linspace(0.0, mel_max, n_mels + 2) / 2595.0) - 1.0)
    fb = np.zeros((n_freqs, n_mels))
    for m in range(n_mels):
        lo, ce, hi = edges[m], edges[m + 1], edges[m + 2]
        rising = (freqs - lo) / (ce - lo)
        falling = (hi - freqs) / (hi - ce)
        fb[:, m] = np.maximum(0.0, np.minimum(rising, falling))
    return fb


def mel_scale(mag: np.ndarray, fb: np.ndarray) -> np.ndarray:
    """(B, K, T) x (K, M) -> (B, M, T)."""
    return np.einsum("bkt,km->bmt", np.asarray(mag, np.float64), np.asarray(fb, np.float64))


def inverse_mel_scale(mel: np.ndarray, fb: np.ndarray) -> np.ndarray:
    """Minimum-norm least squares of fb^T s = mel, then relu.  (B, M, T) -> (B, K, T)."""
    pinv = np.linalg.pinv(np.asarray(fb, np.float64).T)          # (K, M)
    return np.maximum(np.einsum("km,bmt->bkt", pinv, np.asarray(mel, np.float64)), 0.0)


def griffinlim(mag: np.ndarray, n_fft: int, hop: int, init_angles: np.ndarray, n_iter: int = 32,
               momentum: float = 0.99) -> np.ndarray:
    """Fast Griffin-Lim with momentum/(1+momentum); init_angles (B,K,T) complex."""
    mag = np.asarray(mag, np.float64)
    ang = np.asarray(init_angles, np.complex128)
    mu = momentum / (1.0 + momentum)
    prev = np.zeros_like(ang)
    for _ in range(n_iter):
        rebuilt = stft(istft(ang * mag, n_fft, hop), n_fft, hop)
        ang = rebuilt - mu * prev
        ang = ang / (np.abs(ang) + 1e-16)
        prev = rebuilt
    return istft(ang * mag, n_fft, hop)
