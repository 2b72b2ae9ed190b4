"""Float64 evaluation of the per-hop loop body (app3.py:178-217) on the SAME constants the fp32 paths use
(TEST INFRASTRUCTURE; DSP stages PARITY UNPINNED like oracle/dsp_np64.py, which does the arithmetic).

Purpose: a yardstick that tells rounding from error.  The GPU path and the fp32 CPU oracle both differ from
the exact result of the algorithm; 32 Griffin-Lim iterations amplify those differences by a frame-dependent
factor.  Evaluating the same algorithm in float64 -- same fp32 window, same fp32 mel
filterbank, same initial phases, the model in float64 -- gives the point both should be close to, so a test
can ask "is the GPU as close to it as the reference's own fp32 arithmetic is?" instead of comparing two fp32
results with each other.

``model64(x, hx) -> (out, hx)`` is the GRUUNet2 forward in float64: the reference's own class with
``.double()`` when fixtures are generated (oracle/make_f64_golden.py), or oracle/model_ref.forward on float64
weights.
"""
from __future__ import annotations

import numpy as np

from . import dsp_np64


def _stft(x, n_fft, hop, w):
    xp = dsp_np64.reflect_pad(x, n_fft // 2)
    n_cols = 1 + (xp.shape[1] - n_fft) // hop
    return np.stack([np.fft.rfft(xp[:, t * hop:t * hop + n_fft] * w, axis=1) for t in range(n_cols)], axis=2)


def _istft(spec, n_fft, hop, w):
    b, _, n_cols = spec.shape
    total = n_fft + hop * (n_cols - 1)
    acc, env = np.zeros((b, total)), np.zeros(total)
    for t in range(n_cols):
        col = spec[:, :, t].copy()
        col[:, 0] = col[:, 0].real
        col[:, -1] = col[:, -1].real
        acc[:, t * hop:t * hop + n_fft] += np.fft.irfft(col, n=n_fft, axis=1) * w
        env[t * hop:t * hop + n_fft] += w * w
    lo, hi = n_fft // 2, total - n_fft // 2
    return acc[:, lo:hi] / env[lo:hi]


def process_frame64(frames, hx0, model64, window, fb, init_angles, n_fft, hop, n_iter=32, momentum=0.99):
    """frames (B, n_fft) fp32 values, window (n_fft,) and fb (K, M) the fp32 constants of the fp32 paths, init_angles
    (B, K, 3) complex64 -> dict of float64 arrays: out (B, n_fft), hx, model_input (B, 3, M), predicted_diff, lin_mag (B, K, 3), peak."""
    x = np.asarray(frames, np.float64)
    w = np.asarray(window, np.float64)
    fb = np.asarray(fb, np.float64)
    peak = np.abs(x).max(axis=1)                                       # P1  app3.py:181-186
    peak = np.where(peak > 1e-6, peak, 1.0)
    x = x / peak[:, None] * w                                          # P2  app3.py:188
    spec = _stft(x, n_fft, hop, w)                                     # P4  app3.py:191 (the transform windows again)
    mel = np.log1p(np.einsum("bkt,km->bmt", np.abs(spec), fb))         # P5  app3.py:192-193
    model_input = np.ascontiguousarray(mel.transpose(0, 2, 1))         # P6  app3.py:195
    diff, hx = model64(model_input, hx0)                               # P7  app3.py:200-201
    rec = model_input - diff                                           # P8  app3.py:203-205
    rec = np.where(rec >= 0, rec, 0.2 * rec)
    mel_mag = np.maximum(np.expm1(rec.transpose(0, 2, 1)), 0.0)        # P9  app3.py:206-208
    lin = np.maximum(dsp_np64.inverse_mel_scale(mel_mag, fb), 0.0)     # P10 app3.py:210-211
    ang = np.asarray(init_angles, np.complex128)                       # P11 app3.py:213
    mu = momentum / (1.0 + momentum)
    prev = np.zeros_like(ang)
    for _ in range(n_iter):
        rebuilt = _stft(_istft(ang * lin, n_fft, hop, w), n_fft, hop, w)
        ang = rebuilt - mu * prev
        ang = ang / (np.abs(ang) + 1e-16)
        prev = rebuilt
    y = _istft(ang * lin, n_fft, hop, w)
    return dict(out=y * peak[:, None], hx=hx, model_input=model_input, predicted_diff=diff, lin_mag=lin, peak=peak)    # app3.py:217
