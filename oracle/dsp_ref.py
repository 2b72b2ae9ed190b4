"""torch-CPU restatement of the torchaudio 2.6.0 transforms the reference calls
(TEST INFRASTRUCTURE; PARITY UNPINNED -- see oracle/__init__.py).

The reference constructs these at app3.py:135-153 (server.py:173-176):
  Spectrogram(power=None, n_fft, win_length=n_fft, hop_length, window_fn=hann)
  MelScale(n_mels, n_stft, sample_rate)
  InverseMelScale(n_mels, n_stft, sample_rate)
  GriffinLim(n_fft, win_length=n_fft, hop_length, window_fn=hann, power=1.0)
and calls them at app3.py:191-193, 210, 213.  torchaudio (requirements.txt:4,
==2.6.0) is not vendored and not installed, so its published algorithm is
restated here over first-party torch ops.
"""
from __future__ import annotations

import math

import torch


def hann(n: int, dtype=torch.float32) -> torch.Tensor:
    """torch.hann_window(n) (periodic): 0.5 - 0.5 cos(2 pi k / n).  app3.py:155."""
    return torch.hann_window(n, dtype=dtype)


def spectrogram(x: torch.Tensor, n_fft: int, hop: int) -> torch.Tensor:
    """Spectrogram(power=None, center=True, pad_mode='reflect', onesided=True,
    normalized=False).  (B, L) -> (B, n_fft/2+1, 1+L/hop) complex.  app3.py:135-139,191."""
    return torch.stft(x, n_fft, hop_length=hop, win_length=n_fft, window=hann(n_fft, x.dtype),
                      center=True, pad_mode="reflect", normalized=False, onesided=True,
                      return_complex=True)


def inverse_spectrogram(spec: torch.Tensor, n_fft: int, hop: int) -> torch.Tensor:
    """InverseSpectrogram / torch.istft(center=True, length=None).  server.py:174,216."""
    real_dtype = torch.float64 if spec.dtype == torch.complex128 else torch.float32
    return torch.istft(spec, n_fft, hop_length=hop, win_length=n_fft, window=hann(n_fft, real_dtype),
                       center=True, normalized=False, onesided=True, length=None)


def melscale_fbanks(n_freqs: int, n_mels: int, sample_rate: int, f_min: float = 0.0,
                    f_max: float | None = None) -> torch.Tensor:
    """HTK triangular filterbank, norm=None.  (n_freqs, n_mels) fp32.
    Defaults as MelScale(n_mels, sample_rate, n_stft): f_min=0, f_max=sr//2.  app3.py:140-143."""
    f_max = float(sample_rate // 2) if f_max is None else f_max
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)          # (n_freqs, n_mels+2)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))


def mel_scale(mag: torch.Tensor, fb: torch.Tensor) -> torch.Tensor:
    """MelScale.forward: (.., K, T) -> (.., M, T) = (mag^T @ fb)^T.  app3.py:193."""
    return torch.matmul(mag.transpose(-1, -2), fb.to(mag.dtype)).transpose(-1, -2)


def inverse_mel_scale(mel: torch.Tensor, fb: torch.Tensor) -> torch.Tensor:
    """InverseMelScale.forward (2.6.0, driver='gels'):
    relu(lstsq(fb^T (1,M,K), mel (B,M,T)).solution) -> (B, K, T).  app3.py:210."""
    shape = mel.shape
    mel3 = mel.reshape(-1, shape[-2], shape[-1])
    sol = torch.linalg.lstsq(fb.to(mel.dtype).transpose(-1, -2)[None], mel3, driver="gels").solution
    return torch.relu(sol).reshape(shape[:-2] + (fb.shape[0], shape[-1]))


def griffinlim(mag: torch.Tensor, n_fft: int, hop: int, n_iter: int = 32, momentum: float = 0.99,
               init_angles: torch.Tensor | None = None, power: float = 1.0,
               generator: torch.Generator | None = None) -> torch.Tensor:
    """GriffinLim(power=1, n_iter=32, momentum=0.99, rand_init=True, length=None).
    (B, K, T) magnitude -> (B, hop*(T-1)) waveform.  app3.py:149-153, 213.

    ``init_angles`` (B, K, T) complex replaces the reference's
    ``torch.rand(shape, dtype=complex64)`` draw so that runs are comparable."""
    cdtype = torch.complex128 if mag.dtype == torch.float64 else torch.complex64
    window = hann(n_fft, mag.dtype)
    m = momentum / (1.0 + momentum)
    shape = mag.shape
    spec = mag.reshape(-1, shape[-2], shape[-1]).pow(1.0 / power)
    if init_angles is None:
        angles = torch.rand(spec.shape, dtype=cdtype, generator=generator)
    else:
        angles = init_angles.reshape(spec.shape).to(cdtype)
    tprev = torch.tensor(0.0, dtype=spec.dtype)
    for _ in range(n_iter):
        inverse = torch.istft(angles * spec, n_fft, hop_length=hop, win_length=n_fft, window=window, length=None)
        rebuilt = torch.stft(inverse, n_fft, hop_length=hop, win_length=n_fft, window=window, center=True,
                             pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
        angles = rebuilt
        if m:
            angles = angles - tprev * m
        angles = angles / (angles.abs() + 1e-16)
        tprev = rebuilt
    wave = torch.istft(angles * spec, n_fft, hop_length=hop, win_length=n_fft, window=window, length=None)
    return wave.reshape(shape[:-2] + wave.shape[-1:])
