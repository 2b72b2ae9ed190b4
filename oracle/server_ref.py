"""CPU restatement of the request-loop body of the reference's socket server (server.py:199-217)
(TEST INFRASTRUCTURE; DSP stages PARITY UNPINNED, model stage pinned -- see oracle/__init__.py).

    abs_spec = T0(X); phase = angle; magn = abs            server.py:207-209
    log_mel_mag = M0T(magn).log1p()                         server.py:210
    out, hx = model(log_mel_mag.T, hx)                      server.py:212
    out = leaky_relu(out.T, negative_slope=0) * 3           server.py:213   (= relu * 3)
    hx = hx * 0.9                                           server.py:214
    O = M0I((log_mel_mag - out).exp() - 1)                  server.py:215
    O = I0(torch.polar(O, phase))                           server.py:216

Parameters R2 (server.py:166-170): n_fft 1024, hop 512, 64 mels, 48 kHz; checkpoint GRUUNet2-good (server.py:151).
The chunk X may have any length L (hop-multiple here); the output has hop*(L//hop) samples (length=None).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import dsp_ref, model_ref
from .pipeline_ref import PARAMS_R2, Params


def process_chunk(sd: dict, x: torch.Tensor, hx: torch.Tensor | None, p: Params = PARAMS_R2, fb: torch.Tensor | None = None):
    """x (B, L) -> dict(out (B, hop*(T-1)), hx, log_mel (B,M,T), model_out (B,T,M), lin (B,K,T))."""
    if fb is None:
        fb = dsp_ref.melscale_fbanks(p.n_stft, p.n_mels, p.sample_rate)
    spec = dsp_ref.spectrogram(x, p.n_fft, p.hop)
    phase, magn = spec.angle(), spec.abs()
    log_mel = dsp_ref.mel_scale(magn, fb).log1p()
    model_out, hx = model_ref.forward(sd, log_mel.transpose(-1, -2), hx, num_compressed_bins=p.num_compressed_bins)
    out = F.leaky_relu(model_out.transpose(-1, -2), negative_slope=0) * 3
    hx = hx * 0.9
    lin = dsp_ref.inverse_mel_scale((log_mel - out).exp() - 1, fb)
    wave = dsp_ref.inverse_spectrogram(torch.polar(lin, phase), p.n_fft, p.hop)
    return dict(out=wave, hx=hx, log_mel=log_mel, model_out=model_out, lin=lin, spec=spec)
