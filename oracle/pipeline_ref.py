"""CPU restatement of the per-hop loop body of DenoisingAudioProcessor.recv
(app3.py:178-226 == app2.py:185-233), batched over B independent streams
(TEST INFRASTRUCTURE; DSP stages PARITY UNPINNED, model stage pinned --
see oracle/__init__.py).

Stage ids P1..P12 are SURVEY.md section 8(a).
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn.functional as F

from . import dsp_ref, model_ref


@dataclass
class Params:
    sample_rate: int
    n_fft: int
    hop: int
    n_mels: int

    @property
    def n_stft(self) -> int:
        return self.n_fft // 2 + 1

    @property
    def num_compressed_bins(self) -> int:
        return self.n_mels // 16


PARAMS_S = Params(16000, 1024, 512, 80)     # BASELINE.json synthetic config
PARAMS_R1 = Params(48000, 1536, 768, 64)    # app3.py:29-33
PARAMS_R2 = Params(48000, 1024, 512, 64)    # server.py:166-170


def analysis(frames: torch.Tensor, p: Params, fb: torch.Tensor):
    """P1,P2,P4,P5,P6: raw frames (B,N) -> model_input (B,3,M), peak (B,).
    app3.py:179-195."""
    peak = frames.abs().amax(dim=1)                                   # P1  app3.py:181
    ok = peak > 1e-6
    peak = torch.where(ok, peak, torch.ones_like(peak))               #     app3.py:182-186
    x = frames / peak[:, None]
    x = x * dsp_ref.hann(p.n_fft, frames.dtype)                       # P2  app3.py:188
    spec = dsp_ref.spectrogram(x, p.n_fft, p.hop)                     # P4  app3.py:191
    mel = dsp_ref.mel_scale(spec.abs(), fb).log1p()                   # P5  app3.py:192-193
    return mel.transpose(-1, -2).contiguous(), peak                   # P6  app3.py:195


def residual_to_mel_mag(model_input: torch.Tensor, predicted_diff: torch.Tensor) -> torch.Tensor:
    """P8,P9: (B,3,M),(B,3,M) -> (B,M,3).  app3.py:203-208."""
    rec = F.leaky_relu(model_input - predicted_diff, negative_slope=0.2)
    return torch.clamp(torch.expm1(rec.transpose(-1, -2)), min=0)


def synthesis(mel_mag: torch.Tensor, p: Params, fb: torch.Tensor, init_angles: torch.Tensor | None,
              generator: torch.Generator | None = None):
    """P10,P11: (B,M,3) -> linear magnitude (B,K,3), waveform (B,N).  app3.py:210-213."""
    lin = torch.clamp(dsp_ref.inverse_mel_scale(mel_mag, fb), min=0)
    y = dsp_ref.griffinlim(lin, p.n_fft, p.hop, init_angles=init_angles, generator=generator)
    return lin, y


def process_frame(sd: dict, frames: torch.Tensor, hx: torch.Tensor, p: Params, fb: torch.Tensor,
                  init_angles: torch.Tensor | None = None, generator: torch.Generator | None = None):
    """P1..P11 + the `* peak` of P12 for a batch of frames.

    frames (B,N) raw fp32, hx (B,17,C) -> dict(out (B,N), hx, model_input,
    predicted_diff, lin_mag, peak)."""
    model_input, peak = analysis(frames, p, fb)
    diff, hx = model_ref.forward(sd, model_input, hx)                 # P7  app3.py:200-201
    mel_mag = residual_to_mel_mag(model_input, diff)
    lin, y = synthesis(mel_mag, p, fb, init_angles, generator)
    return dict(out=y * peak[:, None], hx=hx, model_input=model_input, predicted_diff=diff,
                mel_mag=mel_mag, lin_mag=lin, peak=peak)


class StreamRef:
    """Streaming state of B streams: input ring, output overlap-add buffer, hx.
    app3.py:130-133 (state), 178, 219-226 (P12)."""

    def __init__(self, sd: dict, p: Params, batch: int):
        self.sd, self.p, self.b = sd, p, batch
        self.fb = dsp_ref.melscale_fbanks(p.n_stft, p.n_mels, p.sample_rate)
        self.inbuf = torch.zeros(batch, 0)
        self.ola = torch.zeros(batch, p.n_fft)
        self.hx = torch.zeros(batch, 17, p.num_compressed_bins)

    def push(self, chunk: torch.Tensor, init_angles_per_hop=None) -> torch.Tensor:
        """Append (B, n) samples; run as many hops as fit; return (B, hops*hop) output."""
        p = self.p
        self.inbuf = torch.cat([self.inbuf, chunk], dim=1)
        outs, i = [], 0
        while self.inbuf.size(1) >= p.n_fft:                          # app3.py:178
            ia = None if init_angles_per_hop is None else init_angles_per_hop[i]
            r = process_frame(self.sd, self.inbuf[:, :p.n_fft], self.hx, p, self.fb, init_angles=ia)
            self.hx = r["hx"]
            outs.append(self.ola[:, :p.hop].clone())                  # app3.py:219-220
            self.ola = torch.cat([self.ola[:, p.hop:], torch.zeros(self.b, p.hop)], dim=1)   # 222-223
            self.ola += r["out"]                                      # app3.py:224
            self.inbuf = self.inbuf[:, p.hop:]                        # app3.py:226
            i += 1
        return torch.cat(outs, dim=1) if outs else torch.zeros(self.b, 0)
