"""Equivalents of the torchaudio transforms the reference constructs around the model
(app3.py:135-153, server.py:173-176), with the constructor keywords used at those sites:

    Spectrogram(power=None, n_fft, win_length, hop_length, window_fn)         app3.py:135-139
    MelScale(n_mels, n_stft, sample_rate)                                     app3.py:140-143
    InverseMelScale(n_mels, n_stft, sample_rate)                              app3.py:145-148
    GriffinLim(n_fft, win_length, hop_length, window_fn, power=1.0)           app3.py:149-153
    InverseSpectrogram(n_fft, win_length, hop_length)                         server.py:174

Each ``forward`` is one HIP kernel launch through the C ABI; inputs must be float32 (complex64 for
InverseSpectrogram) CUDA tensors shaped as the reference shapes them.  Internally every spectral
tensor is stored [stream][column][bin]; the (.., bins, columns) tensors the reference's layout
calls for are returned as transposed views, which is also what MelScale returns in torchaudio.
No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
import weakref

import torch
from torch import nn

from . import _lib

_N_FFTS = (1024, 1536)   # the one-wave FFT kernels: 512 = 8*8*8 and 768 = 4*4*4*12 complex points (hop = n_fft/2)


def melscale_fbanks(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> torch.Tensor:
    """HTK triangular filterbank, norm=None, in float32 with the operation order of
    torchaudio.functional.melscale_fbanks.  (n_freqs, n_mels).
    Construction-time host code: runs once per transform object."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + (f_min / 700.0))
    m_max = 2595.0 * math.log10(1.0 + (f_max / 700.0))
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down_slopes = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up_slopes = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down_slopes, up_slopes))


class DspPlan:
    """Owns one ``dn_dsp*`` (twiddles, window, banded filterbank, pseudo-inverse) on one device."""

    def __init__(self, device: torch.device, sample_rate: int, n_fft: int, hop: int, n_mels: int = 0,
                 fb: torch.Tensor | None = None, window: torch.Tensor | None = None, pinv: torch.Tensor | None = None):
        if not torch.device(device).type == "cuda":
            raise RuntimeError("DSP plans live on a 'cuda' device; there is no CPU path in this package")
        self.lib = _lib.get_lib()
        self.device = torch.device(device)
        self.sample_rate, self.n_fft, self.hop, self.n_mels = sample_rate, n_fft, hop, n_mels
        self.n_stft = n_fft // 2 + 1
        cfg = _lib.DspCfg(int(sample_rate), int(n_fft), int(hop), int(n_mels))
        keep = []

        def host(t, shape):
            if t is None:
                return None
            t = t.detach().to("cpu", torch.float32).contiguous()
            if tuple(t.shape) != shape:
                raise ValueError(f"expected shape {shape}, got {tuple(t.shape)}")
            keep.append(t)
            return C.c_void_p(t.data_ptr())

        handle = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        with torch.cuda.device(idx):
            self.lib.check(self.lib.dn_dsp_create(C.byref(cfg), host(fb, (self.n_stft, n_mels)), host(pinv, (self.n_stft, n_mels)),
                                                  host(window, (n_fft,)), C.byref(handle)))
        self.handle = handle
        self._fin = weakref.finalize(self, self.lib.dn_dsp_destroy, handle)

    def tables(self):
        fb = torch.empty(self.n_stft, max(self.n_mels, 1))
        pinv = torch.empty(self.n_stft, max(self.n_mels, 1))
        win = torch.empty(self.n_fft)
        self.lib.check(self.lib.dn_dsp_get_tables(self.handle, fb.data_ptr(), pinv.data_ptr(), win.data_ptr()))
        return fb[:, :self.n_mels], pinv[:, :self.n_mels], win

    def workspace_bytes(self, batch: int) -> int:
        return int(self.lib.dn_workspace_bytes(self.handle, batch))


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require(t: torch.Tensor, dtype, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: expected a CUDA tensor (this package has no CPU path)")
    if t.dtype != dtype:
        raise TypeError(f"{what}: expected {dtype}, got {t.dtype}")


class _PlanModule(nn.Module):
    """Lazily builds one DspPlan per device the module is called on."""

    def __init__(self):
        super().__init__()
        self._plans = {}

    def _plan_args(self):  # -> dict for DspPlan
        raise NotImplementedError

    def plan(self, device: torch.device) -> DspPlan:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        p = self._plans.get(idx)
        if p is None:
            p = DspPlan(torch.device("cuda", idx), **self._plan_args())
            self._plans[idx] = p
        return p

    def __deepcopy__(self, memo):
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        import copy
        for k, v in self.__dict__.items():
            new.__dict__[k] = {} if k == "_plans" else copy.deepcopy(v, memo)
        return new


def _check_fft_args(n_fft, win_length, hop_length):
    win_length = n_fft if win_length is None else win_length
    hop_length = win_length // 2 if hop_length is None else hop_length
    if n_fft not in _N_FFTS or win_length != n_fft or hop_length != n_fft // 2:
        raise NotImplementedError(f"the FFT kernels are built for n_fft=win_length in {_N_FFTS} with hop_length=n_fft/2; "
                                  f"got n_fft={n_fft}, win_length={win_length}, hop_length={hop_length}")
    return n_fft, win_length, hop_length


class Spectrogram(_PlanModule):
    """Spectrogram(power=None|1|2, center=True, pad_mode='reflect', onesided=True, normalized=False):
    (.., n_fft) -> (.., n_fft/2+1, 3).  app3.py:135-139,191."""

    def __init__(self, n_fft=400, win_length=None, hop_length=None, pad=0, window_fn=torch.hann_window, power=2.0,
                 normalized=False, wkwargs=None, center=True, pad_mode="reflect", onesided=True):
        super().__init__()
        self.n_fft, self.win_length, self.hop_length = _check_fft_args(n_fft, win_length, hop_length)
        if pad != 0 or normalized or not center or pad_mode != "reflect" or not onesided:
            raise NotImplementedError("only pad=0, normalized=False, center=True, pad_mode='reflect', onesided=True")
        self.power = power
        window = window_fn(self.win_length) if wkwargs is None else window_fn(self.win_length, **wkwargs)
        self.register_buffer("window", window)

    def _plan_args(self):
        return dict(sample_rate=0, n_fft=self.n_fft, hop=self.hop_length, n_mels=0, window=self.window)

    def forward(self, waveform: torch.Tensor) -> torch.Tensor:
        _require(waveform, torch.float32, "Spectrogram")
        if waveform.shape[-1] != self.n_fft:
            raise NotImplementedError(f"frames of exactly n_fft={self.n_fft} samples (3 STFT columns) are supported; got {waveform.shape[-1]}")
        lead = waveform.shape[:-1]
        x = waveform.reshape(-1, self.n_fft).contiguous()
        B = x.shape[0]
        plan = self.plan(x.device)
        spec = torch.empty(B, 3, plan.n_stft, 2, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            plan.lib.check(plan.lib.dn_stft(plan.handle, x.data_ptr(), spec.data_ptr(), B, 0, _stream_ptr(x.device)))
        out = torch.view_as_complex(spec).reshape(lead + (3, plan.n_stft)).transpose(-1, -2)
        if self.power is None:
            return out
        return out.abs() if self.power == 1.0 else out.abs().pow(self.power)


def _rows_layout(t: torch.Tensor) -> torch.Tensor:
    """(.., R, T) logical -> contiguous [.., T, R] storage (free when t is already a transposed view)."""
    return t.transpose(-1, -2).contiguous()


class MelScale(_PlanModule):
    """MelScale(n_mels, sample_rate, f_min=0, f_max=sr//2, n_stft, norm=None, mel_scale='htk'):
    (.., n_stft, T) -> (.., n_mels, T).  app3.py:140-143,193."""

    def __init__(self, n_mels=128, sample_rate=16000, f_min=0.0, f_max=None, n_stft=201, norm=None, mel_scale="htk"):
        super().__init__()
        if norm is not None or mel_scale != "htk":
            raise NotImplementedError("only norm=None, mel_scale='htk' (what the reference uses)")
        if 2 * (n_stft - 1) not in _N_FFTS:
            raise NotImplementedError(f"n_stft must be one of {[n // 2 + 1 for n in _N_FFTS]}")
        self.n_mels, self.sample_rate, self.n_stft = n_mels, sample_rate, n_stft
        self.f_min = f_min
        self.f_max = f_max if f_max is not None else float(sample_rate // 2)
        self.register_buffer("fb", melscale_fbanks(n_stft, self.f_min, self.f_max, n_mels, sample_rate))

    def _plan_args(self):
        n_fft = 2 * (self.n_stft - 1)
        return dict(sample_rate=self.sample_rate, n_fft=n_fft, hop=n_fft // 2, n_mels=self.n_mels, fb=self.fb)

    def forward(self, specgram: torch.Tensor) -> torch.Tensor:
        _require(specgram, torch.float32, "MelScale")
        if specgram.shape[-2] != self.n_stft:
            raise RuntimeError(f"expected {self.n_stft} frequency bins, got {specgram.shape[-2]}")
        rows = _rows_layout(specgram)                      # [.., T, K]
        lead, T = rows.shape[:-2], rows.shape[-2]
        B = rows.numel() // (T * self.n_stft)
        plan = self.plan(rows.device)
        mel = torch.empty(lead + (T, self.n_mels), dtype=torch.float32, device=rows.device)
        with torch.cuda.device(rows.device):
            plan.lib.check(plan.lib.dn_mel_scale(plan.handle, rows.data_ptr(), mel.data_ptr(), B, T, _stream_ptr(rows.device)))
        return mel.transpose(-1, -2)


class InverseMelScale(_PlanModule):
    """InverseMelScale(n_stft, n_mels, sample_rate, f_min, f_max, norm=None, mel_scale='htk', driver='gels'):
    (.., n_mels, T) -> (.., n_stft, T) = relu(min-norm least squares).  app3.py:145-148,210."""

    def __init__(self, n_stft, n_mels=128, sample_rate=16000, f_min=0.0, f_max=None, norm=None, mel_scale="htk", driver="gels"):
        super().__init__()
        if norm is not None or mel_scale != "htk":
            raise NotImplementedError("only norm=None, mel_scale='htk' (what the reference uses)")
        if driver not in ("gels", "gelsy", "gelsd", "gelss"):
            raise ValueError(f'driver must be one of ["gels", "gelsy", "gelsd", "gelss"]. Found {driver}.')
        if 2 * (n_stft - 1) not in _N_FFTS:
            raise NotImplementedError(f"n_stft must be one of {[n // 2 + 1 for n in _N_FFTS]}")
        self.n_mels, self.sample_rate, self.n_stft, self.driver = n_mels, sample_rate, n_stft, driver
        self.f_min = f_min
        self.f_max = f_max if f_max is not None else float(sample_rate // 2)
        fb = melscale_fbanks(n_stft, self.f_min, self.f_max, n_mels, sample_rate)
        self.register_buffer("fb", fb)

    def _plan_args(self):
        n_fft = 2 * (self.n_stft - 1)
        return dict(sample_rate=self.sample_rate, n_fft=n_fft, hop=n_fft // 2, n_mels=self.n_mels, fb=self.fb)

    def forward(self, melspec: torch.Tensor) -> torch.Tensor:
        _require(melspec, torch.float32, "InverseMelScale")
        if melspec.shape[-2] != self.n_mels:
            raise ValueError(f"Expected an input with {self.n_mels} mel bins. Found: {melspec.shape[-2]}")
        rows = _rows_layout(melspec)                       # [.., T, M]
        lead, T = rows.shape[:-2], rows.shape[-2]
        B = rows.numel() // (T * self.n_mels)
        plan = self.plan(rows.device)
        lin = torch.empty(lead + (T, self.n_stft), dtype=torch.float32, device=rows.device)
        with torch.cuda.device(rows.device):
            plan.lib.check(plan.lib.dn_invmel(plan.handle, rows.data_ptr(), lin.data_ptr(), B, T, _stream_ptr(rows.device)))
        return lin.transpose(-1, -2)


class GriffinLim(_PlanModule):
    """GriffinLim(n_fft, n_iter=32, win_length, hop_length, window_fn, power=2.0, momentum=0.99, length=None,
    rand_init=True): (.., n_fft/2+1, 3) magnitude**power -> (.., n_fft).  app3.py:149-153,213.

    ``rand_init=True`` draws the initial phases on the device from a counter-based generator whose
    seed comes from torch's global RNG (so ``torch.manual_seed`` makes runs repeatable); pass
    ``init_angles`` (complex64, same shape as the input) to ``forward`` to inject them instead --
    that is how parity with the reference's ``torch.rand`` draw is tested."""

    def __init__(self, n_fft=400, n_iter=32, win_length=None, hop_length=None, window_fn=torch.hann_window, power=2.0,
                 wkwargs=None, momentum=0.99, length=None, rand_init=True):
        super().__init__()
        if not (0 <= momentum < 1):
            raise ValueError("momentum must be in the range [0, 1). Found: {}".format(momentum))
        self.n_fft, self.win_length, self.hop_length = _check_fft_args(n_fft, win_length, hop_length)
        if length is not None and length != n_fft:
            raise NotImplementedError("length=None (3 columns -> n_fft samples) is what the hop path uses")
        self.n_iter, self.power, self.momentum, self.rand_init = n_iter, power, momentum, rand_init
        window = window_fn(self.win_length) if wkwargs is None else window_fn(self.win_length, **wkwargs)
        self.register_buffer("window", window)

    def _plan_args(self):
        return dict(sample_rate=0, n_fft=self.n_fft, hop=self.hop_length, n_mels=0, window=self.window)

    def forward(self, specgram: torch.Tensor, init_angles: torch.Tensor | None = None) -> torch.Tensor:
        _require(specgram, torch.float32, "GriffinLim")
        K = self.n_fft // 2 + 1
        if specgram.shape[-2] != K or specgram.shape[-1] != 3:
            raise NotImplementedError(f"expected (.., {K}, 3) (one n_fft-sample frame = 3 columns); got {tuple(specgram.shape)}")
        if self.power != 1.0:
            specgram = specgram.pow(1.0 / self.power)
        rows = _rows_layout(specgram)                      # [.., 3, K]
        lead = rows.shape[:-2]
        B = rows.numel() // (3 * K)
        plan = self.plan(rows.device)
        ia_ptr, seed = None, 0
        if init_angles is not None:
            _require(init_angles, torch.complex64, "GriffinLim init_angles")
            if init_angles.shape != specgram.shape:
                raise ValueError("init_angles must have the shape of the spectrogram")
            ia = torch.view_as_real(_rows_layout(init_angles).resolve_conj()).contiguous()
            ia_ptr = ia.data_ptr()
        elif self.rand_init:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        else:   # angles = 1 + 1j?  torchaudio uses torch.full(.., 1, dtype=complex) = 1 + 0j
            ia = torch.zeros(rows.shape + (2,), dtype=torch.float32, device=rows.device)
            ia[..., 0] = 1.0
            ia_ptr = ia.data_ptr()
        wave = torch.empty(lead + (self.n_fft,), dtype=torch.float32, device=rows.device)
        with torch.cuda.device(rows.device):
            plan.lib.check(plan.lib.dn_griffinlim(plan.handle, rows.data_ptr(), ia_ptr, seed, 0, None, wave.data_ptr(), B,
                                                  int(self.n_iter), float(self.momentum), _stream_ptr(rows.device)))
        return wave


class InverseSpectrogram(_PlanModule):
    """InverseSpectrogram(n_fft, win_length, hop_length): complex (.., n_fft/2+1, 3) -> (.., n_fft).  server.py:174,216."""

    def __init__(self, n_fft=400, win_length=None, hop_length=None, pad=0, window_fn=torch.hann_window, normalized=False,
                 wkwargs=None, center=True, pad_mode="reflect", onesided=True):
        super().__init__()
        self.n_fft, self.win_length, self.hop_length = _check_fft_args(n_fft, win_length, hop_length)
        if pad != 0 or normalized or not center or not onesided:
            raise NotImplementedError("only pad=0, normalized=False, center=True, onesided=True")
        window = window_fn(self.win_length) if wkwargs is None else window_fn(self.win_length, **wkwargs)
        self.register_buffer("window", window)

    def _plan_args(self):
        return dict(sample_rate=0, n_fft=self.n_fft, hop=self.hop_length, n_mels=0, window=self.window)

    def forward(self, spectrogram: torch.Tensor, length=None) -> torch.Tensor:
        _require(spectrogram, torch.complex64, "InverseSpectrogram")
        K = self.n_fft // 2 + 1
        if spectrogram.shape[-2] != K or spectrogram.shape[-1] != 3 or (length is not None and length != self.n_fft):
            raise NotImplementedError(f"expected (.., {K}, 3) and length None; got {tuple(spectrogram.shape)}")
        rows = torch.view_as_real(_rows_layout(spectrogram).resolve_conj()).contiguous()   # [.., 3, K, 2]
        lead = rows.shape[:-3]
        B = rows.numel() // (3 * K * 2)
        plan = self.plan(rows.device)
        wave = torch.empty(lead + (self.n_fft,), dtype=torch.float32, device=rows.device)
        with torch.cuda.device(rows.device):
            plan.lib.check(plan.lib.dn_istft(plan.handle, rows.data_ptr(), wave.data_ptr(), B, _stream_ptr(rows.device)))
        return wave
