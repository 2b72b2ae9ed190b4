"""The per-hop pipeline of the reference's ``DenoisingAudioProcessor.recv`` loop body
(app3.py:178-226 == app2.py:185-233), batched over B independent streams and fused behind
``dn_process_frame`` / ``dn_stream_step``.

``Denoiser.process_frame`` is the new name for that loop body:
peak-normalise, Hann, 3-column STFT, mel, log1p, GRUUNet2 (3 steps), residual, expm1, inverse mel,
32-iteration Griffin-Lim, ``* peak`` -- P1..P12 of SURVEY.md section 8(a).  ``DenoiserStream`` adds the
per-stream state the reference keeps in ``input_buffer`` / ``output_ola_buffer`` / ``hx``
(app3.py:130-133), resident on the GPU.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib
from .gruunet2 import GRUUNet2
from .transforms import DspPlan, melscale_fbanks


class Denoiser:
    """Batched, fused hop pipeline bound to one model and one DSP plan on one GPU."""

    def __init__(self, model: GRUUNet2, sample_rate: int, n_fft: int = 1024, hop_length: int = 512, n_mels: int = 80,
                 n_iter: int = 32, momentum: float = 0.99, device=None):
        self.device = torch.device(device if device is not None else next(model.parameters()).device)
        if self.device.type != "cuda":
            raise RuntimeError("Denoiser needs a 'cuda' device; there is no CPU path in this package")
        if n_mels % 16 != 0:
            raise ValueError("n_mels must be a multiple of 16 (four stride-2 encoder levels)")
        self.lib = _lib.get_lib()
        self.model = model
        self.sample_rate, self.n_fft, self.hop, self.n_mels = sample_rate, n_fft, hop_length, n_mels
        self.n_stft = n_fft // 2 + 1
        self.n_iter, self.momentum = int(n_iter), float(momentum)
        self.num_compressed_bins = n_mels // 16
        # the same construction-time tensors the reference's transforms build (app3.py:135-155)
        fb = melscale_fbanks(self.n_stft, 0.0, float(sample_rate // 2), n_mels, sample_rate)
        self.plan = DspPlan(self.device, sample_rate, n_fft, hop_length, n_mels, fb=fb, window=torch.hann_window(n_fft))
        self._ws = None
        self._calls = 0

    def _flags(self) -> int:
        """DN_CONV_BF16 when the model asks for bf16 MFMA conv tiles (GRUUNet2.conv_precision, BASELINE config 3)."""
        if self.model.conv_precision not in ("fp32", "bf16"):
            raise ValueError("conv_precision must be 'fp32' or 'bf16'")
        return _lib.DN_CONV_BF16 if self.model.conv_precision == "bf16" else 0

    # -- helpers
    def _workspace(self, batch: int) -> torch.Tensor:
        need = self.plan.workspace_bytes(batch)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def init_hx(self, batch: int) -> torch.Tensor:
        """zeros(B, 17, C): app3.py:158-165."""
        return torch.zeros(batch, self.model.latent_size, self.num_compressed_bins, dtype=torch.float32, device=self.device)

    def _check(self, t, shape, name, dtype=torch.float32):
        if not t.is_cuda or t.device != self.device:
            raise RuntimeError(f"{name} must live on {self.device} (no CPU path)")
        if t.dtype != dtype or tuple(t.shape) != shape or not t.is_contiguous():
            raise ValueError(f"{name} must be contiguous {dtype} of shape {shape}; got {t.dtype} {tuple(t.shape)}")

    def _angles_ptr(self, init_angles, batch):
        """(B, K, 3) complex64 as the reference draws it -> [B][3][K] interleaved storage."""
        if init_angles is None:
            return None, None
        if tuple(init_angles.shape) != (batch, self.n_stft, 3) or init_angles.dtype != torch.complex64:
            raise ValueError(f"init_angles must be complex64 of shape {(batch, self.n_stft, 3)}")
        ia = torch.view_as_real(init_angles.to(self.device).transpose(-1, -2).contiguous())
        return ia, ia.data_ptr()

    def draw_phases(self, batch: int, seed: int, stream_id0: int = 0) -> torch.Tensor:
        """The initial Griffin-Lim phases the hop draws for (seed, stream_id0 + stream) when no ``init_angles`` are passed
        (dn_griffinlim_draw_phases): complex64 (B, n_fft/2+1, 3), real and imaginary part ~ U[0,1) as the reference's
        ``GriffinLim(rand_init=True)`` (app3.py:149-153).  Passing it back as ``init_angles`` reproduces the seeded call bit for bit."""
        buf = torch.empty(batch, 3, self.n_stft, 2, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            self.lib.check(self.lib.dn_griffinlim_draw_phases(self.plan.handle, seed, stream_id0, buf.data_ptr(), batch, st))
        return torch.view_as_complex(buf).transpose(-1, -2)

    # -- the hop body
    def process_frame(self, frames: torch.Tensor, hx: torch.Tensor | None = None, init_angles: torch.Tensor | None = None,
                      seed: int | None = None, stream_id0: int = 0, return_residual: bool = False):
        """frames (B, n_fft) raw samples, hx (B,17,C) or None -> (out (B, n_fft), hx_new[, predicted_diff_mel (B,3,M)]).

        ``hx`` is not modified (the reference rebinds ``self.hx`` to the returned tensor, app3.py:201)."""
        B = frames.shape[0]
        self._check(frames, (B, self.n_fft), "frames")
        hx_new = self.init_hx(B) if hx is None else hx.clone()
        self._check(hx_new, (B, self.model.latent_size, self.num_compressed_bins), "hx")
        out = torch.empty_like(frames)
        resid = torch.empty(B, 3, self.n_mels, dtype=torch.float32, device=self.device) if return_residual else None
        keep, ia_ptr = self._angles_ptr(init_angles, B)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if init_angles is None else 0
        ws = self._workspace(B)
        model_h = self.model._native(self.device)
        with torch.cuda.device(self.device):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            self.lib.check(self.lib.dn_process_frame(model_h, self.plan.handle, frames.data_ptr(), hx_new.data_ptr(), out.data_ptr(),
                                                     None if resid is None else resid.data_ptr(), ia_ptr, seed, stream_id0,
                                                     self.n_iter, self.momentum, ws.data_ptr(), B, self._flags(), st))
        return (out, hx_new, resid) if return_residual else (out, hx_new)

    def process_frame_(self, frames: torch.Tensor, hx: torch.Tensor, out: torch.Tensor, seed: int = 0, stream_id0: int = 0) -> None:
        """Allocation-free variant for steady-state loops and hipGraph capture: ``hx`` is advanced in place and
        the denoised frames are written to ``out``; initial phases come from the device generator."""
        B = frames.shape[0]
        ws = self._workspace(B)
        model_h = self.model._native(self.device)
        with torch.cuda.device(self.device):          # the launch goes to the CURRENT device: make it the denoiser's
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            self.lib.check(self.lib.dn_process_frame(model_h, self.plan.handle, frames.data_ptr(), hx.data_ptr(), out.data_ptr(), None, None,
                                                     seed, stream_id0, self.n_iter, self.momentum, ws.data_ptr(), B, self._flags(), st))


class ServerDenoiser:
    """The request-loop body of the reference's socket server (server.py:199-217) for B mono chunks at once:
    STFT -> mel -> log1p -> GRUUNet2 over all T columns -> relu(out)*3, hx*=0.9 -> inverse mel of exp(log_mel-out)-1
    -> torch.polar with the NOISY phase -> ISTFT.  Chunks may have any length L > n_fft/2; the result has
    hop*(L//hop) samples (InverseSpectrogram with length=None).  Four launches; no CPU fallback."""

    def __init__(self, model: GRUUNet2, sample_rate: int = 48000, n_fft: int = 1024, hop_length: int = 512, n_mels: int = 64,
                 hx_decay: float = 0.9, device=None):
        self.device = torch.device(device if device is not None else next(model.parameters()).device)
        if self.device.type != "cuda":
            raise RuntimeError("ServerDenoiser needs a 'cuda' device; there is no CPU path in this package")
        self.lib = _lib.get_lib()
        self.model, self.hx_decay = model, float(hx_decay)
        self.sample_rate, self.n_fft, self.hop, self.n_mels = sample_rate, n_fft, hop_length, n_mels
        self.n_stft = n_fft // 2 + 1
        fb = melscale_fbanks(self.n_stft, 0.0, float(sample_rate // 2), n_mels, sample_rate)        # server.py:175-176
        self.plan = DspPlan(self.device, sample_rate, n_fft, hop_length, n_mels, fb=fb, window=torch.hann_window(n_fft))

    def process(self, x: torch.Tensor, hx: torch.Tensor | None = None):
        """x (B, L) float32 on the GPU, hx (B,17,C) or None -> (wave (B, hop*(L//hop)), hx_new)."""
        if not x.is_cuda or x.device != self.device or x.dtype != torch.float32 or x.dim() != 2:
            raise ValueError(f"x must be a float32 (B, L) tensor on {self.device}")
        B, L = x.shape
        T = 1 + L // self.hop
        Cb = self.n_mels // 16
        dev, lib, plan = self.device, self.lib, self.plan
        x = x.contiguous()
        spec = torch.empty(B, T, self.n_stft, 2, dtype=torch.float32, device=dev)
        logmel = torch.empty(B, T, self.n_mels, dtype=torch.float32, device=dev)
        out = torch.empty_like(logmel)
        hx0 = torch.zeros(B, self.model.latent_size, Cb, dtype=torch.float32, device=dev) if hx is None else hx.contiguous()
        hx1 = torch.empty_like(hx0)
        wave = torch.empty(B, self.hop * (T - 1), dtype=torch.float32, device=dev)
        model_h = self.model._native(dev)
        with torch.cuda.device(dev):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            lib.check(lib.dn_stft_general(plan.handle, x.data_ptr(), spec.data_ptr(), logmel.data_ptr(), B, L, st))
            lib.check(lib.dn_cell_forward_ex(model_h, logmel.data_ptr(), hx0.data_ptr(), out.data_ptr(), hx1.data_ptr(), B, T, self.n_mels, Cb,
                                             self.hx_decay, st))
            lib.check(lib.dn_server_rows(plan.handle, logmel.data_ptr(), out.data_ptr(), spec.data_ptr(), spec.data_ptr(), B * T, st))
            lib.check(lib.dn_istft_general(plan.handle, spec.data_ptr(), wave.data_ptr(), B, T, st))
        return wave, hx1


class _Pipe:
    """Common part of the two front ends of ``dn_pipe``: owns the native pipe, keeps it bound to the model's CURRENT
    weights (the pipe itself holds a reference on the dn_model it launches with, so reloading weights between two hops
    can never leave it with freed device memory: ``_bind`` rebinds it to the new handle before the next launch)."""

    def __init__(self, denoiser: "Denoiser", batch: int, streaming: bool):
        import weakref
        self.dn, self.batch = denoiser, batch
        self.lib = denoiser.lib
        handle = C.c_void_p()
        self._owner = denoiser.model._native_owner(denoiser.device)      # keeps the dn_model alive on the Python side too
        self._flags = denoiser._flags()
        create = self.lib.dn_pipe_stream_create if streaming else self.lib.dn_pipe_create
        with torch.cuda.device(denoiser.device):
            self.lib.check(create(self._owner.handle, denoiser.plan.handle, batch, self._flags, C.byref(handle)))
        self.handle = handle
        self._fin = weakref.finalize(self, self.lib.dn_pipe_destroy, handle)
        hs = os.environ.get("DN_GL_HEAD_START")       # experiment knob of tools/head_start_sweep.sh (the library itself reads no environment)
        if hs is not None:
            self.lib.check(self.lib.dn_pipe_set_head_start(handle, int(hs)))
        self.depth = 1
        self.group = 0
        dp = os.environ.get("DN_PIPE_DEPTH")           # likewise: hops of one stream in flight
        if dp is not None:
            self.set_depth(int(dp))
        sch = os.environ.get("DN_GL_SCHEDULE")         # likewise: 0 auto, 1 a wavefront per column, 2 a wavefront per stream
        if sch is not None:
            self.lib.check(self.lib.dn_pipe_set_gl_schedule(handle, int(sch)))

        sp = os.environ.get("DN_PIPE_SPLIT")           # likewise: -1 auto, 0 one launch per hop, 1 two (chains, then front halves)
        if sp is not None:
            self.set_split(int(sp))

    def set_split(self, mode: int) -> None:
        """``_lib.DN_SPLIT_AUTO`` / ``DN_SPLIT_OFF`` / ``DN_SPLIT_ON`` (dn_pipe_set_split): a hop as two launches, the chains of the hops in
        flight and then the new hop's front halves under their own register budget; same samples."""
        self.lib.check(self.lib.dn_pipe_set_split(self.handle, int(mode)))

    def set_gl_schedule(self, schedule: int) -> None:
        """``_lib.DN_GL_AUTO`` / ``DN_GL_WAVE_PER_COLUMN`` / ``DN_GL_WAVE_PER_STREAM`` (dn_pipe_set_gl_schedule): how the pending hop's
        Griffin-Lim is laid out on the GPU; results are bit-identical, call between hops."""
        self.lib.check(self.lib.dn_pipe_set_gl_schedule(self.handle, int(schedule)))

    def set_depth(self, depth: int) -> None:
        """dn_pipe_set_depth: hops of ONE stream in flight (1 .. 4, n_fft 1024).  With depth D a frame's Griffin-Lim chain runs as D segments
        in the D launches after its submit (bit-identical results): about one stream per CU then reaches the throughput of the saturated
        regime, for D - 1 more hops of latency.  Call while nothing is in flight."""
        with torch.cuda.device(self.dn.device):
            self.lib.check(self.lib.dn_pipe_set_depth(self.handle, int(depth)))
        self.depth = int(depth)

    def set_group(self, hops: int) -> None:
        """dn_pipe_set_group: a launch carries ``hops`` (1 .. 4; 0 = back to single hops) CONSECUTIVE hops of every stream -- their front halves in
        order, hx handed on -- beside the WHOLE Griffin-Lim chains of the hops the previous launch fronted (one wavefront each): no chain is ever
        parked between launches.  Same frames, hx and samples as the one-hop pipe, bit for bit; n_fft 1024.  Call while nothing is in flight."""
        with torch.cuda.device(self.dn.device):
            self.lib.check(self.lib.dn_pipe_set_group(self.handle, int(hops)))
        self.group = int(hops)

    def set_head_start(self, iterations: int) -> None:
        """dn_pipe_set_head_start: Griffin-Lim iterations a front workgroup runs of its own frame's chain (0 = off)."""
        self.lib.check(self.lib.dn_pipe_set_head_start(self.handle, int(iterations)))

    def _bind(self) -> None:
        """Follow the model: weights reloaded / moved / updated since the last hop -> rebind the pipe (one C call)."""
        owner = self.dn.model._native_owner(self.dn.device)
        if owner is not self._owner:
            self.lib.check(self.lib.dn_pipe_set_model(self.handle, owner.handle))
            self._owner = owner
        if self.dn._flags() != self._flags:
            raise RuntimeError("conv_precision changed after the pipeline was created; create a new pipeline")

    def counters(self):
        """(pushes, frames, pending) of the device-resident control block.  Synchronises the current stream."""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_int32()
        with torch.cuda.device(self.dn.device):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            self.lib.check(self.lib.dn_pipe_get_counters(self.handle, C.byref(a), C.byref(b), C.byref(c), st))
        return a.value, b.value, bool(c.value)


class HopPipeline(_Pipe):
    """Software-pipelined hops (``dn_pipe_*``): one launch per hop carries hop n's Griffin-Lim workgroups next to
    hop n+1's analysis + GRUUNet2 + inverse-mel workgroups; ``hx`` is the only dependency between hops.
    ``submit`` enqueues one hop for the whole batch on the current stream; the output of a hop is complete (in
    stream order) after the next ``submit`` (the ``depth``-th next one of a deeper pipe, ``set_depth``) or ``flush``.
    ``frames``/``out``/``hx`` must not be touched until then.
    The Griffin-Lim of the f-th submitted hop draws from ``seed + f``.  A ``submit`` captured in a hipGraph can be
    replayed: slot parity, pending flag and frame index live on the device."""

    def __init__(self, denoiser: "Denoiser", batch: int):
        super().__init__(denoiser, batch, streaming=False)

    def submit(self, frames: torch.Tensor, hx: torch.Tensor, out: torch.Tensor, seed: int = 0, stream_id0: int = 0,
               init_angles: torch.Tensor | None = None, check_weights: bool = True) -> None:
        d = self.dn
        if check_weights:
            self._bind()
        keep, ia_ptr = d._angles_ptr(init_angles, self.batch)
        with torch.cuda.device(d.device):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            self.lib.check(self.lib.dn_pipe_submit(self.handle, frames.data_ptr(), hx.data_ptr(), out.data_ptr(), ia_ptr, seed, stream_id0,
                                                   d.n_iter, d.momentum, st))
            if keep is not None and not torch.cuda.is_current_stream_capturing():
                keep.record_stream(torch.cuda.current_stream())     # the launch copies the phases into its scratch slot

    def submit_group(self, frames: torch.Tensor, hx: torch.Tensor, out: torch.Tensor, seed: int = 0, stream_id0: int = 0,
                     init_angles: torch.Tensor | None = None, check_weights: bool = True) -> None:
        """``frames`` / ``out`` of shape ``(H, B, n_fft)``, H <= the pipe's group size (``set_group``): H consecutive hops of every stream
        in ONE launch (dn_pipe_submit_group).  ``hx`` is advanced through all H; ``out`` is complete after the next ``submit_group`` or
        ``flush``.  ``init_angles`` (parity mode): ``(H, B, K, 3)`` complex64.  The f-th frame of the pipe draws from ``seed + f``."""
        d = self.dn
        if frames.dim() != 3 or tuple(out.shape) != tuple(frames.shape) or frames.shape[1:] != (self.batch, d.n_fft):
            raise ValueError(f"frames and out must both be (H, {self.batch}, {d.n_fft})")
        for t in (frames, out):
            if t.dtype != torch.float32 or t.device != d.device or t.stride(2) != 1 or t.stride(1) != d.n_fft:
                raise ValueError("frames and out must be float32 on the denoiser's device with contiguous (B, n_fft) hops")
        H = frames.shape[0]
        if check_weights:
            self._bind()
        ia_ptr, ia_stride, keep = None, 0, []
        if init_angles is not None:
            if init_angles.shape[0] != H:
                raise ValueError("init_angles: one set of phases per hop of the group")
            keep = [d._angles_ptr(init_angles[h], self.batch)[0] for h in range(H)]
            packed = torch.stack(keep)                    # (H, B, 3, K, 2) float32, contiguous
            keep = [packed]
            ia_ptr, ia_stride = packed.data_ptr(), packed.stride(0)
        with torch.cuda.device(d.device):
            cur = torch.cuda.current_stream()
            self.lib.check(self.lib.dn_pipe_submit_group(self.handle, frames.data_ptr(), frames.stride(0) if H > 1 else 0, hx.data_ptr(), out.data_ptr(),
                                                         out.stride(0) if H > 1 else 0, ia_ptr, ia_stride, seed, stream_id0, H, d.n_iter, d.momentum,
                                                         C.c_void_p(cur.cuda_stream)))
            for k in keep:
                if not torch.cuda.is_current_stream_capturing():
                    k.record_stream(cur)

    def flush(self) -> None:
        with torch.cuda.device(self.dn.device):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            self.lib.check(self.lib.dn_pipe_flush(self.handle, self.dn.n_iter, self.dn.momentum, st))


class PipelinedStream(_Pipe):
    """B concurrent streams, state (ring, overlap-add line, hx) owned by the native pipe, ONE launch per hop
    (``dn_pipe_stream_*``; BASELINE config 5).  ``push(hop)`` takes ``(B, hop_length)`` new samples (float32, or int16
    PCM) and returns ``(B, hop_length)`` output samples in the same format.  Hops are software-pipelined, so the
    samples the reference emits while processing frame f come out one push later (zeros until then);
    ``flush()`` returns the last pending hop.  Frame f's Griffin-Lim draws from ``seed + f`` (as ``DenoiserStream``).
    ``push_`` writes into a caller-provided tensor and allocates nothing: captured in a hipGraph it is the replayable
    steady-state step (``graph_step``)."""

    def __init__(self, denoiser: "Denoiser", batch: int, stream_id0: int = 0, seed: int = 0):
        super().__init__(denoiser, batch, streaming=True)
        self.stream_id0, self.seed = stream_id0, seed

    def _out(self, like_s16: bool) -> torch.Tensor:
        return torch.empty(self.batch, self.dn.hop, dtype=torch.int16 if like_s16 else torch.float32, device=self.dn.device)

    def push_(self, hop: torch.Tensor, out: torch.Tensor, init_angles: torch.Tensor | None = None, check_weights: bool = True) -> None:
        """Allocation-free push: ``out`` (same shape/dtype family as ``hop``) receives the emitted samples."""
        d = self.dn
        for t, name in ((hop, "hop"), (out, "out")):
            if t.device != d.device or tuple(t.shape) != (self.batch, d.hop) or t.dtype not in (torch.float32, torch.int16) \
                    or not t.is_contiguous():
                raise ValueError(f"{name} must be contiguous float32 or int16 of shape {(self.batch, d.hop)} on {d.device}")
        if check_weights:
            self._bind()
        keep, ia_ptr = d._angles_ptr(init_angles, self.batch)
        with torch.cuda.device(d.device):
            cur = torch.cuda.current_stream()
            self.lib.check(self.lib.dn_pipe_stream_push(self.handle, hop.data_ptr(), int(hop.dtype == torch.int16), out.data_ptr(),
                                                        int(out.dtype == torch.int16), ia_ptr, self.seed, self.stream_id0, d.n_iter,
                                                        d.momentum, C.c_void_p(cur.cuda_stream)))
            if keep is not None and not torch.cuda.is_current_stream_capturing():
                keep.record_stream(cur)

    def push(self, hop: torch.Tensor, init_angles: torch.Tensor | None = None) -> torch.Tensor:
        out = self._out(hop.dtype == torch.int16)
        self.push_(hop, out, init_angles)
        hop.record_stream(torch.cuda.current_stream(self.dn.device))
        return out

    def push_group_(self, hops: torch.Tensor, out: torch.Tensor, init_angles: torch.Tensor | None = None, check_weights: bool = True) -> None:
        """Allocation-free push of a GROUP (``set_group(H)``): ``hops`` and ``out`` of shape ``(H, B, hop_length)`` (float32 or int16 PCM) -- H
        consecutive hops of every stream in ONE launch (dn_pipe_stream_push_group), H hops emitted: the stream a one-hop pipe emits,
        H - 1 hops later (zeros until then).  Capturable like ``push_``."""
        d = self.dn
        H = self.group
        for t, name in ((hops, "hops"), (out, "out")):
            if t.device != d.device or tuple(t.shape) != (H, self.batch, d.hop) or t.dtype not in (torch.float32, torch.int16) or t.stride(2) != 1 \
                    or t.stride(1) != d.hop:
                raise ValueError(f"{name} must be float32 or int16 of shape {(H, self.batch, d.hop)} on {d.device} with contiguous (B, hop) rows")
        if check_weights:
            self._bind()
        ia_ptr, ia_stride, keep = None, 0, None
        if init_angles is not None:
            keep = torch.stack([d._angles_ptr(init_angles[h], self.batch)[0] for h in range(H)])
            ia_ptr, ia_stride = keep.data_ptr(), keep.stride(0)
        with torch.cuda.device(d.device):
            cur = torch.cuda.current_stream()
            self.lib.check(self.lib.dn_pipe_stream_push_group(self.handle, hops.data_ptr(), hops.stride(0), int(hops.dtype == torch.int16), out.data_ptr(),
                                                              out.stride(0), int(out.dtype == torch.int16), ia_ptr, ia_stride, self.seed,
                                                              self.stream_id0, d.n_iter, d.momentum, C.c_void_p(cur.cuda_stream)))
            if keep is not None and not torch.cuda.is_current_stream_capturing():
                keep.record_stream(cur)

    def push_group(self, hops: torch.Tensor, init_angles: torch.Tensor | None = None) -> torch.Tensor:
        out = torch.empty_like(hops)
        self.push_group_(hops, out, init_angles)
        hops.record_stream(torch.cuda.current_stream(self.dn.device))
        return out

    def flush_group(self, s16: bool = False):
        """Drains a group pipe: ``(out (H, B, hop_length), valid)`` -- the pending frames' hops first, zero hops behind them."""
        d = self.dn
        out = torch.empty(self.group, self.batch, d.hop, dtype=torch.int16 if s16 else torch.float32, device=d.device)
        valid = C.c_int32()
        with torch.cuda.device(d.device):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            self.lib.check(self.lib.dn_pipe_stream_flush_group(self.handle, out.data_ptr(), out.stride(0), int(s16), C.byref(valid), st))
        return out, valid.value

    def graph_step(self, hop: torch.Tensor, out: torch.Tensor, init_angles: torch.Tensor | None = None) -> "torch.cuda.CUDAGraph":
        """Capture ONE push into a hipGraph (BASELINE config 5: the hipGraph-captured step).  Replaying it is a push of
        whatever ``hop`` (and ``init_angles``, parity mode) hold at that moment into ``out``; it can be replayed
        indefinitely and mixed with eager pushes.  ``hop`` / ``out`` of shape ``(K, B, hop_length)`` capture K consecutive pushes
        (hop k into out k) as ONE graph: a replay then costs one graph launch per K hops (a graph launch leaves ~5 us of idle
        GPU behind it where eager launches run back to back)."""
        self._bind()
        if init_angles is not None:
            # the slot buffers for injected phases are allocated by the first parity-mode launch: do that outside the capture
            self._warm_parity()
        if isinstance(hop, (list, tuple)):             # K separate (B, hop) tensors instead of one (K, B, hop): the same K pushes
            if not isinstance(out, (list, tuple)) or len(out) != len(hop) or (init_angles is not None and len(init_angles) != len(hop)):
                raise ValueError("K hops need K outputs (and K sets of init_angles)")
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for k in range(len(hop)):
                    self.push_(hop[k], out[k], None if init_angles is None else init_angles[k], check_weights=False)
            return g
        if hop.dim() == 3 and (out.dim() != 3 or out.shape[0] != hop.shape[0] or (init_angles is not None and init_angles.shape[0] != hop.shape[0])):
            raise ValueError("K hops need K outputs (and K sets of init_angles)")
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            if hop.dim() == 3:
                for k in range(hop.shape[0]):
                    self.push_(hop[k], out[k], None if init_angles is None else init_angles[k], check_weights=False)
            else:
                self.push_(hop, out, init_angles, check_weights=False)
        return g

    def _warm_parity(self) -> None:
        with torch.cuda.device(self.dn.device):
            self.lib.check(self.lib.dn_pipe_reserve_parity(self.handle))

    def flush(self, s16: bool = False) -> torch.Tensor:
        """Drains the pipe: the hops still in flight, ``(B, depth * hop_length)`` (zeros where nothing was pending)."""
        outs = []
        with torch.cuda.device(self.dn.device):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            for _ in range(self.depth):
                out = self._out(s16)
                self.lib.check(self.lib.dn_pipe_stream_flush(self.handle, out.data_ptr(), int(s16), self.dn.n_iter, self.dn.momentum, st))
                outs.append(out)
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)

    def state(self):
        """Snapshot (ring, ola, hx, frames) of the pipe-owned stream state (checkpointing live streams; call after
        ``flush()``).  ``frames`` is the number of frames processed so far: it keys the Griffin-Lim seed sequence."""
        d = self.dn
        ring = torch.empty(self.batch, d.n_fft, dtype=torch.float32, device=d.device)
        ola = torch.empty_like(ring)
        hx = torch.empty(self.batch, d.model.latent_size, d.num_compressed_bins, dtype=torch.float32, device=d.device)
        with torch.cuda.device(d.device):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            self.lib.check(self.lib.dn_pipe_stream_get_state(self.handle, ring.data_ptr(), ola.data_ptr(), hx.data_ptr(), st))
        return ring, ola, hx, self.counters()[1]

    def load_state(self, ring: torch.Tensor, ola: torch.Tensor, hx: torch.Tensor, frames: int = 0) -> None:
        """Resume streams from a snapshot taken with ``state()``: the next frame is frame ``frames`` of the seed sequence."""
        for t in (ring, ola, hx):
            if t.device != self.dn.device or t.dtype != torch.float32 or not t.is_contiguous():
                raise ValueError("state tensors must be contiguous float32 on the denoiser's device")
        with torch.cuda.device(self.dn.device):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            self.lib.check(self.lib.dn_pipe_stream_set_state(self.handle, ring.data_ptr(), ola.data_ptr(), hx.data_ptr(), int(frames), st))


class HostFedStream(PipelinedStream):
    """``PipelinedStream`` fed from HOST buffers, as the reference feeds its hop (int16 frames on the host, ``.to(device)`` / ``.cpu()`` around
    the model: app3.py:168-172,189,215,244-250) -- with the transfers overlapped: ``dn_pipe_stream_push_host`` uploads hop i+1 and downloads
    the result of hop i-1 on two copy queues while hop i computes (device staging double-buffered; page-locked rings of four buffers here).
    ``push(hop)`` takes a CPU tensor ``(B, hop_length)`` (int16 or float32) and returns the samples emitted ``LAG`` pushes earlier (zeros
    until then): the host stays two launches ahead of the GPU, so it never waits for a hop that is still computing and the GPU never waits
    for the host.  ``drain()`` returns what is still on its way.  Same samples as the device-fed stream, bit for bit.
    ``defer`` (default, zero copy only): a hop's samples leave the device during the NEXT launch (``DN_HOST_DEFER``), spread over it instead
    of ending their own launch as one PCIe burst -- ``LAG`` is 3 then, 2 otherwise."""

    RING = 4

    def __init__(self, denoiser: "Denoiser", batch: int, stream_id0: int = 0, seed: int = 0, s16: bool = True, depth: int = 1,
                 staged: bool = False, defer: bool = True):
        super().__init__(denoiser, batch, stream_id0, seed)
        if depth != 1:
            self.set_depth(depth)
        # default: zero copy (the launch reads the page-locked input itself; its output follows one launch later, or directly with defer=False)
        self._hflags = _lib.DN_HOST_STAGED if staged else (_lib.DN_HOST_DEFER if defer else 0)
        self.LAG = 3 if (defer and not staged) else 2
        dt = torch.int16 if s16 else torch.float32
        self._pin_in = [torch.zeros(batch, denoiser.hop, dtype=dt).pin_memory() for _ in range(self.RING)]
        self._pin_out = [torch.zeros(batch, denoiser.hop, dtype=dt).pin_memory() for _ in range(self.RING)]
        self._in_ptr = [t.data_ptr() for t in self._pin_in]
        self._out_ptr = [t.data_ptr() for t in self._pin_out]
        self._nbytes = self._pin_in[0].numel() * self._pin_in[0].element_size()
        self._dtype = dt
        self._s16 = int(s16)
        self._n = 0                # pushes made
        self._taken = 0            # results handed out
        self._ticket = C.c_uint64()

    def push(self, hop: torch.Tensor, copy: bool = True) -> torch.Tensor:
        d = self.dn
        if hop.dtype != self._dtype or hop.device.type != "cpu" or tuple(hop.shape) != (self.batch, d.hop) or not hop.is_contiguous():
            raise ValueError(f"hop must be a contiguous CPU tensor {self._dtype} of shape {(self.batch, d.hop)}")
        k = self._n % self.RING
        # ring slot k was last used by push n - RING: its upload finished before the result of push n - RING was handed out, LAG pushes ago
        C.memmove(self._in_ptr[k], hop.data_ptr(), self._nbytes)
        with torch.cuda.device(d.device):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            self.lib.check(self.lib.dn_pipe_stream_push_host(self.handle, self._in_ptr[k], self._s16, self._out_ptr[k], self._s16,
                                                             self.seed, self.stream_id0, d.n_iter, d.momentum, self._hflags, st, C.byref(self._ticket)))
        self._n += 1
        if self._n <= self.LAG:
            return torch.zeros_like(self._pin_out[0])
        return self._result(self._n - 1 - self.LAG, copy)

    def _result(self, i: int, copy: bool) -> torch.Tensor:
        self.lib.check(self.lib.dn_pipe_stream_host_wait(self.handle, i))
        self._taken = i + 1
        out = self._pin_out[i % self.RING]
        return out.clone() if copy else out

    def drain(self) -> torch.Tensor:
        """The results of the pushes not handed out yet, then the hops still in flight in the pipe (``flush``), on the host."""
        parts = [self._result(i, True) for i in range(self._taken, self._n)]
        tail = self.flush(s16=bool(self._s16)).cpu()
        return torch.cat(parts + [tail], dim=1)


class _Queues:
    """B independent streams as Q pipes of B/Q streams, each on a HIP stream (hardware queue) of its own, every hop split in two launches
    (``dn_pipe_set_split``: the chains of the hops in flight, then the new hop's front halves as a kernel of their own).  Streams do not
    depend on one another, so neither do the pipes: no event ever crosses from one queue to another, and while one pipe's chains (243
    registers a wavefront) occupy a SIMD the other pipe's front halves (101-112 registers) run beside them -- three wavefronts per SIMD where
    one launch allows two.  Measured on one MI355X: 1,024 streams 153.9 -> 138.9 us per hop (two queues, depth 2), 2,048 streams
    292 -> 262 us (two queues, depth 1); below 1,024 streams a single pipe is faster.  Same samples as one pipe of B streams bit for bit
    (the generator is keyed by the global stream id).
    The queues are this object's own: ``after(stream)`` makes them wait for work already enqueued on ``stream`` (inputs produced there),
    ``before(stream)`` makes ``stream`` wait for everything enqueued on them so far (outputs consumed there), ``synchronize()`` blocks the
    host.  Without those calls the caller's tensors must simply be ready before ``submit`` / ``push_`` and untouched until
    ``synchronize`` -- the usual contract of a stream.
    One event does cross, once: queues that start a run together stay in lockstep -- both in their chains, then both in their front halves, for as
    long as nothing synchronises them -- and that state measured 12 % slower than the staggered one at 8,192 streams (1.10 against 0.976 ms per hop after
    a flush; a run that starts from hops in flight drifts apart by itself).  The SECOND hop after creation or a flush -- the first one whose launches
    carry chains -- therefore makes queue q wait for the first pipe of queue q - 1: an offset of one pipe's whole hop, paid once."""

    def _setup(self, denoiser: "Denoiser", batch: int, queues: int, depth: int, split: bool, make, pipes: int | None = None):
        n_pipes = queues if pipes is None else pipes          # more pipes than queues: pipe i runs on queue i % queues, one after the other
        if queues < 1 or n_pipes < queues or batch < n_pipes:
            raise ValueError("1 <= queues <= pipes <= number of streams")
        self.dn, self.batch, self.depth = denoiser, batch, depth
        self.bounds = [(batch * q) // n_pipes for q in range(n_pipes + 1)]        # contiguous blocks of streams, sizes differing by at most one
        with torch.cuda.device(denoiser.device):
            self.streams = [torch.cuda.Stream(device=denoiser.device) for _ in range(queues)]
        self.pipes = []
        self._stagger = 2                  # hops until the queues are set one pipe's hop apart (see above): the second hop of a run
        for q in range(n_pipes):
            pipe = make(self.bounds[q + 1] - self.bounds[q], self.bounds[q])
            if depth != 1:
                pipe.set_depth(depth)
            elif denoiser.n_fft == 1024:
                pipe.set_gl_schedule(_lib.DN_GL_WAVE_PER_STREAM)                 # (what a split hop needs)
            if split and denoiser.n_fft == 1024:
                pipe.set_head_start(0)          # (a front workgroup that is a launch of its own cannot go on with its frame's chain: the library refuses the combination)
                pipe.set_split(_lib.DN_SPLIT_ON)
            else:
                pipe.set_split(_lib.DN_SPLIT_OFF)
            self.pipes.append(pipe)

    def _each(self):
        for q, pipe in enumerate(self.pipes):
            yield pipe, self.streams[q % len(self.streams)], self.bounds[q], self.bounds[q + 1]

    def _each_staggered(self):
        """``_each`` for a hop: on the second hop of a run queue q starts behind the first pipe of queue q - 1."""
        nq = len(self.streams)
        stagger = self._stagger == 1 and nq > 1 and not torch.cuda.is_current_stream_capturing()
        self._stagger = max(0, self._stagger - 1)
        ev = None
        for q, item in enumerate(self._each()):
            st = item[1]
            if stagger and 0 < q < nq:
                st.wait_event(ev)
            yield item
            if stagger and q < nq - 1:
                ev = st.record_event()

    def after(self, stream: "torch.cuda.Stream | None" = None) -> None:
        ev = (stream or torch.cuda.current_stream(self.dn.device)).record_event()
        for st in self.streams:
            st.wait_event(ev)

    def before(self, stream: "torch.cuda.Stream | None" = None) -> None:
        tgt = stream or torch.cuda.current_stream(self.dn.device)
        for st in self.streams:
            tgt.wait_event(st.record_event())

    def synchronize(self) -> None:
        for st in self.streams:
            st.synchronize()


class QueuedHopPipelines(_Queues):
    """``HopPipeline`` for B streams as ``queues`` pipes on as many HIP streams (see ``_Queues``).  ``submit`` / ``flush`` as ``HopPipeline``;
    row block q of ``frames`` / ``hx`` / ``out`` belongs to pipe q."""

    def __init__(self, denoiser: "Denoiser", batch: int, queues: int = 2, depth: int = 2, split: bool = True, pipes: int | None = None):
        self._setup(denoiser, batch, queues, depth, split, lambda n, lo: HopPipeline(denoiser, n), pipes)

    def submit(self, frames: torch.Tensor, hx: torch.Tensor, out: torch.Tensor, seed: int = 0, stream_id0: int = 0,
               init_angles: torch.Tensor | None = None, check_weights: bool = True) -> None:
        for pipe, st, lo, hi in self._each_staggered():
            with torch.cuda.stream(st):
                pipe.submit(frames[lo:hi], hx[lo:hi], out[lo:hi], seed=seed, stream_id0=stream_id0 + lo,
                            init_angles=None if init_angles is None else init_angles[lo:hi], check_weights=check_weights)

    def flush(self) -> None:
        for pipe, st, lo, hi in self._each():
            with torch.cuda.stream(st):
                pipe.flush()
        self._stagger = 2


class QueuedPipelinedStreams(_Queues):
    """``PipelinedStream`` for B streams as ``queues`` pipes on as many HIP streams (see ``_Queues``): ``push_`` / ``flush`` / ``graph_steps``."""

    def __init__(self, denoiser: "Denoiser", batch: int, queues: int = 2, depth: int = 2, split: bool = True, stream_id0: int = 0, seed: int = 0,
                 pipes: int | None = None):
        self._setup(denoiser, batch, queues, depth, split, lambda n, lo: PipelinedStream(denoiser, n, stream_id0=stream_id0 + lo, seed=seed), pipes)

    def push_(self, hop: torch.Tensor, out: torch.Tensor, check_weights: bool = True) -> None:
        for pipe, st, lo, hi in self._each_staggered():
            with torch.cuda.stream(st):
                pipe.push_(hop[lo:hi], out[lo:hi], check_weights=check_weights)

    def flush(self, s16: bool = False) -> torch.Tensor:
        parts = []
        for pipe, st, lo, hi in self._each():
            with torch.cuda.stream(st):
                parts.append(pipe.flush(s16=s16))
        self.synchronize()
        self._stagger = 2
        return torch.cat(parts, dim=0)

    def graph_steps(self, hop: torch.Tensor, out: torch.Tensor):
        """One captured graph per pipe (``PipelinedStream.graph_step`` of its row block of the (B, hop_length) tensors -- or of K such tensors
        each, given as lists: K pushes per graph); returns ``replay()``, which launches each graph on its queue."""
        graphs = []
        many = isinstance(hop, (list, tuple))
        for pipe, st, lo, hi in self._each():
            with torch.cuda.stream(st):
                graphs.append((pipe.graph_step([h[lo:hi] for h in hop], [o[lo:hi] for o in out]) if many
                               else pipe.graph_step(hop[lo:hi], out[lo:hi]), st))

        def replay():
            for g, st in graphs:
                with torch.cuda.stream(st):
                    g.replay()
        return replay


def throughput_plan(batch: int, n_fft: int = 1024) -> dict:
    """How to run ``batch`` independent streams on one MI355X for throughput, by the measurements in DESIGN.md (sections 4.6-4.10;
    ``profiles/r03_v6_*``, ``profiles/r04_group_sweep.txt``): ``{"queues", "pipes", "depth", "split", "group"}`` for ``hop_pipeline`` /
    ``QueuedHopPipelines``.  Every choice gives the same samples; what changes is how many hops lie between handing a hop over and its output.
    ``group`` > 0: input that can be handed over ``group`` hops at a time goes through hop groups (``submit_group``: whole Griffin-Lim chains
    per launch, nothing parked); ``depth`` is what a caller that submits hop by hop gets instead.
      up to 384 streams    one pipe; groups of four hops (batch 256: 36.9 us per hop) -- hop by hop: four hops in flight (45.9 us; 54.7 at depth 1)
      385 .. 1,023         one pipe; groups of two hops, two streams a chain workgroup (512 streams: 6.9 M frames/s) -- hop by hop: depth 2 (6.4 M)
      1,024 .. 2,047       two pipes on two HIP streams, split hops, two hops in flight (1,024 streams: 8.0 M frames/s against 6.9 M)
      2,048 and up         an even number of pipes of about 1,024 streams, taking turns on two HIP streams, split hops, one hop in flight
                           (2,048 / 4,096 / 8,192 streams: 8.4 M frames/s against 7.2 / 7.4 / 7.7 M for one pipe)
    n_fft 1536 (the wavefront-per-stream schedule is not built there): one pipe at depth 1."""
    if n_fft != 1024:
        return {"queues": 1, "pipes": 1, "depth": 1, "split": False, "group": 0}
    if batch <= 384:
        return {"queues": 1, "pipes": 1, "depth": 4, "split": False, "group": 4}
    if batch < 1024:
        return {"queues": 1, "pipes": 1, "depth": 2, "split": False, "group": 2}
    if batch < 2048:
        return {"queues": 2, "pipes": 2, "depth": 2, "split": True, "group": 0}
    return {"queues": 2, "pipes": 2 * (batch // 2048), "depth": 1, "split": True, "group": 0}


def hop_pipeline(denoiser: "Denoiser", batch: int, grouped: bool = False):
    """The frame-mode pipe ``throughput_plan`` picks for ``batch`` streams: a ``HopPipeline`` or ``QueuedHopPipelines`` (same ``submit`` /
    ``flush``; the queued form also wants ``after()`` / ``before()`` or ``synchronize()`` around the caller's own stream).  ``grouped``: the caller
    hands hops over in groups (``submit_group`` with ``pipe.group`` hops at a time) where the plan has a group size."""
    plan = throughput_plan(batch, denoiser.n_fft)
    if plan["queues"] > 1:
        return QueuedHopPipelines(denoiser, batch, queues=plan["queues"], depth=plan["depth"], split=plan["split"], pipes=plan["pipes"])
    pipe = HopPipeline(denoiser, batch)
    if grouped and plan["group"] > 0:
        pipe.set_group(plan["group"])
    else:
        pipe.set_depth(plan["depth"])
    return pipe


class DenoiserStream:
    """B concurrent streams with persistent device state: input ring, output overlap-add buffer, hx.

    ``push(chunk)`` mirrors one ``recv`` call of the reference for every stream at once
    (app3.py:174-226): append samples, run one hop per ``hop_length`` new samples once ``n_fft``
    samples are buffered, return the emitted output samples.  No added latency (one launch per hop,
    ``dn_stream_step``); frame f's Griffin-Lim draws from ``seed + f``."""

    def __init__(self, denoiser: Denoiser, batch: int, stream_id0: int = 0, seed: int = 0):
        self.dn, self.batch, self.stream_id0, self.seed = denoiser, batch, stream_id0, seed
        d = denoiser
        dev = d.device
        self.ring = torch.zeros(batch, d.n_fft, dtype=torch.float32, device=dev)
        self.ola = torch.zeros(batch, d.n_fft, dtype=torch.float32, device=dev)     # app3.py:133
        self.hx = d.init_hx(batch)
        self.pending = torch.zeros(batch, 0, dtype=torch.float32, device=dev)      # samples not yet shifted in
        self.filled = 0          # samples accepted so far, saturating at n_fft - hop (priming)
        self.hops = 0

    def push(self, chunk: torch.Tensor, init_angles_per_hop=None) -> torch.Tensor:
        d = self.dn
        if chunk.device != d.device or chunk.dtype != torch.float32 or chunk.shape[0] != self.batch:
            raise ValueError("chunk must be float32 (B, n) on the denoiser's device")
        self.pending = torch.cat([self.pending, chunk], dim=1)
        prime = d.n_fft - d.hop
        if self.filled < prime:                  # the first frame needs n_fft samples: fill ring[hop:] first
            take = min(prime - self.filled, self.pending.shape[1])
            self.ring[:, d.hop + self.filled: d.hop + self.filled + take] = self.pending[:, :take]
            self.pending = self.pending[:, take:]
            self.filled += take
        outs = []
        ws = d._workspace(self.batch)
        model_h = d.model._native(d.device)
        i = 0
        while self.filled == prime and self.pending.shape[1] >= d.hop:
            hop_in = self.pending[:, :d.hop].contiguous()
            self.pending = self.pending[:, d.hop:]
            hop_out = torch.empty(self.batch, d.hop, dtype=torch.float32, device=d.device)
            ia = None if init_angles_per_hop is None else init_angles_per_hop[i]
            keep, ia_ptr = d._angles_ptr(ia, self.batch)
            with torch.cuda.device(d.device):
                st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                d.lib.check(d.lib.dn_stream_step(model_h, d.plan.handle, hop_in.data_ptr(), self.ring.data_ptr(), self.ola.data_ptr(),
                                                 self.hx.data_ptr(), hop_out.data_ptr(), ia_ptr, self.seed + self.hops, self.stream_id0,
                                                 d.n_iter, d.momentum, ws.data_ptr(), self.batch, d._flags(), st))
            outs.append(hop_out)
            self.hops += 1
            i += 1
        if not outs:
            return torch.zeros(self.batch, 0, dtype=torch.float32, device=d.device)
        return torch.cat(outs, dim=1)
