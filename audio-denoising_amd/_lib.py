"""ctypes binding of the C ABI declared in include/dn_denoise.h.

The product path loads ``lib/libdn_denoise.so`` (hipcc, gfx950) that sits next to this file and
FAILS LOUDLY when it is missing -- there is no CPU or eager-PyTorch fallback anywhere in this
package.  (``DnLib(path)`` can bind another build of the same ABI; the tests use that to bind
the host-emulation build of the kernel sources, never the product.)
"""
from __future__ import annotations

import ctypes as C
import os
import threading

# PyTorch must load ITS HIP runtime first: libdn_denoise.so then binds to the same libamdhip64 instance (same device
# context, same streams).  Loaded the other way round, the process ends up with a runtime that sees no device.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# DN_LIB_PATH binds another build of the same sources (the stamped diagnostic build of tools/gl_probe.py); unset = the product
LIB_PATH = os.environ.get("DN_LIB_PATH") or os.path.join(_HERE, "lib", "libdn_denoise.so")

# every symbol include/dn_denoise.h declares
SYMBOLS = (
    "dn_model_create", "dn_model_destroy", "dn_cell_forward", "dn_cell_forward_ex", "dn_cell_forward_bf16", "dn_stft_general", "dn_server_rows", "dn_istft_general", "dn_dsp_create", "dn_dsp_destroy",
    "dn_dsp_get_tables", "dn_stft", "dn_stft_mel_log1p", "dn_mel_scale", "dn_invmel", "dn_residual_invmel",
    "dn_griffinlim", "dn_griffinlim_draw_phases", "dn_synthesis", "dn_istft", "dn_workspace_bytes", "dn_process_frame", "dn_stream_step",
    "dn_pipe_create", "dn_pipe_destroy", "dn_pipe_set_model", "dn_pipe_set_head_start", "dn_pipe_set_gl_schedule", "dn_pipe_set_split", "dn_pipe_set_depth", "dn_pipe_set_group", "dn_pipe_submit_group", "dn_pipe_stream_push_group", "dn_pipe_stream_flush_group", "dn_pipe_reserve_parity", "dn_pipe_get_counters", "dn_pipe_submit", "dn_pipe_flush", "dn_pipe_stream_create", "dn_pipe_stream_push",
    "dn_pipe_stream_flush", "dn_pipe_stream_push_host", "dn_pipe_stream_host_wait", "dn_pipe_stream_get_state", "dn_pipe_stream_set_state", "dn_momo_create", "dn_momo_destroy",
    "dn_momo_forward", "dn_last_error", "dn_abi_version",
)

DN_PEAK_NORMALIZE = 1
DN_PRE_WINDOW = 2
DN_CONV_BF16 = 1
DN_GL_AUTO, DN_GL_WAVE_PER_COLUMN, DN_GL_WAVE_PER_STREAM = 0, 1, 2
DN_HOST_STAGED = 1
DN_HOST_DEFER = 2
DN_SPLIT_AUTO, DN_SPLIT_OFF, DN_SPLIT_ON = -1, 0, 1
ABI_VERSION = 4


class ModelCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("num_compressed_bins", "in_size", "n_levels", "hidden_size",
                                         "kernel_size", "stride", "padding", "num_gaussians")]


class MomoCfg(C.Structure):
    _fields_ = [("num_compressed_bins", C.c_int32), ("in_size", C.c_int32), ("n_levels", C.c_int32), ("hidden_size", C.c_int32),
                ("kernel_size", C.c_int32), ("stride", C.c_int32), ("paddings", C.c_int32 * 3), ("num_gaussians", C.c_int32)]


class DspCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("sample_rate", "n_fft", "hop", "n_mels")]


class DnError(RuntimeError):
    """A C-ABI call returned a negative status; the message is dn_last_error()."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"[dn status {code}] {msg}")
        self.code = code


class DnLib:
    def __init__(self, path: str = LIB_PATH):
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: the HIP extension has not been built. Run "
                f"`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C audio-denoising_amd/csrc`). "
                f"This package has no CPU fallback.")
        self.path = path
        self.lib = C.CDLL(path)
        L, p, i32, u32, u64, f32, vp = self.lib, C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64, C.c_float, C.c_void_p
        L.dn_last_error.restype = C.c_char_p
        L.dn_abi_version.restype = C.c_int
        L.dn_model_create.argtypes = [vp, C.c_size_t, C.POINTER(ModelCfg), C.POINTER(vp)]
        L.dn_model_destroy.argtypes = [vp]
        L.dn_model_destroy.restype = None
        L.dn_cell_forward.argtypes = [vp, p, p, p, p, i32, i32, i32, i32, vp]
        L.dn_cell_forward_bf16.argtypes = [vp, p, p, p, p, i32, i32, i32, i32, vp]
        L.dn_cell_forward_ex.argtypes = [vp, p, p, p, p, i32, i32, i32, i32, f32, vp]
        L.dn_stft_general.argtypes = [vp, p, p, p, i32, i32, vp]
        L.dn_server_rows.argtypes = [vp, p, p, p, p, i32, vp]
        L.dn_istft_general.argtypes = [vp, p, p, i32, i32, vp]
        L.dn_momo_create.argtypes = [vp, C.c_size_t, C.POINTER(MomoCfg), C.POINTER(vp)]
        L.dn_momo_destroy.argtypes = [vp]
        L.dn_momo_destroy.restype = None
        L.dn_momo_forward.argtypes = [vp, p, p, p, p, p, p, i32, i32, i32, i32, vp]
        L.dn_dsp_create.argtypes = [C.POINTER(DspCfg), vp, vp, vp, C.POINTER(vp)]
        L.dn_dsp_destroy.argtypes = [vp]
        L.dn_dsp_destroy.restype = None
        L.dn_dsp_get_tables.argtypes = [vp, vp, vp, vp]
        L.dn_stft.argtypes = [vp, p, p, i32, u32, vp]
        L.dn_stft_mel_log1p.argtypes = [vp, p, p, p, i32, u32, vp]
        L.dn_mel_scale.argtypes = [vp, p, p, i32, i32, vp]
        L.dn_invmel.argtypes = [vp, p, p, i32, i32, vp]
        L.dn_residual_invmel.argtypes = [vp, p, p, p, i32, i32, vp]
        L.dn_griffinlim.argtypes = [vp, p, p, u64, u64, p, p, i32, i32, f32, vp]
        L.dn_griffinlim_draw_phases.argtypes = [vp, u64, u64, p, i32, vp]
        L.dn_synthesis.argtypes = [vp, p, p, p, u64, u64, p, p, i32, i32, f32, vp]
        L.dn_istft.argtypes = [vp, p, p, i32, vp]
        L.dn_workspace_bytes.argtypes = [vp, i32]
        L.dn_workspace_bytes.restype = C.c_size_t
        L.dn_process_frame.argtypes = [vp, vp, p, p, p, p, p, u64, u64, i32, f32, vp, i32, u32, vp]
        L.dn_stream_step.argtypes = [vp, vp, p, p, p, p, p, p, u64, u64, i32, f32, vp, i32, u32, vp]
        L.dn_pipe_create.argtypes = [vp, vp, i32, u32, C.POINTER(vp)]
        L.dn_pipe_destroy.argtypes = [vp]
        L.dn_pipe_destroy.restype = None
        L.dn_pipe_set_model.argtypes = [vp, vp]
        L.dn_pipe_set_head_start.argtypes = [vp, i32]
        L.dn_pipe_set_gl_schedule.argtypes = [vp, i32]
        L.dn_pipe_set_split.argtypes = [vp, i32]
        L.dn_pipe_set_depth.argtypes = [vp, i32]
        L.dn_pipe_set_group.argtypes = [vp, i32]
        i64 = C.c_int64
        L.dn_pipe_submit_group.argtypes = [vp, p, i64, p, p, i64, p, i64, u64, u64, i32, i32, f32, vp]
        L.dn_pipe_stream_push_group.argtypes = [vp, p, i64, i32, p, i64, i32, p, i64, u64, u64, i32, f32, vp]
        L.dn_pipe_stream_flush_group.argtypes = [vp, p, i64, i32, C.POINTER(i32), vp]
        L.dn_pipe_reserve_parity.argtypes = [vp]
        L.dn_pipe_get_counters.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(i32), vp]
        L.dn_pipe_submit.argtypes = [vp, p, p, p, p, u64, u64, i32, f32, vp]
        L.dn_pipe_flush.argtypes = [vp, i32, f32, vp]
        L.dn_pipe_stream_create.argtypes = [vp, vp, i32, u32, C.POINTER(vp)]
        L.dn_pipe_stream_push.argtypes = [vp, p, i32, p, i32, p, u64, u64, i32, f32, vp]
        L.dn_pipe_stream_flush.argtypes = [vp, p, i32, i32, f32, vp]
        L.dn_pipe_stream_push_host.argtypes = [vp, p, i32, p, i32, u64, u64, i32, f32, u32, vp, C.POINTER(u64)]
        L.dn_pipe_stream_host_wait.argtypes = [vp, u64]
        L.dn_pipe_stream_get_state.argtypes = [vp, p, p, p, vp]
        L.dn_pipe_stream_set_state.argtypes = [vp, p, p, p, u64, vp]
        if L.dn_abi_version() != ABI_VERSION:
            raise ImportError(f"{path}: ABI version {L.dn_abi_version()} != {ABI_VERSION}; rebuild the extension")

    def check(self, rc: int) -> None:
        if rc != 0:
            raise DnError(rc, self.lib.dn_last_error().decode("utf-8", "replace"))

    def __getattr__(self, name):
        return getattr(self.lib, name)


_lock = threading.Lock()
_default: DnLib | None = None


def get_lib() -> DnLib:
    """The product's library (HIP build).  Raises ImportError when it has not been built."""
    global _default
    with _lock:
        if _default is None:
            _default = DnLib(LIB_PATH)
        return _default
