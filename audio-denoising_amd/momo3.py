"""Drop-in ``MOMO3`` for the reference's sibling model ``momo3.MOMO3`` (momo3.py:247-324; checkpoint saves/MOMO3-4d4ea0).

Same constructor, same ``forward(input, hx=None, prev=None) -> (out, hx)``, same ``state_dict`` keys (the reference checkpoint
loads with ``load_state_dict``), same ``hparams`` / ``get_config()`` / ``latent_size`` / ``num_compressed_bins``.  The forward is
one HIP kernel per call (``dn_momo_forward``: the GRUUNet2 conv tiles generalised to per-level paddings, plus the frame-delta
channel of momo3.py:285-289).  As in the reference, ``prev`` (the frame before ``input[:, 0]``) is the caller's to carry across
calls -- ``forward`` does not return it (momo3.py:300-324); ``last_frame(input)`` gives the value to pass next time.
GPU only: no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import weakref

import torch
from torch import nn

from . import _lib
from .gruunet2 import _Blocks, _ConvBlock, _Holder


class _UpBlocks(_Holder):
    """UpBlocks has no position code of its own (momo3.py:159-189): no ``gs`` buffer in the state_dict."""

    def __init__(self, convs):
        super().__init__()
        self.ups = nn.ModuleList([_ConvBlock(c) for c in convs])


class _MomoCell(_Holder):
    """Parameter layout of MOMOCell (momo3.py:191-226)."""

    def __init__(self, in_size, hidden_sizes, kernel_sizes, strides, paddings, num_gaussians):
        super().__init__()
        g, hs = num_gaussians, list(hidden_sizes)
        hs2 = hs[:-1] + [3 * hs[-1]]
        sizes = [in_size + g] + hs2                  # DownBlocks: the position code is concatenated once, at the input
        self.input_gate = _Blocks("downs", [nn.Conv1d(sizes[i], sizes[i + 1], kernel_size=kernel_sizes[i], stride=strides[i], padding=paddings[i])
                                            for i in range(len(hs))], g)
        self.reset_gate = _Blocks("downs", [nn.Conv1d(hs[-1] + g, 3 * hs[-1], kernel_size=3, stride=1, padding=1)], g)
        rs = ([1] + hs)[::-1]
        rk, rst, rp = list(kernel_sizes)[::-1], list(strides)[::-1], list(paddings)[::-1]
        self.output_gate = _UpBlocks([nn.ConvTranspose1d(rs[i] if i == 0 else 2 * rs[i], rs[i + 1], kernel_size=rk[i], stride=rst[i], padding=rp[i])
                                      for i in range(len(hs))])


class MOMO3(nn.Module):
    def __init__(self, num_compressed_bins, in_size, hidden_sizes, kernel_sizes, strides, paddings, num_gaussians=6):
        super().__init__()
        self.hparams = dict(num_compressed_bins=num_compressed_bins, in_size=in_size, hidden_sizes=hidden_sizes,
                            kernel_sizes=kernel_sizes, strides=strides, paddings=paddings, num_gaussians=num_gaussians)
        self.latent_size = hidden_sizes[-1]
        self.num_compressed_bins = num_compressed_bins
        self.cell = _MomoCell(in_size + 1, hidden_sizes, kernel_sizes, strides, paddings, num_gaussians)     # momo3.py:260
        self._supported = (in_size == 1 and len(hidden_sizes) == 3 and all(h == 16 for h in hidden_sizes) and
                           all(k == 3 for k in kernel_sizes) and all(s == 2 for s in strides) and
                           all(p in (0, 1) for p in paddings) and num_gaussians == 6)

    def get_config(self):
        return self.hparams

    @staticmethod
    def last_frame(input: torch.Tensor) -> torch.Tensor:
        """What to pass as ``prev`` on the next call: the last frame of this one, shaped (B, 1, F) as momo3.py:289 keeps it."""
        x = input.unsqueeze(0) if input.dim() == 2 else input
        return x[:, -1:, :].detach().clone()

    def compressed_bins(self, F: int) -> int:
        L = F
        for p in self.hparams["paddings"]:
            L = (L + 2 * p - 3) // 2 + 1
        return L

    def _native(self, device: torch.device):
        tensors = list(self.state_dict(keep_vars=True).values())
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        per_model = _NATIVE.setdefault(self, {})
        hit = per_model.get(idx)
        if hit is not None and hit[0] == key:
            return hit[1].handle
        if not self._supported:
            raise NotImplementedError("the HIP kernels are built for the architecture of the reference's MOMO3 checkpoint (in_size 1, 3 levels, "
                                      "hidden 16, kernel 3, stride 2, paddings in {0,1}, 6 gaussians); got " + repr(self.hparams))
        lib = _lib.get_lib()
        blob = torch.cat([v.detach().reshape(-1).to(device="cpu", dtype=torch.float32) for v in tensors]).contiguous()
        cfg = _lib.MomoCfg(int(self.num_compressed_bins), 1, 3, 16, 3, 2, (C.c_int32 * 3)(*[int(p) for p in self.hparams["paddings"]]), 6)
        handle = C.c_void_p()
        with torch.cuda.device(idx):
            lib.check(lib.dn_momo_create(C.c_void_p(blob.data_ptr()), blob.numel(), C.byref(cfg), C.byref(handle)))
        per_model[idx] = (key, _Owner(lib, handle))
        return handle

    def forward(self, input, hx=None, prev=None):
        """input (B,T,F) or (T,F); hx (B,16,C) or None; prev (B,1,F) / (B,F) or None -> (out like input, hx).  momo3.py:300-324."""
        two_dimmed = input.dim() == 2
        if two_dimmed:
            input = input.unsqueeze(0)
        if input.dim() != 3:
            raise RuntimeError(f"expected a (B,T,F) or (T,F) input, got {tuple(input.shape)}")
        if not input.is_cuda:
            raise RuntimeError("MOMO3 (MI355X build) runs on the GPU only; there is no CPU path in this package")
        if input.dtype != torch.float32:
            raise TypeError(f"the HIP kernels compute in float32; got {input.dtype}")
        B, T, F = input.shape
        if hx is None:
            hx = torch.zeros(B, self.latent_size, self.num_compressed_bins, dtype=input.dtype, device=input.device)
        if hx.dim() != 3 or hx.shape[0] != B or hx.shape[1] != self.latent_size or hx.device != input.device or hx.dtype != input.dtype:
            raise RuntimeError(f"hx must be ({B}, {self.latent_size}, C) with the dtype and device of the input; got {tuple(hx.shape)}")
        Cb = hx.shape[2]
        if self.compressed_bins(F) != Cb:
            raise RuntimeError(f"The size of tensor a ({self.compressed_bins(F)}) must match the size of tensor b ({Cb}) at non-singleton "
                               f"dimension 2 (input of {F} bins compresses to {self.compressed_bins(F)}, hx has {Cb})")
        p = None
        if prev is not None:
            p = prev.detach().reshape(B, F).to(dtype=torch.float32).contiguous()
            if p.device != input.device:
                raise RuntimeError("prev must live on the input's device")
        lib = _lib.get_lib()
        handle = self._native(input.device)
        x, h0 = input.detach().contiguous(), hx.detach().contiguous()
        out, h1 = torch.empty_like(x), torch.empty_like(h0)
        with torch.cuda.device(input.device):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            lib.check(lib.dn_momo_forward(handle, x.data_ptr(), h0.data_ptr(), None if p is None else p.data_ptr(), out.data_ptr(), h1.data_ptr(),
                                          None, B, T, F, Cb, st))
        return (out.squeeze(0) if two_dimmed else out), h1


class _Owner:
    def __init__(self, lib, handle):
        self.handle = handle
        self._fin = weakref.finalize(self, lib.dn_momo_destroy, handle)


_NATIVE: "weakref.WeakKeyDictionary[MOMO3, dict]" = weakref.WeakKeyDictionary()
