#!/usr/bin/env python3
"""When do the two kinds of workgroups of one pipelined launch start and finish?  Stamped diagnostic build (make probe), batch 256:
Griffin-Lim workgroup 0 and front workgroup 0 of the last launch, in s_memtime ticks relative to the earlier start.
    DN_GL_HEAD_START=n python tools/hop_wg_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DN_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audio-denoising_amd", "lib", "libdn_probe.so"))
import torch  # noqa: E402

import bench  # noqa: E402
from audio_denoising_amd.pipeline import HopPipeline  # noqa: E402

dev = torch.device("cuda", 0)
dn = bench.build_denoiser(dev, os.environ.get("DN_PRESET", "S"))
B = 256
frames = (0.1 * torch.randn(B, dn.n_fft)).to(dev)
hx = dn.init_hx(B)
out = torch.empty_like(frames)
pipe = HopPipeline(dn, B)
for i in range(300):
    pipe.submit(frames, hx, out, seed=1, check_weights=False)
torch.cuda.synchronize()
buf = (C.c_uint64 * 8)()
assert dn.lib.lib.dn_probe_read_hop_wg(buf) == 0
t = [buf[i] for i in range(7)]
t0 = min(t[0], t[2])
print(f"head start {os.environ.get('DN_GL_HEAD_START', 'default')}: Griffin-Lim WG 0: start {t[0] - t0}, end {t[1] - t0};  "
      f"front WG 0: start {t[2] - t0}, stft done {t[5] - t0}, cell done {t[6] - t0}, P1-P10 done {t[3] - t0}, head start done {t[4] - t0 if t[4] > t[2] else '-'}  (ticks)")
inv = (C.c_uint64 * 8)()
if dn.lib.lib.dn_probe_read_hop_invmel(inv) == 0:
    v = [inv[i] for i in range(6)]
    print(f"    inverse mel of front WG 0 (factored form): requests + residual -> mel {v[1] - v[0]}, banded G^-1 mel {v[2] - v[1]}, "
          f"fb rows + stores {v[3] - v[2]}  (ticks)")
st = (C.c_uint64 * 8)()
if dn.lib.lib.dn_probe_read_hop_stft(st) == 0:
    v = [st[i] for i in range(7)]
    print(f"    analysis of front WG 0, wave 0: frame load + peak {v[1] - v[0]}, normalise + window -> LDS {v[2] - v[1]}, constants + column {v[3] - v[2]}, "
          f"fft + split {v[4] - v[3]}, magnitudes {v[5] - v[4]}, mel + log1p {v[6] - v[5]}  (ticks)")
