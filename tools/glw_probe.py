#!/usr/bin/env python3
"""Timeline of one launch of a deep / wavefront-per-stream pipe: the four wavefronts of Griffin-Lim workgroup 0 and front workgroup 0
(stamped diagnostic build, make probe), in s_memtime ticks relative to the earliest start.
    python tools/glw_probe.py batch depth [n_iter]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DN_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audio-denoising_amd", "lib", "libdn_probe.so"))
import torch  # noqa: E402

import bench  # noqa: E402
from audio_denoising_amd import _lib  # noqa: E402
from audio_denoising_amd.pipeline import HopPipeline  # noqa: E402

B, depth = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
dn = bench.build_denoiser(dev, "S", os.environ.get("DN_CONV", "fp32"))
if len(sys.argv) > 3:
    dn.n_iter = int(sys.argv[3])
frames = (0.1 * torch.randn(B, dn.n_fft)).to(dev)
hx = dn.init_hx(B)
out = torch.empty_like(frames)
pipe = HopPipeline(dn, B)
pipe.set_depth(depth)
if depth == 1:
    pipe.set_gl_schedule(_lib.DN_GL_WAVE_PER_STREAM)
for i in range(300):
    pipe.submit(frames, hx, out, seed=1, check_weights=False)
torch.cuda.synchronize()
g = (C.c_uint64 * 32)()
assert dn.lib.lib.dn_probe_read_glw(g) == 0
w = (C.c_uint64 * 8)()
assert dn.lib.lib.dn_probe_read_hop_wg(w) == 0
t0 = min([g[8 * k] for k in range(4) if g[8 * k]] + [w[2]])
names = ["entry", "tables", "state loaded", "first synthesis", "one iteration", "loop done", "parked", "exit"]
print(f"batch {B} depth {depth} n_iter {dn.n_iter}")
for k in range(4):
    print(f"  chain wave {k}: " + ", ".join(f"{names[i]} {g[8 * k + i] - t0 if g[8 * k + i] >= t0 else '-'}" for i in range(8)))
print(f"  front WG 0: start {w[2] - t0}, stft done {w[5] - t0}, cell done {w[6] - t0}, P1-P10 done {w[3] - t0}")
