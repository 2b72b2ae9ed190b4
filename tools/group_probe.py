#!/usr/bin/env python3
"""Timeline of one launch of a group pipe (dn_pipe_set_group): the four chain wavefronts of workgroup 0, front workgroup 0 hop by hop, and
the spread of workgroup end times over the grid (stamped diagnostic build, make probe), in s_memtime ticks relative to the earliest start.
    python tools/group_probe.py [batch] [hops per launch] [n_iter]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DN_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audio-denoising_amd", "lib", "libdn_probe.so"))
import torch  # noqa: E402

import bench  # noqa: E402
from audio_denoising_amd.pipeline import HopPipeline  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
dn = bench.build_denoiser(dev, "S", os.environ.get("DN_CONV", "fp32"))
if len(sys.argv) > 3:
    dn.n_iter = int(sys.argv[3])
frames = (0.1 * torch.randn(H, B, dn.n_fft)).to(dev)
hx = dn.init_hx(B)
out = torch.empty_like(frames)
pipe = HopPipeline(dn, B)
pipe.set_group(H)
for i in range(100):
    pipe.submit_group(frames, hx, out, seed=1, check_weights=False)
torch.cuda.synchronize()
g = (C.c_uint64 * 32)()
assert dn.lib.lib.dn_probe_read_group_glw(g) == 0
f = (C.c_uint64 * 16)()
assert dn.lib.lib.dn_probe_read_group_front(f) == 0
w = (C.c_uint64 * 4096)()
assert dn.lib.lib.dn_probe_read_group_wg(w) == 0
n_wg = min(2 * B, 2048)
# (s_memtime is a per-XCD counter: stamps of different workgroups compare only within an XCD -- workgroup 0 and front workgroup 0 share XCD 0 when the
# batch is a multiple of 8 -- so the grid is summarised by each workgroup's own duration)
dur = [w[2 * i + 1] - w[2 * i] for i in range(n_wg)]
t0 = min(g[8 * k] for k in range(4))
names = ["entry", "-", "magnitudes + phases drawn, tables in LDS", "first synthesis", "one iteration", "loop done", "-", "frame stored"]
print(f"batch {B}, {H} hops per launch, n_iter {dn.n_iter}  (ticks of s_memtime, relative to the entry of chain workgroup 0)")
for k in range(4):
    print(f"  chain wave {k} of workgroup 0: " + ", ".join(f"{names[i]} {g[8 * k + i] - t0}" for i in (0, 2, 3, 4, 5, 7) if g[8 * k + i] >= t0))
if B % 8 == 0:
    for h in range(H):
        v = [f[4 * h + i] - t0 for i in range(4)]
        print(f"  front workgroup 0, hop {h}: start {v[0]}, analysis done {v[1]} (+{v[1] - v[0]}), model done {v[2]} (+{v[2] - v[1]}), inverse mel done {v[3]} (+{v[3] - v[2]})")
cd, fd = sorted(dur[:min(B, n_wg)]), sorted(dur[B:n_wg])
q = lambda a, p: a[min(len(a) - 1, int(p * len(a)))]
print(f"  chain workgroups, own duration: min {cd[0]}, median {q(cd, .5)}, p90 {q(cd, .9)}, p99 {q(cd, .99)}, max {cd[-1]}")
if fd:
    print(f"  front workgroups, own duration: min {fd[0]}, median {q(fd, .5)}, p90 {q(fd, .9)}, p99 {q(fd, .99)}, max {fd[-1]}")
