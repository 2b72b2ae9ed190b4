#!/usr/bin/env python3
"""Do the Griffin-Lim workgroup and the front workgroup of a stream-per-CU launch really share the CUs one and one?  Stamped diagnostic build:
every workgroup of the last launch records the CU it ran on; prints the time per hop and the census of CUs by (Griffin-Lim workgroups, front workgroups).
    python tools/pairing_probe.py frames|stream depth [batch]"""
import collections
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DN_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audio-denoising_amd", "lib", "libdn_probe.so"))
import torch  # noqa: E402

import bench  # noqa: E402
from audio_denoising_amd.pipeline import HopPipeline, PipelinedStream  # noqa: E402

mode, depth = sys.argv[1], int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda", 0)
dn = bench.build_denoiser(dev, os.environ.get("DN_PRESET", "S"))
if mode == "stream":
    ps = PipelinedStream(dn, B)
    ps.set_depth(depth)
    hop = (0.1 * torch.randn(B, dn.hop)).to(dev)
    out = torch.empty_like(hop)
    step = lambda: ps.push_(hop, out, check_weights=False)
else:
    pipe = HopPipeline(dn, B)
    pipe.set_depth(depth)
    frames = (0.1 * torch.randn(B, dn.n_fft)).to(dev)
    hx = dn.init_hx(B)
    out = torch.empty_like(frames)
    step = lambda: pipe.submit(frames, hx, out, seed=1, check_weights=False)
t_pre = time.perf_counter()
while time.perf_counter() - t_pre < 0.4:
    for _ in range(50):
        step()
    torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(1000):
    step()
torch.cuda.synchronize()
us = (time.perf_counter() - t0) * 1e3
buf = (C.c_uint32 * 2048)()
assert dn.lib.lib.dn_probe_read_blk_hw(buf) == 0
spb = 4 // depth if depth > 1 else (4 if B >= 768 else 1)
nb = (B + spb - 1) // spb
cus = collections.defaultdict(lambda: [0, 0])
for i in range(min(2048, nb + B)):
    hw = buf[i]
    key = (hw >> 28, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)
    cus[key][0 if i < nb else 1] += 1
hist = collections.Counter(tuple(v) for v in cus.values())
print(f"{mode} depth {depth} batch {B}: {us:.1f} us/hop; CUs by (Griffin-Lim, front) workgroups: {dict(hist)}")
tb = (C.c_uint64 * 4096)()
assert dn.lib.lib.dn_probe_read_blk_t(tb) == 0
import numpy as np
t = np.array(tb[:], dtype=np.int64).reshape(2048, 2)[:nb + B]
t0 = t[:, 0].min()
dur = t[:, 1] - t[:, 0]
for name, sl in (("Griffin-Lim", slice(0, nb)), ("front", slice(nb, nb + B))):
    d, st, en = dur[sl], t[sl, 0] - t0, t[sl, 1] - t0
    print(f"  {name} workgroups: start {st.min()}..{st.max()}, duration min {d.min()} median {int(np.median(d))} max {d.max()}, end max {en.max()} (block {int(np.argmax(en))})")
xcc = np.array([buf[i] >> 28 for i in range(nb + B)])
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"  XCC {x}: {int(m.sum())} workgroups, end max {int((t[m, 1] - t0).max())}, start min {int((t[m, 0] - t0).min())}")

