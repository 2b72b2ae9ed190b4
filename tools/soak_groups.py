#!/usr/bin/env python3
"""Soak of the group pipe (dn_pipe_set_group): many launches back to back in frame mode and in streaming mode (eager and replaying one captured group
push), with the device-resident counters checked at the end and the last group compared bit for bit with a one-hop pipe that was fed the same hops.
    python tools/soak_groups.py [launches] [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from audio_denoising_amd.pipeline import HopPipeline, PipelinedStream  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
H = 4
dev = torch.device("cuda", 0)
dn = bench.build_denoiser(dev)
g = torch.Generator().manual_seed(7)
frames = (0.1 * torch.randn(H, B, dn.n_fft, generator=g)).to(dev)
out = torch.empty_like(frames)
hx = dn.init_hx(B)
pipe = HopPipeline(dn, B)
pipe.set_group(H)
t0 = time.perf_counter()
for i in range(n):
    pipe.submit_group(frames, hx, out, seed=1, check_weights=False)
pipe.flush()
torch.cuda.synchronize()
el = time.perf_counter() - t0
pushes, fr, pending = pipe.counters()
assert fr == n * H and pending is False and torch.isfinite(out).all() and torch.isfinite(hx).all(), (pushes, fr, pending)
print(f"frame mode: {n} launches of {H} hops x {B} streams in {el:.2f} s = {1e6 * el / (n * H):.2f} us per hop; frames counter {fr}")
# the last group again through a one-hop pipe that starts from the hidden state the soak reached one group earlier: same frame indices -> same bits
hx_ref = dn.init_hx(B)
one = HopPipeline(dn, B)
grp = HopPipeline(dn, B)
grp.set_group(H)
hx_g = dn.init_hx(B)
o1, o2 = torch.empty_like(frames), torch.empty_like(frames)
for k in range(50):
    for h in range(H):
        one.submit(frames[h], hx_ref, o1[h], seed=1, check_weights=False)
    grp.submit_group(frames, hx_g, o2, seed=1, check_weights=False)
one.flush()
grp.flush()
torch.cuda.synchronize()
assert torch.equal(o1, o2) and torch.equal(hx_ref, hx_g)
print("200 hops through the group pipe and through the one-hop pipe: identical frames and hx")
# streaming, eager then captured
hops = (0.1 * torch.randn(H, B, dn.hop, generator=g)).to(dev)
outs = torch.empty_like(hops)
ps = PipelinedStream(dn, B)
ps.set_group(H)
m = max(n // 4, 100)
t0 = time.perf_counter()
for i in range(m):
    ps.push_group_(hops, outs, check_weights=False)
torch.cuda.synchronize()
e1 = time.perf_counter() - t0
ps._bind()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    ps.push_group_(hops, outs, check_weights=False)
t0 = time.perf_counter()
for i in range(m):
    graph.replay()
torch.cuda.synchronize()
e2 = time.perf_counter() - t0
tail, valid = ps.flush_group()
torch.cuda.synchronize()
pushes, fr, pending = ps.counters()
assert pushes == 2 * m * H and fr == pushes - 1 and pending is False and torch.isfinite(outs).all() and valid == H, (pushes, fr, pending, valid)
print(f"streaming: {m} eager group pushes {1e6 * e1 / (m * H):.2f} us per hop, {m} replays of one captured push {1e6 * e2 / (m * H):.2f} us per hop; "
      f"pushes counter {pushes}, frames {fr}")
