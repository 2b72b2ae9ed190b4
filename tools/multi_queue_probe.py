#!/usr/bin/env python3
"""B streams as Q independent pipes on Q HIP streams (tools only): python tools/multi_queue_probe.py total Q split [depth]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from audio_denoising_amd.pipeline import HopPipeline  # noqa: E402

total, Q, split = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 1
n_hw = int(sys.argv[5]) if len(sys.argv) > 5 else Q          # HIP streams: pipe i runs on stream i % n_hw
dev = torch.device("cuda", 0)
dn = bench.build_denoiser(dev)
g = torch.Generator().manual_seed(1)
N = total // Q
pipes = []
hw = [torch.cuda.Stream() for _ in range(n_hw)] if Q > 1 else [torch.cuda.current_stream()]
for q in range(Q):
    frames = (0.1 * torch.randn(N, dn.n_fft, generator=g)).to(dev)
    pipe = HopPipeline(dn, N)
    pipe.set_depth(depth)
    if depth == 1:
        pipe.set_gl_schedule(2)
    pipe.set_split(split)
    pipes.append((pipe, frames, dn.init_hx(N), torch.empty_like(frames), hw[q % len(hw)]))


def run(steps):
    for i in range(steps):
        for pipe, fr, hx, out, st in pipes:
            with torch.cuda.stream(st):
                pipe.submit(fr, hx, out, seed=1, check_weights=False)
    torch.cuda.synchronize()


steps = max(20, 200 * 1024 // total)
run(steps // 2)
t0 = time.perf_counter()
run(steps)
us = 1e6 * (time.perf_counter() - t0) / steps
print(f"{total} streams as {Q} pipe(s) of {N} on {len(hw)} HIP stream(s), depth {depth}, split {split}: {us:.1f} us per hop  {total / us:.3f} M frames/s")
