#!/usr/bin/env python3
"""Do a chain-only launch and a front-only launch of two DIFFERENT pipes overlap when they sit on two HIP streams?  (tools only)
    python tools/two_queue_probe.py [streams per pipe] [split 0|1]
Two independent pipes of N streams each, one per HIP stream, against one pipe of 2N streams."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from audio_denoising_amd.pipeline import HopPipeline  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
split = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
dn = bench.build_denoiser(dev)
g = torch.Generator().manual_seed(1)


def make(B):
    frames = (0.1 * torch.randn(B, dn.n_fft, generator=g)).to(dev)
    pipe = HopPipeline(dn, B)
    pipe.set_gl_schedule(2)
    return pipe, frames, dn.init_hx(B), torch.empty_like(frames)


def run(pipes, streams, steps):
    for i in range(steps):
        for (pipe, fr, hx, out), st in zip(pipes, streams):
            with torch.cuda.stream(st):
                pipe.submit(fr, hx, out, seed=1, check_weights=False)
    torch.cuda.synchronize()


one = [make(2 * N)]
one[0][0].set_split(0)
two = [make(N), make(N)]
for p in two:
    p[0].set_split(split)
s0 = torch.cuda.current_stream()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for pipes, streams, name in ((one, [s0], f"one pipe of {2 * N}"), (two, [sa, sb], f"two pipes of {N} on two streams (split {split})"), (two, [s0, s0], f"two pipes of {N} on one stream (split {split})")):
    run(pipes, streams, 100)
    t0 = time.perf_counter()
    run(pipes, streams, 200)
    print(f"{name}: {1e6 * (time.perf_counter() - t0) / 200:.1f} us per hop of {2 * N} streams")
