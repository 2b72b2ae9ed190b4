#!/usr/bin/env python3
"""Time the software-pipelined hop for one (batch, depth, Griffin-Lim schedule, n_iter) combination: microseconds per hop.
    python tools/pipe_time.py batch depth schedule n_iter [steps]
Used for the depth / schedule experiments of DESIGN.md section 8 (n_iter = 0 leaves the fixed cost of a launch)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    batch, depth, sched, n_iter = (int(x) for x in sys.argv[1:5])
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 300
    dev = torch.device("cuda", 0)
    dn = bench.build_denoiser(dev, os.environ.get("DN_PRESET", "S"), os.environ.get("DN_CONV", "fp32"))
    dn.n_iter = n_iter
    from audio_denoising_amd.pipeline import HopPipeline
    g = torch.Generator().manual_seed(1234)
    frames = (0.1 * torch.randn(batch, dn.n_fft, generator=g)).to(dev)
    hx = dn.init_hx(batch)
    out = torch.empty_like(frames)
    pipe = HopPipeline(dn, batch)
    pipe.set_depth(depth)
    if depth == 1:
        pipe.set_gl_schedule(sched)
    if len(sys.argv) > 6:
        pipe.set_head_start(int(sys.argv[6]))
    import time
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.4:          # bring the GPU out of its idle power state
        for _ in range(50):
            pipe.submit(frames, hx, out, seed=1, check_weights=False)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        pipe.submit(frames, hx, out, seed=1, check_weights=False)
    e1.record()
    pipe.flush()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / steps
    print(f"batch {batch} depth {depth} sched {sched} n_iter {n_iter}: {us:.1f} us/hop  {batch / us:.3f} M frames/s")


if __name__ == "__main__":
    main()
