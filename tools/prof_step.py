#!/usr/bin/env python3
"""Profiling driver: a few hops at batch 256 (same workload and schedule as bench.py: hop groups of four up to 384 streams, else the pipe at its
default depth), nothing else.
    python tools/prof_step.py [steps] [batch] [serial | group H | depth D]
Run under rocprofv3 (--kernel-trace --stats, or a --pmc pass) from the repo root."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else bench.BATCH
    mode = sys.argv[3] if len(sys.argv) > 3 else ""
    arg = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    dev = torch.device("cuda", 0)
    dn = bench.build_denoiser(dev)
    g = torch.Generator().manual_seed(1234)
    group = arg if mode == "group" else 0 if mode in ("serial", "depth") else bench.default_group(batch, bench.N_FFT)
    G = max(group, 1)
    frames = (0.1 * torch.randn(G, batch, bench.N_FFT, generator=g)).to(dev)
    hx = dn.init_hx(batch)
    out = torch.empty_like(frames)
    if mode == "serial":
        for i in range(steps):
            dn.process_frame_(frames[0], hx, out[0], seed=1000 + i, stream_id0=0)
    else:
        from audio_denoising_amd.pipeline import HopPipeline
        pipe = HopPipeline(dn, batch)
        if group > 0:
            pipe.set_group(group)
            for i in range(0, steps, group):
                pipe.submit_group(frames, hx, out, seed=1000, stream_id0=0)
        else:
            pipe.set_depth(arg if mode == "depth" else bench.default_depth(batch, bench.N_FFT))
            for i in range(steps):
                pipe.submit(frames[0], hx, out[0], seed=1000, stream_id0=0)
        pipe.flush()
    torch.cuda.synchronize()
    print("ok", float(out.abs().mean()))


if __name__ == "__main__":
    main()
