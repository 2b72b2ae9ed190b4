#!/usr/bin/env python3
"""Graph replay of K captured pushes per launch against eager pushes (tools only): python tools/graph_hops.py [batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from audio_denoising_amd.pipeline import PipelinedStream  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
dn = bench.build_denoiser(dev)
for K in (0, 1, 2, 4, 8):
    ps = PipelinedStream(dn, B)
    hop = (0.1 * torch.randn(max(K, 1), B, dn.hop)).to(dev)
    out = torch.empty_like(hop)
    if K == 0:
        step = lambda: ps.push_(hop[0], out[0], check_weights=False)  # noqa: E731
    else:
        g = ps.graph_step(hop if K > 1 else hop[0], out if K > 1 else out[0])
        step = g.replay
    n = 400 // max(K, 1)
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    print(f"batch {B} {'eager' if K == 0 else f'graph of {K}'}: {1e6 * (time.perf_counter() - t0) / (n * max(K, 1)):.1f} us/hop")
    ps.flush()
