#!/usr/bin/env python3
"""Soak: N hops of 1,024 streams through two queued split pipes and through one pipe; hx and the last outputs must agree bit for bit.  (tools only)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from audio_denoising_amd.pipeline import HopPipeline, QueuedHopPipelines  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
B = 1024
dev = torch.device("cuda", 0)
dn = bench.build_denoiser(dev)
g = torch.Generator().manual_seed(3)
frames = [(0.1 * torch.randn(B, dn.n_fft, generator=g)).to(dev) for _ in range(4)]
res = []
for kind in ("one", "queued"):
    pipe = HopPipeline(dn, B) if kind == "one" else QueuedHopPipelines(dn, B, queues=2, depth=2)
    if kind == "one":
        pipe.set_depth(2)
    hx = dn.init_hx(B)
    outs = [torch.empty(B, dn.n_fft, device=dev) for _ in range(4)]
    torch.cuda.synchronize()
    for i in range(N):
        pipe.submit(frames[i & 3], hx, outs[i & 3], seed=i, check_weights=False)
    pipe.flush()
    if kind == "queued":
        pipe.synchronize()
    torch.cuda.synchronize()
    res.append([o.clone() for o in outs] + [hx.clone()])
ok = all(torch.equal(x, y) for x, y in zip(*res))
print(f"{N} hops of {B} streams: queued == one pipe: {ok}; finite: {all(bool(torch.isfinite(t).all()) for t in res[1])}; max |out| {max(float(t.abs().max()) for t in res[1][:4]):.3f}")
sys.exit(0 if ok else 1)
