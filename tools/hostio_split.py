#!/usr/bin/env python3
"""Which direction of the zero-copy host transport costs the launch its time?  (tools only)
    python tools/hostio_split.py [batch] [depth]
The same streaming push with its input / output in device memory or in page-locked host memory (all four combinations)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from audio_denoising_amd.pipeline import PipelinedStream  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
dn = bench.build_denoiser(dev)
ps = PipelinedStream(dn, B)
if depth > 1:
    ps.set_depth(depth)
hop = (0.1 * torch.randn(B, dn.hop) * 32767).to(torch.int16)
h_in, h_out = hop.clone().pin_memory(), torch.zeros_like(hop).pin_memory()
d_in, d_out = hop.to(dev), torch.zeros_like(hop).to(dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, i, o in (("device in, device out", d_in, d_out), ("host in, device out", h_in, d_out), ("device in, host out", d_in, h_out), ("host in, host out", h_in, h_out)):
    def run(n):
        for _ in range(n):
            ps.lib.check(ps.lib.dn_pipe_stream_push(ps.handle, i.data_ptr(), 1, o.data_ptr(), 1, None, 0, 0, 32, 0.99, st))
        torch.cuda.synchronize()
    run(300)
    t0 = time.perf_counter()
    run(500)
    print(f"batch {B} depth {depth} {name}: {1e6 * (time.perf_counter() - t0) / 500:.1f} us/hop")
