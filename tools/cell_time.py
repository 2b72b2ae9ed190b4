#!/usr/bin/env python3
"""Time GRUUNet2.forward (batch 256, T=3, F=80) with fp32 and bf16 MFMA conv tiles (HIP events)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
dn = bench.build_denoiser(dev)
m = dn.model
x = torch.rand(256, 3, 80, device=dev) * 6
hx = torch.zeros(256, 17, 5, device=dev)
for prec in ("fp32", "bf16"):
    m.conv_precision = prec
    for _ in range(20):
        m(x, hx)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        m(x, hx)
    e1.record()
    torch.cuda.synchronize()
    print(prec, f"{1e3 * e0.elapsed_time(e1) / 200:.1f} us per forward (batch 256, incl. Python dispatch)")
