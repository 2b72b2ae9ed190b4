#!/usr/bin/env python3
"""GRUUNet2.forward at batch 256 with fp32 and with bf16 MFMA conv tiles, a few launches each (run under rocprofv3 --pmc / --kernel-trace:
the two precisions are different kernels, so the summary separates them).   python tools/cell_both.py [launches]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
dn = bench.build_denoiser(dev)
m = dn.model
x = torch.rand(256, 3, 80, device=dev) * 6
hx = torch.zeros(256, 17, 5, device=dev)
for prec in ("fp32", "bf16"):
    m.conv_precision = prec
    for _ in range(n):
        m(x, hx)
torch.cuda.synchronize()
print("ok")
