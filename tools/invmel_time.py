#!/usr/bin/env python3
"""Stand-alone time of the inverse-mel contraction kernel (dn_residual_invmel) at batch 256."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda", 0)
dn = bench.build_denoiser(dev)
B = 256
x = torch.rand(B, 3, 80, device=dev)
d = torch.rand(B, 3, 80, device=dev)
lin = torch.empty(B, 3, 513, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(200):
    dn.lib.check(dn.lib.dn_residual_invmel(dn.plan.handle, x.data_ptr(), d.data_ptr(), lin.data_ptr(), B, 3, st))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(500):
    dn.lib.check(dn.lib.dn_residual_invmel(dn.plan.handle, x.data_ptr(), d.data_ptr(), lin.data_ptr(), B, 3, st))
e1.record()
torch.cuda.synchronize()
print("invmel stand-alone, batch 256: %.2f us per launch" % (e0.elapsed_time(e1) / 500 * 1e3))
