#!/usr/bin/env python3
"""Where does a host-fed push spend its time?  (tools only)
    python tools/hostio_probe.py [batch] [depth] [staged|zc] [none|memmove|nt]
(feed "nt": gcc -O2 -shared -fPIC -o tools/build/nt_copy.so tools/nt_copy.c first)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from audio_denoising_amd.pipeline import HostFedStream  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 1
staged = len(sys.argv) > 3 and sys.argv[3] == "staged"
feed = sys.argv[4] if len(sys.argv) > 4 else "none"       # none | memmove | nt : rewrite the input buffer before every push
dev = torch.device("cuda", 0)
dn = bench.build_denoiser(dev)
hs = HostFedStream(dn, B, depth=depth, staged=staged)
hop = (0.1 * torch.randn(B, dn.hop) * 32767).to(torch.int16)
for _ in range(200):
    hs.push(hop, copy=False)
torch.cuda.synchronize()
tp, tw = [], []
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
nt = None
if feed == "nt":
    nt = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "build", "nt_copy.so")).nt_copy
    nt.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    nt.restype = None
nbytes = hop.numel() * 2
tf = []
for i in range(300):
    k = hs._n % hs.RING
    tf0 = time.perf_counter()
    if feed == "memmove":
        C.memmove(hs._pin_in[k].data_ptr(), hop.data_ptr(), nbytes)
    elif feed == "nt":
        nt(hs._pin_in[k].data_ptr(), hop.data_ptr(), nbytes)
    t0 = time.perf_counter()
    tf.append(t0 - tf0)
    ticket = C.c_uint64()
    hs.lib.check(hs.lib.dn_pipe_stream_push_host(hs.handle, hs._pin_in[k].data_ptr(), 1, hs._pin_out[k].data_ptr(), 1, 0, 0, 32, 0.99, hs._hflags, st, C.byref(ticket)))
    t1 = time.perf_counter()
    hs._n += 1
    hs.lib.check(hs.lib.dn_pipe_stream_host_wait(hs.handle, hs._n - 1 - hs.LAG))
    t2 = time.perf_counter()
    tp.append(t1 - t0)
    tw.append(t2 - t1)
torch.cuda.synchronize()
import numpy as np
print(f"batch {B} depth {depth} {'staged' if staged else 'zero copy'} feed {feed} ({1e6 * np.median(tf):.1f} us): total {1e6 * (np.median(tf) + np.median(tp) + np.median(tw)):.1f} us/hop; push call median {1e6 * np.median(tp):.1f} us, wait median {1e6 * np.median(tw):.1f} us (p90 {1e6 * np.quantile(tw, 0.9):.1f})")

# the class's own push(), as bench.py drives it
import cProfile, pstats
for _ in range(200):
    hs.push(hop, copy=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    hs.push(hop, copy=False)
torch.cuda.synchronize()
print(f"HostFedStream.push loop: {1e6 * (time.perf_counter() - t0) / 2000:.1f} us/hop")
pr = cProfile.Profile()
pr.enable()
for _ in range(1000):
    hs.push(hop, copy=False)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
