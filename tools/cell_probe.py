#!/usr/bin/env python3
"""Where one GRUUNet2 forward (T = 3) spends its cycles: the stamped diagnostic build (make -C audio-denoising_amd/csrc probe)
at batch 256, phases of workgroup 0 of the stand-alone cell kernel (tools/gl_probe.py is the same for Griffin-Lim)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DN_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audio-denoising_amd", "lib", "libdn_probe.so"))
import torch  # noqa: E402

import bench  # noqa: E402

NAMES = ["hx + pinned gate weights", "d0 (1->17)", "d1 (17->17)", "d2 (17->17)", "d3 (17->51)", "gru t=0", "gru t=1", "gru t=2",
         "u0 (17->17)", "u1 (34->17)", "u2 (34->17)", "u3 (34->1, VALU)"]


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda", 0)
    dn = bench.build_denoiser(dev)
    x = torch.rand(batch, 3, 80, device=dev) * 6
    for _ in range(20):
        dn.model(x)
    torch.cuda.synchronize()
    buf = (C.c_uint64 * 32)()
    assert dn.lib.lib.dn_probe_read_cell(buf) == 0
    t = [buf[i] for i in range(13)]
    tot = t[12] - t[0]
    print(f"cell forward batch {batch}: {tot} ticks")
    for i, n in enumerate(NAMES):
        print(f"    {n:28s} {t[i + 1] - t[i]:6d}  {100.0 * (t[i + 1] - t[i]) / tot:5.1f} %")
    f = [buf[i] for i in range(17, 22)]
    print(f"  inside d1, wave 0: stage requests {f[0] - t[2]}, bias + operand loads {f[1] - f[0]}, 30 MFMAs issued {f[2] - f[1]}, "
          f"results + stores {f[3] - f[2]}, stage commit {f[4] - f[3]}, barrier {t[3] - f[4]}")


if __name__ == "__main__":
    main()
