#!/usr/bin/env python3
"""Where one Griffin-Lim iteration spends its cycles: runs the stamped diagnostic build (make -C audio-denoising_amd/csrc probe)
at batch 256 and prints, per wavefront (= STFT column) of workgroup 0, the s_memtime deltas between the stamps of iteration 10.
    DN_LIB_PATH=audio-denoising_amd/lib/libdn_probe.so python tools/gl_probe.py [serial|hop] [batch]
Read the SHARES, not the total: the stamps' own waits forbid overlaps the product build has."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DN_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audio-denoising_amd", "lib", "libdn_probe.so"))
import torch  # noqa: E402

import bench  # noqa: E402

NAMES = ["merge", "ifft", "window+ola store", "barrier", "column build (lds read)", "fft", "split", "update"]


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "serial"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    dev = torch.device("cuda", 0)
    dn = bench.build_denoiser(dev, os.environ.get("DN_PRESET", "S"))
    g = torch.Generator().manual_seed(1234)
    frames = (0.1 * torch.randn(batch, dn.n_fft, generator=g)).to(dev)
    hx = dn.init_hx(batch)
    out = torch.empty_like(frames)
    if mode == "serial":
        for i in range(30):
            dn.process_frame_(frames, hx, out, seed=1000 + i, stream_id0=0)
        reader = dn.lib.lib.dn_probe_read_gl
    else:
        from audio_denoising_amd.pipeline import HopPipeline
        pipe = HopPipeline(dn, batch)
        for i in range(30):
            pipe.submit(frames, hx, out, seed=1000, stream_id0=0)
        pipe.flush()
        reader = dn.lib.lib.dn_probe_read_hop
    torch.cuda.synchronize()
    buf = (C.c_uint64 * 48)()
    rc = reader(buf)
    assert rc == 0, rc
    for w in range(3):
        t = [buf[w * 16 + i] for i in range(9)]
        d = [t[i + 1] - t[i] for i in range(8)]
        tot = t[8] - t[0]
        print(f"{mode} batch {batch} column {w}: iteration = {tot} ticks")
        for n, v in zip(NAMES, d):
            print(f"    {n:28s} {v:6d}  {100.0 * v / tot:5.1f} %")
    # alignment of the three waves: stamp 3 (arrive at barrier) and 4 (leave)
    a = [buf[w * 16 + 3] for w in range(3)]
    l = [buf[w * 16 + 4] for w in range(3)]
    print("barrier arrival skew (ticks, relative to the first):", [x - min(a) for x in a], "leave:", [x - min(a) for x in l])


if __name__ == "__main__":
    main()
