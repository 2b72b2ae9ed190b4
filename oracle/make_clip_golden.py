#!/usr/bin/env python3
"""Realistic-signal fixture for BASELINE configs[0] at its stated size (TEST INFRASTRUCTURE): ONE 3 s, 16 kHz mono clip
(93 hops of 512 at n_fft 1024), batch 1, GRUUNet2-dari_tult weights, through the CPU oracle's streaming loop
(oracle/pipeline_ref.StreamRef = app3.py:178-226) with injected Griffin-Lim phases.

Run ONLY in the build container, where /root/reference is mounted:  python oracle/make_clip_golden.py

The signal is DATA from the reference's own tree -- data/uncompressed/cats/extras/sequences/I_BLE01_EU_FN_DEL01_2SEQ2.wav
(8 kHz mono s16, read with the stdlib `wave` module) -- resampled to 16 kHz (scipy.signal.resample_poly, up 2) with white
noise at -30 dB re. its peak added (seeded), then quantised to int16 as the app's transport delivers it (app3.py:168-172).
The expected output comes from the oracle, whose DSP half is a restatement (PARITY UNPINNED, oracle/__init__.py).
Initial phases are not stored: frame f uses torch.rand((1, 513, 3), complex64, Generator(seed=4321 + f))."""
import os
import sys
import wave

import numpy as np
import scipy.signal as ss
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
SRC = "/root/reference/data/uncompressed/cats/extras/sequences/I_BLE01_EU_FN_DEL01_2SEQ2.wav"
N_HOPS = 93


def clip_init_angles(f: int, n_stft: int = 513) -> torch.Tensor:
    return torch.rand(1, n_stft, 3, dtype=torch.complex64, generator=torch.Generator().manual_seed(4321 + f))


def main():
    from oracle import model_ref, pipeline_ref
    p = pipeline_ref.PARAMS_S
    with wave.open(SRC, "rb") as w:
        assert w.getframerate() == 8000 and w.getnchannels() == 1 and w.getsampwidth() == 2
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").astype(np.float64) / 32768.0
    n = p.n_fft + p.hop * (N_HOPS - 1)                      # 48,128 samples = 3.008 s
    x = ss.resample_poly(pcm, 2, 1)[2000:2000 + n]          # skip the leading silence of the recording
    assert x.shape[0] == n, x.shape
    x = 0.5 * x / np.abs(x).max()
    x = x + 10 ** (-30 / 20) * 0.5 * torch.randn(n, generator=torch.Generator().manual_seed(7), dtype=torch.float64).numpy()
    s16 = np.clip(np.round(x * 32767.0), -32768, 32767).astype(np.int16)
    sig = torch.from_numpy(s16.astype(np.float32) / np.float32(32767))[None]      # app3.py:172
    sd = model_ref.unflatten_weights(np.fromfile(os.path.join(REPO, "tests", "golden", "weights_dari_tult.bin"), dtype=np.float32))
    st = pipeline_ref.StreamRef(sd, p, 1)
    with torch.no_grad():
        y = st.push(sig, init_angles_per_hop=[clip_init_angles(f) for f in range(N_HOPS)])
    assert y.shape == (1, N_HOPS * p.hop)
    out = os.path.join(REPO, "tests", "golden", "clip_S.npz")
    np.savez_compressed(out, signal_s16=s16, out=y.numpy(), ola=st.ola.numpy(), hx=st.hx.numpy(), source=np.array(os.path.relpath(SRC, "/root/reference")))
    print(out, os.path.getsize(out), "bytes; signal rms", float(sig.pow(2).mean().sqrt()), "out rms", float(y.pow(2).mean().sqrt()))


if __name__ == "__main__":
    main()
