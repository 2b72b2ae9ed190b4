#!/usr/bin/env python3
"""Measured parity margins of the fused hop against the committed goldens (what the -m gpu tests assert, printed):
mel-residual and hx max-abs error (tolerance 1e-4), waveform RMS / max-abs error with shared Griffin-Lim phases."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as t  # noqa: E402  (helpers only: goldens, reference-checkpoint model)


def main():
    from audio_denoising_amd.pipeline import Denoiser
    dev = torch.device("cuda", 0)
    for tag in ("S", "R2", "R1"):
        p = t._params(tag)
        g = t.load_golden(f"dsp_{tag}.npz")
        dn = Denoiser(t._model(dev, p.num_compressed_bins), p.sample_rate, p.n_fft, p.hop, p.n_mels)
        frames = torch.from_numpy(g["frames"]).to(dev)
        out, hx, resid = dn.process_frame(frames, None, init_angles=torch.from_numpy(g["init_angles"]).to(dev), return_residual=True)
        o = out.cpu().numpy()
        e = o - g["out"]
        print(f"{tag}: residual max-abs err {np.abs(resid.cpu().numpy() - g['predicted_diff']).max():.2e}, hx {np.abs(hx.cpu().numpy() - g['hx']).max():.2e}, "
              f"waveform rms err {np.sqrt((e ** 2).mean()):.2e} (signal rms {np.sqrt((g['out'] ** 2).mean()):.2e}), max-abs err {np.abs(e).max():.2e}")


if __name__ == "__main__":
    main()
