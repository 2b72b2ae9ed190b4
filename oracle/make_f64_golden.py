#!/usr/bin/env python3
"""Float64 yardstick for the metric's batch (TEST INFRASTRUCTURE): tests/golden/metric_f64_B256.npz.

Run ONLY in the build container, where /root/reference is mounted:

    python oracle/make_f64_golden.py

The 256 synthetic frames of BASELINE configs[1] (seed 1234) and shared Griffin-Lim phases (seed 4321) through
oracle/pipeline_np64.process_frame64 -- float64 numpy DSP on the fp32 window / filterbank constants -- with the model stage
run by the REFERENCE's own gruunet2.GRUUNet2 cast to float64 (`.double()`), imported exactly as
oracle/make_golden.py imports it.  The fixture holds results only (float64 waveform, mel residual, hx, peak);
the test regenerates the inputs from the seeds.  oracle/model_ref.forward on float64 weights must agree with the
reference class to 1e-12 (checked here), so tests can also evaluate other inputs in float64 without the reference.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden  # noqa: E402


def metric_inputs(batch=256, n_fft=1024, n_stft=513):
    g = torch.Generator().manual_seed(1234)
    frames = 0.1 * torch.randn(batch, n_fft, generator=g)
    init = torch.rand(batch, n_stft, 3, dtype=torch.complex64, generator=torch.Generator().manual_seed(4321))
    return frames, init


def main():
    gruunet2 = make_golden.import_reference_model()
    from oracle import dsp_ref, model_ref, pipeline_np64, pipeline_ref
    p = pipeline_ref.PARAMS_S
    frames, init = metric_inputs(256, p.n_fft, p.n_stft)
    m, ck = make_golden.load_reference(gruunet2, "GRUUNet2-dari_tult", p.num_compressed_bins)
    m = m.double()
    sd64 = {k: v.double() for k, v in ck["model_state_dict"].items()}

    def model64(x, hx):
        with torch.no_grad():
            out, h = m(torch.from_numpy(x), torch.from_numpy(hx))
            out2, h2 = model_ref.forward(sd64, torch.from_numpy(x), torch.from_numpy(hx))
        assert (out - out2).abs().max().item() <= 1e-12 and (h - h2).abs().max().item() <= 1e-12
        return out.numpy(), h.numpy()

    fb = dsp_ref.melscale_fbanks(p.n_stft, p.n_mels, p.sample_rate).numpy()           # the fp32 constants the fp32 paths use
    window = torch.hann_window(p.n_fft).numpy()
    hx0 = np.zeros((256, 17, p.num_compressed_bins))
    r = pipeline_np64.process_frame64(frames.numpy(), hx0, model64, window, fb, init.numpy(), p.n_fft, p.hop)
    # the fp32 oracle beside it, for the record (the test recomputes it)
    with torch.no_grad():
        r32 = pipeline_ref.process_frame(model_ref.unflatten_weights(np.fromfile(os.path.join(make_golden.GOLD, "weights_dari_tult.bin"), dtype=np.float32)),
                                         frames, torch.zeros(256, 17, p.num_compressed_bins), p, torch.from_numpy(fb), init_angles=init)
    e = r32["out"].numpy().astype(np.float64) - r["out"]
    print(f"fp32 oracle vs float64: waveform RMS {np.sqrt((e ** 2).mean()):.3e}, max-abs {np.abs(e).max():.3e}; "
          f"residual max-abs {np.abs(r32['predicted_diff'].numpy() - r['predicted_diff']).max():.3e}; signal RMS {np.sqrt((r['out'] ** 2).mean()):.3e}")
    np.savez_compressed(os.path.join(make_golden.GOLD, "metric_f64_B256.npz"), frames_seed=1234, init_seed=4321,
                        out=r["out"], hx=r["hx"], predicted_diff=r["predicted_diff"], peak=r["peak"])


if __name__ == "__main__":
    main()
