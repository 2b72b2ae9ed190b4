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


def batch256():
    """All 256 streams of the metric's batch against the oracle: one-launch hop, one-hop pipe (head start), depth-4 pipe; fp32 and bf16 convs."""
    from audio_denoising_amd.pipeline import Denoiser, HopPipeline
    from oracle import dsp_ref, pipeline_ref
    dev = torch.device("cuda", 0)
    p = pipeline_ref.PARAMS_S
    g = torch.Generator().manual_seed(1234)
    hops = [0.1 * torch.randn(256, p.n_fft, generator=g) for _ in range(2)]
    inits = [torch.rand(256, p.n_stft, 3, dtype=torch.complex64, generator=torch.Generator().manual_seed(4321 + i)) for i in range(2)]
    fb = dsp_ref.melscale_fbanks(p.n_stft, p.n_mels, p.sample_rate)
    sd = t._state_dict("dari_tult")
    refs, h = [], torch.zeros(256, 17, 5)
    with torch.no_grad():
        for i in range(2):
            r = pipeline_ref.process_frame(sd, hops[i], h, p, fb, init_angles=inits[i])
            h = r["hx"]
            refs.append(r)
    for conv in ("fp32", "bf16"):
        m = t._model(dev, 5)
        m.conv_precision = conv
        dn = Denoiser(m, p.sample_rate, p.n_fft, p.hop, p.n_mels)
        out, hx, resid = dn.process_frame(hops[0].to(dev), None, init_angles=inits[0].to(dev), return_residual=True)
        e = (out.cpu() - refs[0]["out"]).numpy()
        print(f"batch 256 {conv} one-launch hop: residual {float((resid.cpu() - refs[0]['predicted_diff']).abs().max()):.2e}, hx {float((hx.cpu() - refs[0]['hx']).abs().max()):.2e}, "
              f"waveform rms {np.sqrt((e ** 2).mean()):.2e} max {np.abs(e).max():.2e} (signal rms {float(refs[0]['out'].pow(2).mean().sqrt()):.2e})")
        for depth in (1, 4):
            pipe = HopPipeline(dn, 256)
            pipe.set_depth(depth)
            hxp = dn.init_hx(256)
            outs = [torch.empty(256, p.n_fft, device=dev) for _ in range(2)]
            for i in range(2):
                pipe.submit(hops[i].to(dev), hxp, outs[i], seed=0, init_angles=inits[i].to(dev))
            pipe.flush()
            torch.cuda.synchronize()
            for i in range(2):
                e = (outs[i].cpu() - refs[i]["out"]).numpy()
                print(f"batch 256 {conv} pipe depth {depth} hop {i}: waveform rms {np.sqrt((e ** 2).mean()):.2e} max {np.abs(e).max():.2e}")
            print(f"    hx after two hops {float((hxp.cpu() - refs[1]['hx']).abs().max()):.2e}")


if __name__ == "__main__":
    main()
    batch256()
