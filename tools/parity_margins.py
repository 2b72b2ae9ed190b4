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


def attribution():
    """The 256 metric frames against the float64 yardstick (tests/golden/metric_f64_B256.npz), GPU and fp32 CPU oracle side by side."""
    from audio_denoising_amd.pipeline import Denoiser, HopPipeline
    from oracle import dsp_ref, pipeline_ref
    dev = torch.device("cuda", 0)
    p = pipeline_ref.PARAMS_S
    f64 = t.load_golden("metric_f64_B256.npz")
    frames = 0.1 * torch.randn(256, p.n_fft, generator=torch.Generator().manual_seed(int(f64["frames_seed"])))
    init = torch.rand(256, p.n_stft, 3, dtype=torch.complex64, generator=torch.Generator().manual_seed(int(f64["init_seed"])))
    dn = Denoiser(t._model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    out, hx, resid = dn.process_frame(frames.to(dev), None, init_angles=init.to(dev), return_residual=True)
    with torch.no_grad():
        ref = pipeline_ref.process_frame(t._state_dict("dari_tult"), frames, torch.zeros(256, 17, 5), p, dsp_ref.melscale_fbanks(p.n_stft, p.n_mels, p.sample_rate),
                                         init_angles=init)
    eg = out.cpu().numpy().astype(np.float64) - f64["out"]
    ec = ref["out"].numpy().astype(np.float64) - f64["out"]
    egc = out.cpu().numpy().astype(np.float64) - ref["out"].numpy().astype(np.float64)
    sig = np.sqrt((f64["out"] ** 2).mean())
    qs = (0.1, 0.5, 0.9, 0.99, 1.0)
    fmt = lambda a: "  ".join(f"{np.quantile(a, x):.2e}" for x in qs)
    print(f"float64 attribution, 256 metric frames (signal RMS {sig:.2e}); quantiles 10 % / 50 % / 90 % / 99 % / max over streams")
    for name, e in (("gpu - f64", eg), ("cpu fp32 oracle - f64", ec), ("gpu - cpu fp32 oracle", egc)):
        print(f"  {name:24s} batch RMS {np.sqrt((e ** 2).mean()):.2e} max-abs {np.abs(e).max():.2e}")
        print(f"      per-stream RMS      {fmt(np.sqrt((e ** 2).mean(axis=1)))}")
        print(f"      per-stream max-abs  {fmt(np.abs(e).max(axis=1))}")
    g_rms, c_rms = np.sqrt((eg ** 2).mean(axis=1)), np.sqrt((ec ** 2).mean(axis=1))
    ratio = g_rms / np.maximum(c_rms, 1e-12)
    print(f"  per-stream ratio |gpu - f64| / |cpu - f64| (RMS): {fmt(ratio)};  streams with ratio > 1: {(ratio > 1).sum()}, > 2: {(ratio > 2).sum()}")
    print(f"  worst CPU streams {np.argsort(-c_rms)[:5].tolist()}: cpu {np.sort(c_rms)[::-1][:5]}, gpu there {g_rms[np.argsort(-c_rms)[:5]]}")
    rg = np.abs(resid.cpu().numpy() - f64["predicted_diff"]).reshape(256, -1).max(axis=1)
    rc = np.abs(ref["predicted_diff"].numpy() - f64["predicted_diff"]).reshape(256, -1).max(axis=1)
    print(f"  mel residual max-abs per stream: gpu {fmt(rg)} | cpu fp32 {fmt(rc)}")
    print(f"  hx max-abs: gpu {np.abs(hx.cpu().numpy() - f64['hx']).max():.2e} | cpu fp32 {np.abs(ref['hx'].numpy() - f64['hx']).max():.2e}")
    # the same comparison as a function of the number of Griffin-Lim iterations: the arithmetic is as accurate as the CPU's at every length of the
    # chain; what grows is the chain's own amplification of ANY rounding (the float64 result moves as much when its input moves by one fp32 ulp)
    sub = slice(0, 64)
    rgn = np.random.default_rng(5)
    for n_it in (0, 1, 2, 4, 8, 16, 32):
        dn_k = Denoiser(t._model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels, n_iter=n_it)
        o_k, _ = dn_k.process_frame(frames[sub].to(dev), None, init_angles=init[sub].to(dev))
        with torch.no_grad():
            y_c = dsp_ref.griffinlim(ref["lin_mag"][sub], p.n_fft, p.hop, init_angles=init[sub], n_iter=n_it) * ref["peak"][sub, None]
        y64 = t._f64_frames([frames[sub]], [init[sub]], p, n_iter=n_it)[0][0]
        y64p = t._f64_frames([frames[sub] * torch.from_numpy(1 + 6e-8 * rgn.standard_normal(frames[sub].shape)).float()], [init[sub]], p, n_iter=n_it)[0][0]
        a, b, c = o_k.cpu().numpy() - y64, y_c.numpy() - y64, y64p - y64
        r = lambda e: np.sqrt((e ** 2).mean(axis=1))
        print(f"  n_iter {n_it:2d} (64 streams) per-stream RMS median / max: gpu {np.median(r(a)):.1e} / {r(a).max():.1e} | cpu fp32 {np.median(r(b)):.1e} / {r(b).max():.1e} | "
              f"float64 with the frames perturbed by 6e-8 relative {np.median(r(c)):.1e} / {r(c).max():.1e}")
    # the bench's schedule (hop groups of four) on the same frames: bit-identical to the one-launch hop by construction, measured anyway
    pipe = HopPipeline(dn, 256)
    pipe.set_group(4)
    hxg = dn.init_hx(256)
    fr4 = frames.to(dev)[None].expand(1, -1, -1).contiguous()
    o4 = torch.empty_like(fr4)
    pipe.submit_group(fr4, hxg, o4, seed=0, init_angles=init.to(dev)[None])
    pipe.flush()
    torch.cuda.synchronize()
    print(f"  group pipe == one-launch hop on these frames: {bool(torch.equal(o4[0], out))}")


if __name__ == "__main__":
    main()
    batch256()
    attribution()
