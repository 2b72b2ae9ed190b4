#!/usr/bin/env python3
"""Headline benchmark: denoised 32 ms frames/s (16 kHz, n_fft 1024, hop 512, 80 mels) at batch 256 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the whole per-hop path (P1..P12 of SURVEY.md section 8a: peak-normalise, Hann,
3-column STFT, mel, log1p, GRUUNet2 x3 steps, residual, expm1, inverse mel, 32-iteration Griffin-Lim,
`* peak`) over one batch of 256 synthetic frames per GPU, inputs resident in HBM, hidden state carried
from step to step.  Streams are independent, so N GPUs run N x 256 streams with no data-path collective
(weak scaling); the only collectives are the barriers and the MAX of the elapsed time.

Besides the contract fields the JSON line carries
  roofline     -- the dominant kernel (hop_kernel: one launch = one hop of the batch): algorithmic FLOPs per launch /
                  its average launch duration measured with HIP events in this process, against the fp32 compute peak;
  cpu_baseline -- the CPU oracle (the reference's op sequence restated on torch-CPU, oracle/pipeline_ref.py)
                  timed on this host's cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

BATCH = 256
SR, N_FFT, HOP, N_MELS = 16000, 1024, 512, 80
N_STFT = N_FFT // 2 + 1
GL_ITERS = 32
# algorithmic work per frame (real FFT of length N counted as 2.5 N log2 N = 25,600 flop)
FLOP_PER_RFFT = 2.5 * N_FFT * 10
GL_FFTS = GL_ITERS * 6 + 3
GL_FLOP_PER_FRAME = GL_FFTS * FLOP_PER_RFFT + 246240              # 4.992 MFLOP of FFTs + the fused inverse-mel contraction
TOTAL_FLOP_PER_FRAME = 198 * FLOP_PER_RFFT + 2 * 246240 + 939150  # 6.50 MFLOP
HBM_BYTES_PER_FRAME = 4 * N_FFT + 4 * N_FFT + 2 * 4 * 17 * 5       # 8,872 B compulsory
GL_HBM_BYTES_PER_FRAME = 2 * 4 * 3 * N_MELS + 4 * N_FFT + 4        # kernel-level: model input + residual in, waveform out, peak
PEAK_FP32_TFLOPS = 157.3                                          # MI355X_MICROARCH.md: vector == matrix fp32 peak
PEAK_HBM_GBS = 8000.0


# parameter presets: S = BASELINE.json's synthetic config (the metric); R1 = the reference app's own (app3.py:13-33);
# R2 = server.py:166-170.  Only S is the headline; the others are reported by `--preset` for DESIGN.md.
PRESETS = {"S": (16000, 1024, 512, 80, "dari_tult"), "R1": (48000, 1536, 768, 64, "dari_tult2"), "R2": (48000, 1024, 512, 64, "dari_tult")}


def build_denoiser(dev, preset="S"):
    from audio_denoising_amd.gruunet2 import GRUUNet2
    from audio_denoising_amd.pipeline import Denoiser
    sr, n_fft, hop, n_mels, ckpt = PRESETS[preset]
    blob = np.fromfile(os.path.join(REPO, "tests", "golden", f"weights_{ckpt}.bin"), dtype=np.float32)
    model = GRUUNet2(n_mels // 16, 1, (17, 17, 17, 17), (3, 3, 3, 3), (2, 2, 2, 2), (1, 1, 1, 1))
    keys = list(model.state_dict().keys())
    sd, off = {}, 0
    for k in keys:
        n = model.state_dict()[k].numel()
        sd[k] = torch.from_numpy(blob[off:off + n].copy()).reshape(model.state_dict()[k].shape)
        off += n
    model.load_state_dict(sd)
    model.eval().to(dev)
    return Denoiser(model, sr, n_fft, hop, n_mels, n_iter=GL_ITERS)


def staged_kernel_times(dn, frames, hx, steps):
    """The same three launches dn_process_frame makes, issued one ABI call each with HIP events in between
    (torch.cuda.Event on the stream the kernels are launched on).  Returns mean ms per kernel."""
    import ctypes as C
    from audio_denoising_amd import _lib
    lib, plan, dev = dn.lib, dn.plan, dn.device
    B = frames.shape[0]
    mel = torch.empty(B, 3, N_MELS, device=dev)
    diff = torch.empty_like(mel)
    peak = torch.empty(B, device=dev)
    out = torch.empty_like(frames)
    model_h = dn.model._native(dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    names = ["stft_mel_log1p", "cell", "synthesis"]
    acc = dict.fromkeys(names, 0.0)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)]
    for s in range(steps):
        e = ev[s]
        e[0].record()
        lib.check(lib.dn_stft_mel_log1p(plan.handle, frames.data_ptr(), mel.data_ptr(), peak.data_ptr(), B,
                                        _lib.DN_PEAK_NORMALIZE | _lib.DN_PRE_WINDOW, st))
        e[1].record()
        lib.check(lib.dn_cell_forward(model_h, mel.data_ptr(), hx.data_ptr(), diff.data_ptr(), hx.data_ptr(), B, 3, N_MELS, N_MELS // 16, st))
        e[2].record()
        lib.check(lib.dn_synthesis(plan.handle, mel.data_ptr(), diff.data_ptr(), None, 7 + s, 0, peak.data_ptr(), out.data_ptr(), B, GL_ITERS, 0.99, st))
        e[3].record()
    torch.cuda.synchronize()
    for s in range(steps):
        for i, n in enumerate(names):
            acc[n] += ev[s][i].elapsed_time(ev[s][i + 1])
    return {n: acc[n] / steps for n in names}


def cpu_baseline(budget_s=20.0):
    """The reference's op sequence on the host CPU (oracle, kind 'port'), batch 256 and batch 1."""
    from oracle import dsp_ref, model_ref, pipeline_ref
    p = pipeline_ref.PARAMS_S
    sd = model_ref.unflatten_weights(np.fromfile(os.path.join(REPO, "tests", "golden", "weights_dari_tult.bin"), dtype=np.float32))
    fb = dsp_ref.melscale_fbanks(p.n_stft, p.n_mels, p.sample_rate)
    # the GPU box gives one-GPU jobs a 16-core share of a much larger host: size the thread pool to the share
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("DN_CPU_THREADS", "16")))
    torch.set_num_threads(cores)
    res = {}
    for B, share in ((BATCH, 0.75), (1, 0.25)):
        g = torch.Generator().manual_seed(1234)
        frames = 0.1 * torch.randn(B, p.n_fft, generator=g)
        hx = torch.zeros(B, 17, 5)
        gen = torch.Generator().manual_seed(4321)
        with torch.no_grad():
            tw = time.perf_counter()
            r = pipeline_ref.process_frame(sd, frames, hx, p, fb, generator=gen)     # warm-up
            hx = r["hx"]
            print(f"[bench] cpu baseline B={B}: warm-up step {time.perf_counter() - tw:.2f} s on {cores} threads", file=sys.stderr, flush=True)
            n, t0 = 0, time.perf_counter()
            while True:
                r = pipeline_ref.process_frame(sd, frames, hx, p, fb, generator=gen)
                hx = r["hx"]
                n += 1
                el = time.perf_counter() - t0
                if el >= budget_s * share or n >= 200:
                    break
        res[B] = (B * n / el, n, el)
    v, n, el = res[BATCH]
    return {"value": round(v, 1), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n} steps of batch {BATCH} ({el:.1f} s) of the same synthetic workload through oracle/pipeline_ref.process_frame "
                      f"(torch-CPU stft/matmul/conv1d/lstsq(gels)/32-iter Griffin-Lim, {cores} threads)",
            "batch1_value": round(res[1][0], 1)}


def side_measurement(args, dn, B, dev):
    """Extra, non-headline measurements (single GPU): other parameter presets and the streaming mode."""
    from audio_denoising_amd.pipeline import HopPipeline, PipelinedStream
    g = torch.Generator().manual_seed(1234)
    out = torch.empty(B, dn.n_fft, device=dev)
    hx = dn.init_hx(B)
    if args.stream:
        ps = PipelinedStream(dn, B)
        hop = (0.1 * torch.randn(B, dn.hop, generator=g)).to(dev)

        def step(i):
            ps.push(hop)
        fin = ps.flush
    elif args.pcie:
        # the boundary hands over HOST buffers: every hop moves its frames up and its result down over PCIe (pinned,
        # async on the same stream; double-buffered so hop n's result copy does not wait for its Griffin-Lim early)
        host_in = (0.1 * torch.randn(B, dn.n_fft, generator=g)).pin_memory()
        host_out = [torch.empty(B, dn.n_fft).pin_memory() for _ in range(2)]
        dframes = [torch.empty(B, dn.n_fft, device=dev) for _ in range(2)]
        douts = [torch.empty(B, dn.n_fft, device=dev) for _ in range(2)]
        pipe = HopPipeline(dn, B)

        def step(i):
            s = i & 1
            dframes[s].copy_(host_in, non_blocking=True)
            pipe.submit(dframes[s], hx, douts[s], seed=1000 + i)          # also completes hop i-1 (its Griffin-Lim blocks)
            host_out[s ^ 1].copy_(douts[s ^ 1], non_blocking=True)
        fin = pipe.flush
    else:
        frames = (0.1 * torch.randn(B, dn.n_fft, generator=g)).to(dev)
        pipe = HopPipeline(dn, B)

        def step(i):
            pipe.submit(frames, hx, out, seed=1000 + i)
        fin = pipe.flush
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.5:
        for i in range(50):
            step(i)
        torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fin()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps({"metric": "frames/s (side measurement, not the headline)", "preset": args.preset, "mode": "stream" if args.stream else ("frames+pcie" if args.pcie else "frames"),
                      "value": round(B * args.steps / el, 1), "unit": "frames/s", "streams": B, "ms_per_step": round(1e3 * el / args.steps, 4),
                      "n_fft": dn.n_fft, "hop": dn.hop, "n_mels": dn.n_mels, "sample_rate": dn.sample_rate,
                      "realtime_streams_per_gpu": int(B * args.steps / el / (dn.sample_rate / dn.hop))}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=BATCH, help="streams per GPU (the metric is quoted at 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial", action="store_true", help="one hop at a time, 3 launches (dn_process_frame) instead of the software-pipelined hop")
    ap.add_argument("--preset", choices=sorted(PRESETS), default="S", help="S = the metric's config; R1/R2 = the reference's own parameters (extra measurements)")
    ap.add_argument("--pcie", action="store_true", help="side measurement: frames arrive in pinned host memory and results return to it every hop")
    ap.add_argument("--stream", action="store_true", help="streaming mode (BASELINE config 5): pipe-owned ring/overlap-add/hx state, one hop of new samples per step")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hop path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    dev = torch.device("cuda", local % n_dev)          # one rank per GPU; (rehearsals on a 1-GPU box share cuda:0)
    torch.cuda.set_device(dev)
    dist = None
    backend = os.environ.get("DN_DIST_BACKEND", "nccl")   # "nccl" = RCCL over xGMI; "gloo" only to rehearse the rank plumbing
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from audio_denoising_amd.shard import shard_range
    dn = build_denoiser(dev, args.preset)
    B = args.batch
    if args.preset != "S" or args.stream or args.pcie:
        return side_measurement(args, dn, B, dev)
    lo, hi = shard_range(B * world, world, rank)          # this rank's global stream ids
    g = torch.Generator().manual_seed(1234 + rank)
    frames = (0.1 * torch.randn(B, N_FFT, generator=g)).to(dev)
    hx = dn.init_hx(B)
    out = torch.empty_like(frames)

    # Product configuration for throughput: software-pipelined hops (dn_pipe_*): ONE launch per hop whose workgroups are
    # hop n's Griffin-Lim next to hop n+1's analysis + model + inverse mel; hx is the only inter-hop dependency.
    from audio_denoising_amd.pipeline import HopPipeline
    pipe = None if args.serial else HopPipeline(dn, B)

    def step(i):
        if pipe is None:
            dn.process_frame_(frames, hx, out, seed=1000 + i, stream_id0=lo)
        else:
            pipe.submit(frames, hx, out, seed=1000 + i, stream_id0=lo)

    def fence():
        if pipe is not None:
            pipe.flush()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier(device_ids=[dev.index]) if backend == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    # untimed: bring the GPU out of its idle power state first (a cold start ran the first few hundred hops at roughly
    # half clock: 124 us/step instead of 68), then the W warm-up steps of the contract
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.5:
        for i in range(50):
            step(i)
        torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out).all()
    if rank == 0:
        print(f"[bench] {args.steps} steps x {B * world} frames in {elapsed:.4f} s", file=sys.stderr, flush=True)

    line = None
    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        value = B * world * args.steps / elapsed
        # the same K steps strictly one after another on one stream (dn_process_frame), for reference
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for i in range(args.steps):
            dn.process_frame_(frames, hx, out, seed=5000 + i, stream_id0=lo)
        torch.cuda.synchronize()
        serial_ms = 1e3 * (time.perf_counter() - ts) / args.steps
        kt = staged_kernel_times(dn, frames, hx, min(args.steps, 100))
        # dominant kernel of the timed region: hop_kernel.  Mean launch duration from HIP events recorded on the launch
        # stream around each of 100 further pipelined hops (each launch = one whole hop of work for the batch).
        if pipe is not None:
            # (one event pair around a run of back-to-back launches: an event between every two launches adds ~4 us each)
            n_ev = min(args.steps, 100)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            pipe.submit(frames, hx, out, seed=1, stream_id0=lo)
            e0.record()
            for i in range(n_ev):
                pipe.submit(frames, hx, out, seed=2 + i, stream_id0=lo)
            e1.record()
            pipe.flush()
            torch.cuda.synchronize()
            dom_ms = e0.elapsed_time(e1) / n_ev
            dom_name, dom_flop = "hop_kernel (hop n Griffin-Lim blocks + hop n+1 analysis/model/inverse-mel blocks)", TOTAL_FLOP_PER_FRAME
            dom_note = ("fp32 compute roof (FFT butterflies on the fp32 VALU, convs on fp32 MFMA; vector and matrix fp32 peaks are both 157.3 TF); "
                        "algorithmic = 6.50 MFLOP per frame (198 rFFT-1024 x 25,600 + mel + convs + inverse mel) x 256 frames per launch")
            dom_bytes = HBM_BYTES_PER_FRAME
        else:
            dom_ms, dom_name, dom_flop = kt["synthesis"], "griffinlim_kernel<from mel> (P8-P12)", GL_FLOP_PER_FRAME
            dom_note = "fp32 compute roof; algorithmic = (195 rFFT-1024 x 25,600 + 246,240 inverse-mel) flop per frame x 256 frames per launch"
            dom_bytes = GL_HBM_BYTES_PER_FRAME
        gl_s = dom_ms * 1e-3
        ach = dom_flop * B / gl_s / 1e12
        traffic = None
        pmc = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            traffic = json.load(open(pmc)).get("hop_kernel_hbm_bytes_per_launch" if pipe is not None else "griffinlim_kernel_hbm_bytes_per_launch")
        line = {
            "metric": "denoised audio frames/sec (32 ms, 16 kHz, hop 512) at batch 256; 1/2/4/8 GPU",
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: batch 256 synthetic 32 ms frames per GPU, n_fft=1024 hop=512 n_mels=80, GRUUNet2 fp32 "
                                   "(dari_tult weights, num_compressed_bins=5), 32-iter Griffin-Lim, device-RNG initial phases, hx carried",
                       "streams_per_gpu": B, "frames_per_step": B * world, "sample_rate": SR, "n_fft": N_FFT, "hop": HOP,
                       "n_mels": N_MELS, "griffinlim_iters": GL_ITERS, "parallelism": f"stream-sharded x{world} (no data-path collective)"},
            "roofline": {"bound": "mfma", "kernel": dom_name, "achieved": round(ach, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / PEAK_FP32_TFLOPS, 4), "traffic": traffic, "note": dom_note,
                         "launch_ms": round(dom_ms, 4),
                         "hbm_frac": round(dom_bytes * B / gl_s / 1e9 / PEAK_HBM_GBS, 6)},
            "kernel_ms": {k: round(v, 4) for k, v in kt.items()},
            # MFMA utilisation of the UNet convs: conv FLOPs / (serial cell_kernel time x fp32 MFMA peak)
            "conv_mfma": {"kernel": "cell_kernel (fp32 v_mfma_f32_16x16x4_f32)", "achieved": round(939150 * B / (kt["cell"] * 1e-3) / 1e12, 3),
                          "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": round(939150 * B / (kt["cell"] * 1e-3) / 1e12 / PEAK_FP32_TFLOPS, 4),
                          "note": "0.24 GFLOP per batch-256 launch is 1.5 us at peak: structurally latency-bound (SURVEY section 7)"},
            "schedule": "serial, 3 launches per hop (dn_process_frame)" if pipe is None else "software-pipelined, 1 launch per hop (dn_pipe_submit)",
            "serial_ms_per_step": round(serial_ms, 4),
            "whole_path": {"tflops": round(TOTAL_FLOP_PER_FRAME * value / 1e12, 3),
                           "fp32_frac": round(TOTAL_FLOP_PER_FRAME * value / 1e12 / (PEAK_FP32_TFLOPS * world), 4),
                           "hbm_frac": round(HBM_BYTES_PER_FRAME * value / 1e9 / (PEAK_HBM_GBS * world), 6)},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
    if dist is not None:
        dist.barrier(device_ids=[dev.index]) if backend == "nccl" else dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
