#!/usr/bin/env python3
"""Headline benchmark: denoised 32 ms frames/s (16 kHz, n_fft 1024, hop 512, 80 mels) at batch 256 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (a torchrun child
process, started before this process touches the GPU) and fails loudly when the node has fewer than N GPUs.

A "step" is one pass of the whole per-hop path (P1..P12 of SURVEY.md section 8a: peak-normalise, Hann,
3-column STFT, mel, log1p, GRUUNet2 x3 steps, residual, expm1, inverse mel, 32-iteration Griffin-Lim,
`* peak`) over one batch of 256 synthetic frames per GPU, inputs resident in HBM, hidden state carried
from step to step.  Streams are independent, so N GPUs run N x 256 streams with no data-path collective
(weak scaling); the only collectives of the headline are the barriers and the MAX of the elapsed time.  With N > 1 the
same run also times the ingress-inclusive variant SURVEY.md section 8e asks for: rank 0 holds all N x 256 frames, every
step scatters them to the ranks and gathers the denoised frames back (RCCL grouped send/recv over xGMI, issued on a
second HIP stream and double-buffered so hop i+1's ingress and hop i-1's egress overlap hop i's kernels); it is
reported in `ingress_variant`, never as `value`.

Besides the contract fields the JSON line carries
  roofline     -- the dominant kernel (hop_kernel: one launch = one hop of the batch): algorithmic FLOPs per launch /
                  its average launch duration measured with HIP events in this process, against the fp32 compute peak;
  cpu_baseline -- the CPU oracle (the reference's op sequence restated on torch-CPU, oracle/pipeline_ref.py)
                  timed on this host's cores on a bounded sample of the same workload (rank 0, N = 1 only): all the
                  cores the process may use, and one thread.
"""
import argparse
import json
import os
import subprocess
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
CONV_FLOP_PER_FRAME = 939150                                      # 2 x 469,575 MACs: the MFMA-eligible UNet convs
TOTAL_FLOP_PER_FRAME = 198 * FLOP_PER_RFFT + 2 * 246240 + CONV_FLOP_PER_FRAME  # 6.50 MFLOP
HBM_BYTES_PER_FRAME = 4 * N_FFT + 4 * N_FFT + 2 * 4 * 17 * 5       # 8,872 B compulsory
PEAK_FP32_TFLOPS = 157.3                                          # MI355X_MICROARCH.md: vector == matrix fp32 peak
PEAK_BF16_TFLOPS = 2500.0                                         # dense bf16 MFMA
PEAK_HBM_GBS = 8000.0
METRIC = "denoised audio frames/sec (32 ms, 16 kHz, hop 512) at batch 256; 1/2/4/8 GPU"

# parameter presets: S = BASELINE.json's synthetic config (the metric); R1 = the reference app's own (app3.py:13-33);
# R2 = server.py:166-170.  Only S is the headline; the others are reported by `--preset` for DESIGN.md.
PRESETS = {"S": (16000, 1024, 512, 80, "dari_tult"), "R1": (48000, 1536, 768, 64, "dari_tult2"), "R2": (48000, 1024, 512, 64, "dari_tult")}


def build_denoiser(dev, preset="S", conv="fp32"):
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
    model.conv_precision = conv
    return Denoiser(model, sr, n_fft, hop, n_mels, n_iter=GL_ITERS)


def staged_kernel_times(dn, frames, hx, steps):
    """The stages of the hop as separate launches (dn_stft_mel_log1p, dn_cell_forward[_bf16], dn_synthesis).  Each kernel is timed by ONE
    HIP event pair (torch.cuda.Event on the stream the kernels are launched on) around `steps` back-to-back launches of it -- an event between
    every two launches adds several microseconds to a 6-12 us kernel.  Returns mean ms per launch."""
    import ctypes as C
    from audio_denoising_amd import _lib
    lib, plan, dev = dn.lib, dn.plan, dn.device
    B = frames.shape[0]
    n_mels = dn.n_mels
    mel = torch.empty(B, 3, n_mels, device=dev)
    diff = torch.empty_like(mel)
    peak = torch.empty(B, device=dev)
    out = torch.empty_like(frames)
    hx = hx.clone()
    model_h = dn.model._native(dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    cell = lib.dn_cell_forward_bf16 if dn.model.conv_precision == "bf16" else lib.dn_cell_forward
    calls = {
        "stft_mel_log1p": lambda s: lib.dn_stft_mel_log1p(plan.handle, frames.data_ptr(), mel.data_ptr(), peak.data_ptr(), B,
                                                         _lib.DN_PEAK_NORMALIZE | _lib.DN_PRE_WINDOW, st),
        "cell": lambda s: cell(model_h, mel.data_ptr(), hx.data_ptr(), diff.data_ptr(), hx.data_ptr(), B, 3, n_mels, n_mels // 16, st),
        "synthesis": lambda s: lib.dn_synthesis(plan.handle, mel.data_ptr(), diff.data_ptr(), None, 7 + s, 0, peak.data_ptr(), out.data_ptr(), B,
                                                dn.n_iter, 0.99, st),
    }
    res = {}
    for name, call in calls.items():
        for s_ in range(max(3, steps // 2)):            # (warm: also keeps the clocks up between the measurements)
            lib.check(call(s_))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s_ in range(steps):
            lib.check(call(s_))
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / steps
    return res


def default_depth(batch, n_fft):
    """Hops of one stream in flight (dn_pipe_set_depth) for throughput: enough that every CU of an MI355X (256) holds about four chains
    (measured, tools/pipe_time.py: 384 streams 4; 512 / 768 / 896 streams 6.1 / 6.3 / 6.4 M frames/s at depth 2 against 5.0 / 5.1 / 5.9 M at
    depth 1; from 1,024 streams on depth 1 is the fastest)."""
    if n_fft != 1024:
        return 1
    return 4 if batch <= 384 else 2 if batch < 1024 else 1


def default_group(batch, n_fft):
    """Hops per launch (dn_pipe_set_group) for throughput: with about one stream per CU (up to 384 streams) a launch carries four consecutive hops
    of every stream and the WHOLE Griffin-Lim chains of the previous four -- the occupancy of the depth-4 pipe without its parked chains."""
    if n_fft != 1024:
        return 0
    # (measured, profiles/r04_group_sweep.txt: 384 streams 5.88 M frames/s against 5.32 M at depth 4; 512 / 768 streams as groups of two, two streams a
    # workgroup: 6.52 / 6.52 M against 6.13 / 6.34 M at depth 2; from 1,024 streams the one-hop pipe with a wavefront per stream is faster: 6.73 against 6.42 M)
    return 4 if batch <= 384 else 2 if batch < 1024 else 0


def prewarm(step, seconds=0.3):
    """Untimed: bring the GPU out of its idle power state (a cold GPU runs the first hundreds of hops at roughly half clock)."""
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < seconds:
        for _ in range(50):
            step()
        torch.cuda.synchronize()


def time_pipe(dn, B, dev, steps, depth=1, seconds=None):
    """frames/s of the software-pipelined hop at batch B and the given depth (one event pair around the run, flush included)."""
    from audio_denoising_amd.pipeline import HopPipeline
    g = torch.Generator().manual_seed(4321)
    frames = (0.1 * torch.randn(B, dn.n_fft, generator=g)).to(dev)
    hx = dn.init_hx(B)
    out = torch.empty_like(frames)
    pipe = HopPipeline(dn, B)
    pipe.set_depth(depth)
    prewarm(lambda: pipe.submit(frames, hx, out, seed=1, check_weights=False), 0.3 if B <= 1024 else 0.1)
    pipe.flush()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.submit(frames, hx, out, seed=1, check_weights=False)
    pipe.flush()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return B * steps / el, 1e3 * el / steps


def time_group(dn, B, dev, steps, group):
    """frames/s of hop groups at batch B (dn_pipe_submit_group: `group` hops per launch, whole Griffin-Lim chains): steady state + flush."""
    from audio_denoising_amd.pipeline import HopPipeline
    g = torch.Generator().manual_seed(4321)
    frames = (0.1 * torch.randn(group, B, dn.n_fft, generator=g)).to(dev)
    hx = dn.init_hx(B)
    out = torch.empty_like(frames)
    pipe = HopPipeline(dn, B)
    pipe.set_group(group)
    prewarm(lambda: pipe.submit_group(frames, hx, out, seed=1, check_weights=False), 0.3)
    pipe.flush()
    torch.cuda.synchronize()
    n = max(1, steps // group)
    t0 = time.perf_counter()
    for _ in range(n):
        pipe.submit_group(frames, hx, out, seed=1, check_weights=False)
    pipe.flush()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return B * n * group / el, 1e3 * el / (n * group)


def time_queued(dn, B, dev, steps, queues=2, depth=2, pipes=None):
    """frames/s of B streams as `queues` pipes on as many HIP streams, split hops (pipeline.QueuedHopPipelines): steady state + flush."""
    from audio_denoising_amd.pipeline import QueuedHopPipelines
    g = torch.Generator().manual_seed(4321)
    frames = (0.1 * torch.randn(B, dn.n_fft, generator=g)).to(dev)
    hx = dn.init_hx(B)
    out = torch.empty_like(frames)
    torch.cuda.synchronize()
    qp = QueuedHopPipelines(dn, B, queues=queues, depth=depth, pipes=pipes)

    def step():
        qp.submit(frames, hx, out, seed=1, check_weights=False)
    prewarm(step, 0.3 if B <= 1024 else 0.1)
    torch.cuda.synchronize()
    # steady state, as the headline: the pipes primed on both sides of the timed region, the drain outside it (a run that STARTS from a flush starts
    # its queues together; QueuedHopPipelines staggers them itself since round 4 -- lockstep measured 12 % slower at 8,192 streams)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    qp.flush()
    torch.cuda.synchronize()
    return B * steps / el, 1e3 * el / steps


def extra_measurements(args, dev, budget_steps=1500):
    """Bounded side numbers carried in the same JSON line (about a second each): BASELINE config 3 (bf16 MFMA conv tiles), config 5's per-GPU
    share (1,024 streams, ONE hipGraph-captured push replayed per hop), the reference app's own parameters (n_fft 1536), the saturated regime,
    and the latency of ONE stream (batch 1: how the reference app runs, app3.py:178) through the unpipelined and the pipelined streaming step."""
    from audio_denoising_amd.pipeline import DenoiserStream, PipelinedStream
    res = {}
    # config 3
    dnb = build_denoiser(dev, "S", "bf16")
    v, ms = time_group(dnb, BATCH, dev, 1000, default_group(BATCH, N_FFT))          # the headline's schedule (hop groups of four), bf16 conv tiles
    g = torch.Generator().manual_seed(1)
    fr = (0.1 * torch.randn(BATCH, N_FFT, generator=g)).to(dev)
    kt = staged_kernel_times(dnb, fr, dnb.init_hx(BATCH), 300)
    conv = CONV_FLOP_PER_FRAME * BATCH / (kt["cell"] * 1e-3) / 1e12
    res["config3_bf16"] = {"value": round(v, 1), "unit": "frames/s", "ms_per_step": round(ms, 4), "streams": BATCH,
                           "hops_per_launch": default_group(BATCH, N_FFT), "cell_ms": round(kt["cell"], 4),
                           "conv_mfma": {"achieved": round(conv, 3), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(conv / PEAK_BF16_TFLOPS, 5)}}
    # config 5, one GPU's share: 1,024 streams, streaming state on the device, one captured push per hop
    dn = build_denoiser(dev, "S", "fp32")
    B5 = 1024
    ps = PipelinedStream(dn, B5)
    hop = (0.1 * torch.randn(B5, dn.hop, generator=g)).to(dev)
    hop_out = torch.empty_like(hop)
    graph = ps.graph_step(hop, hop_out)
    prewarm(graph.replay)
    n5 = 400
    t0 = time.perf_counter()
    for _ in range(n5):
        graph.replay()
    ps.flush()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    res["config5_stream_graph_1024"] = {"value": round(B5 * n5 / el, 1), "unit": "frames/s", "ms_per_step": round(1e3 * el / n5, 4), "streams": B5,
                                        "mode": "dn_pipe_stream_push captured once in a hipGraph, replayed per hop; ring/overlap-add/hx resident"}
    # the same with eight consecutive pushes captured as ONE graph (a graph launch leaves ~5 us of idle GPU behind it; eager launches do not)
    K8 = 8
    hops8 = (0.1 * torch.randn(K8, B5, dn.hop, generator=g)).to(dev)
    outs8 = torch.empty_like(hops8)
    graph8 = ps.graph_step(hops8, outs8)
    prewarm(graph8.replay)
    t0 = time.perf_counter()
    for _ in range(n5 // K8):
        graph8.replay()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ps.flush()
    res["config5_stream_graph_1024"]["graph_of_8_hops"] = {"value": round(B5 * (n5 // K8) * K8 / el, 1), "unit": "frames/s",
                                                           "ms_per_step": round(1e3 * el / ((n5 // K8) * K8), 4)}
    # ... and as two independent streaming pipes of 512 on two HIP streams (depth 2, split hops), one captured push per pipe and hop
    from audio_denoising_amd.pipeline import QueuedPipelinedStreams
    torch.cuda.synchronize()
    qs = QueuedPipelinedStreams(dn, B5, queues=2, depth=2)
    replay = qs.graph_steps(hop, hop_out)
    prewarm(replay)
    t0 = time.perf_counter()
    for _ in range(n5):
        replay()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    qs.flush()
    res["config5_stream_graph_1024"]["two_queues_depth2"] = {"value": round(B5 * n5 / el, 1), "unit": "frames/s", "ms_per_step": round(1e3 * el / n5, 4)}
    # ... and eight pushes per captured graph and pipe
    qs8 = QueuedPipelinedStreams(dn, B5, queues=2, depth=2)
    replay8 = qs8.graph_steps([hops8[k] for k in range(K8)], [outs8[k] for k in range(K8)])
    prewarm(replay8)
    t0 = time.perf_counter()
    for _ in range(n5 // K8):
        replay8()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    qs8.flush()
    res["config5_stream_graph_1024"]["two_queues_depth2"]["graph_of_8_hops"] = {"value": round(B5 * (n5 // K8) * K8 / el, 1), "unit": "frames/s",
                                                                                "ms_per_step": round(1e3 * el / ((n5 // K8) * K8), 4)}
    # between one stream per CU and saturation: 512 streams as groups of two hops (two streams a chain workgroup) against the depth-2 pipe
    v, ms = time_group(dn, 512, dev, 600, default_group(512, N_FFT))
    v2, ms2 = time_pipe(dn, 512, dev, 600, 2)
    res["batch_512"] = {"value": round(v, 1), "unit": "frames/s", "ms_per_step": round(ms, 4), "hops_per_launch": default_group(512, N_FFT),
                        "fp32_frac": round(TOTAL_FLOP_PER_FRAME * v / 1e12 / PEAK_FP32_TFLOPS, 4), "depth2_value": round(v2, 1), "depth2_ms_per_step": round(ms2, 4)}
    # saturated regime (wavefront-per-stream Griffin-Lim)
    for b, n in ((1024, 300), (8192, 40)):
        v, ms = time_pipe(dn, b, dev, n, 1)
        res[f"batch_{b}"] = {"value": round(v, 1), "unit": "frames/s", "ms_per_step": round(ms, 4),
                             "fp32_frac": round(TOTAL_FLOP_PER_FRAME * v / 1e12 / PEAK_FP32_TFLOPS, 4)}
    # the same streams as independent split-hop pipes taking turns on two HIP streams (pipeline.throughput_plan: 1,024 streams as two pipes of 512
    # at depth 2, 8,192 as eight pipes of 1,024 at depth 1): one pipe's chains run beside another's front halves
    from audio_denoising_amd.pipeline import throughput_plan
    for b, n in ((1024, 300), (8192, 40)):
        plan = throughput_plan(b, N_FFT)
        v, ms = time_queued(dn, b, dev, n, queues=plan["queues"], depth=plan["depth"], pipes=plan["pipes"])
        res[f"batch_{b}"]["queued"] = {"value": round(v, 1), "unit": "frames/s", "ms_per_step": round(ms, 4), "plan": plan,
                                       "fp32_frac": round(TOTAL_FLOP_PER_FRAME * v / 1e12 / PEAK_FP32_TFLOPS, 4)}
    # the reference app's own STFT parameters (app3.py:29-33): 48 kHz, n_fft 1536, 64 mels
    dnr = build_denoiser(dev, "R1", "fp32")
    v, ms = time_pipe(dnr, BATCH, dev, 600, 1)
    res["r1_app_params"] = {"value": round(v, 1), "unit": "frames/s", "ms_per_step": round(ms, 4), "streams": BATCH, "n_fft": 1536, "n_mels": 64,
                            "sample_rate": 48000, "fp32_frac": round(9.39e6 * v / 1e12 / PEAK_FP32_TFLOPS, 4)}
    # one stream: latency of a hop (synchronised after every hop), beside cpu_baseline.batch1_value
    lat = {}
    ds = DenoiserStream(dn, 1)
    chunk = (0.1 * torch.randn(1, dn.hop, generator=g)).to(dev)
    ds.push(torch.cat([chunk, chunk], 1))
    torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        t0 = time.perf_counter()
        ds.push(chunk)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    lat["dn_stream_step"] = round(1e6 * float(np.median(ts)), 1)
    p1 = PipelinedStream(dn, 1)
    o1 = torch.empty(1, dn.hop, device=dev)
    for _ in range(5):
        p1.push_(chunk, o1, check_weights=False)
    torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        t0 = time.perf_counter()
        p1.push_(chunk, o1, check_weights=False)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    lat["dn_pipe_stream_push"] = round(1e6 * float(np.median(ts)), 1)
    lat["note"] = ("median wall time of one hop of ONE stream, host call to synchronised result (the pipelined push returns the previous hop's "
                   "samples); a 16 kHz / hop 512 stream delivers a hop every 32,000 us")
    res["latency_us_b1"] = lat
    return res


def usable_cores():
    """Cores this process may really use: the scheduler affinity, capped by the cgroup CPU quota when there is one
    (a GPU box hands a one-GPU job a share of a much larger host; threads beyond the quota only thrash)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(budget_s=26.0):
    """The reference's op sequence on the host CPU (oracle, kind 'port') on a bounded sample of the same synthetic workload: batch 256
    with every core this process may use, with 4 threads and with ONE thread (torch's intra-op pool does not always scale on the
    small tensors of this path: on the 16-core share of a GPU box one thread beat sixteen), and batch 1 (how the app really
    runs).  `value` is the best of the batch-256 runs, `cores` the threads it used; all three are reported."""
    from oracle import dsp_ref, model_ref, pipeline_ref
    p = pipeline_ref.PARAMS_S
    sd = model_ref.unflatten_weights(np.fromfile(os.path.join(REPO, "tests", "golden", "weights_dari_tult.bin"), dtype=np.float32))
    fb = dsp_ref.melscale_fbanks(p.n_stft, p.n_mels, p.sample_rate)
    affinity = len(os.sched_getaffinity(0))
    cores = int(os.environ.get("DN_CPU_THREADS", str(usable_cores())))  # every core this process may use (affinity, cgroup quota)
    runs = [(f"{t}t", BATCH, t, 0.28) for t in sorted({cores, min(cores, 4), 1}, reverse=True)] + [("b1", 1, cores, 0.12)]
    res = {}
    for tag, B, threads, share in runs:
        torch.set_num_threads(threads)
        g = torch.Generator().manual_seed(1234)
        frames = 0.1 * torch.randn(B, p.n_fft, generator=g)
        hx = torch.zeros(B, 17, 5)
        gen = torch.Generator().manual_seed(4321)
        with torch.no_grad():
            tw = time.perf_counter()
            r = pipeline_ref.process_frame(sd, frames, hx, p, fb, generator=gen)     # warm-up
            hx = r["hx"]
            print(f"[bench] cpu baseline {tag}: B={B}, {threads} threads, warm-up {time.perf_counter() - tw:.2f} s", file=sys.stderr, flush=True)
            n, t0 = 0, time.perf_counter()
            while True:
                r = pipeline_ref.process_frame(sd, frames, hx, p, fb, generator=gen)
                hx = r["hx"]
                n += 1
                el = time.perf_counter() - t0
                if el >= budget_s * share or n >= 200:
                    break
        res[tag] = (B * n / el, n, el, threads)
    torch.set_num_threads(cores)
    best = max((k for k in res if k != "b1"), key=lambda k: res[k][0])
    v, n, el, th = res[best]
    return {"value": round(v, 1), "unit": "frames/s", "cores": th, "kind": "port",
            "sample": f"{n} steps of batch {BATCH} ({el:.1f} s) of the same synthetic workload through oracle/pipeline_ref.process_frame "
                      f"(torch-CPU stft/matmul/conv1d/lstsq(gels)/32-iter Griffin-Lim, {th} thread(s): the fastest of the thread counts tried)",
            "host": {"sched_getaffinity": affinity, "os_cpu_count": os.cpu_count(), "usable_cores": usable_cores()},
            "by_threads": {str(res[k][3]): round(res[k][0], 1) for k in res if k != "b1"},
            "one_thread_value": round(res["1t"][0], 1), "all_cores_value": round(res[f"{cores}t"][0], 1), "all_cores": cores,
            "batch1_value": round(res["b1"][0], 1)}


def side_measurement(args, dn, B, dev):
    """Extra, non-headline measurements (single GPU): other parameter presets and the streaming mode."""
    from audio_denoising_amd.pipeline import HopPipeline, PipelinedStream
    g = torch.Generator().manual_seed(1234)
    out = torch.empty(B, dn.n_fft, device=dev)
    hx = dn.init_hx(B)
    mode = "frames"
    if args.stream and args.pcie:
        # the host-buffer transport of the streaming form (dn_pipe_stream_push_host): int16 hops in page-locked host memory, upload / hop / download
        # on three queues, double-buffered -- beside the same stream fed from device memory
        from audio_denoising_amd.pipeline import HostFedStream
        staged = os.environ.get("DN_HOST_STAGED") == "1"
        direct = os.environ.get("DN_HOST_DIRECT") == "1"
        hs = HostFedStream(dn, B, s16=True, depth=args.depth if args.depth > 0 else 1, staged=staged, defer=not direct)
        hop_h = (0.1 * torch.randn(B, dn.hop, generator=g) * 32767.0).to(torch.int16)
        how = "staged copies on two queues" if staged else "zero copy, output stored straight to host memory" if direct else \
            "zero copy, output carried out by the next launch"
        mode = f"stream+pcie (int16 hops in pinned host memory, {how}), depth {hs.depth}"

        def step(i):
            hs.push(hop_h, copy=False)
        fin = hs.drain
    elif args.stream and args.group > 0:
        # streaming hop groups (dn_pipe_stream_push_group): `group` hops in, `group` hops out per launch, whole Griffin-Lim chains
        ps = PipelinedStream(dn, B)
        ps.set_group(args.group)
        hops = (0.1 * torch.randn(args.group, B, dn.hop, generator=g)).to(dev)
        hops_out = torch.empty_like(hops)
        mode = f"stream, groups of {args.group} hops per launch"
        args.steps = max(1, args.steps // args.group) * args.group
        args.warmup = -(-args.warmup // args.group) * args.group
        if args.graph:
            ps._bind()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                ps.push_group_(hops, hops_out, check_weights=False)
            mode += " + hipGraph (one captured group push replayed)"

        def step(i):
            if i % args.group == 0:
                graph.replay() if args.graph else ps.push_group_(hops, hops_out, check_weights=False)
        fin = ps.flush_group
    elif args.stream:
        ps = PipelinedStream(dn, B)
        if args.depth > 0:
            ps.set_depth(args.depth)
        hop = (0.1 * torch.randn(B, dn.hop, generator=g)).to(dev)
        hop_out = torch.empty_like(hop)
        mode = "stream"
        if args.graph:
            # BASELINE config 5: ONE hipGraph-captured push replayed per hop (slot parity / seed / pending live on the device)
            gk = max(1, args.graph_hops)
            if gk > 1:
                hop = (0.1 * torch.randn(gk, B, dn.hop, generator=g)).to(dev)
                hop_out = torch.empty_like(hop)
            graph = ps.graph_step(hop, hop_out)
            mode = "stream+hipGraph" + (f" ({gk} pushes per graph)" if gk > 1 else "")
            args.steps = max(1, args.steps // gk) * gk

            def step(i):
                if i % gk == 0:
                    graph.replay()
        else:
            def step(i):
                ps.push_(hop, hop_out, check_weights=False)
        fin = ps.flush
    elif args.pcie:
        # the boundary hands over HOST buffers: every hop moves its frames up and its result down over PCIe (pinned,
        # async on the same stream; double-buffered so hop n's result copy does not wait for its Griffin-Lim early)
        host_in = (0.1 * torch.randn(B, dn.n_fft, generator=g)).pin_memory()
        host_out = [torch.empty(B, dn.n_fft).pin_memory() for _ in range(2)]
        dframes = [torch.empty(B, dn.n_fft, device=dev) for _ in range(2)]
        douts = [torch.empty(B, dn.n_fft, device=dev) for _ in range(2)]
        pipe = HopPipeline(dn, B)
        mode = "frames+pcie"

        def step(i):
            s = i & 1
            dframes[s].copy_(host_in, non_blocking=True)
            pipe.submit(dframes[s], hx, douts[s], seed=1000)              # also completes hop i-1 (its Griffin-Lim blocks)
            host_out[s ^ 1].copy_(douts[s ^ 1], non_blocking=True)
        fin = pipe.flush
    elif args.queues > 1:
        from audio_denoising_amd.pipeline import QueuedHopPipelines
        frames = (0.1 * torch.randn(B, dn.n_fft, generator=g)).to(dev)
        torch.cuda.synchronize()
        pipe = QueuedHopPipelines(dn, B, queues=args.queues, depth=args.depth if args.depth > 0 else 2, pipes=args.pipes if args.pipes > 0 else None)
        mode = f"frames, {len(pipe.pipes)} pipes on {args.queues} HIP streams, split hops, depth {pipe.depth}"

        def step(i):
            pipe.submit(frames, hx, out, seed=1000, check_weights=False)
        fin = pipe.flush
    elif args.group > 0 or (args.group < 0 and args.depth <= 0 and default_group(B, dn.n_fft) > 0):
        G = args.group if args.group > 0 else default_group(B, dn.n_fft)
        frames = (0.1 * torch.randn(G, B, dn.n_fft, generator=g)).to(dev)
        outs = torch.empty_like(frames)
        pipe = HopPipeline(dn, B)
        pipe.set_group(G)
        mode = f"frames, groups of {G} hops per launch"
        args.steps = max(1, args.steps // G) * G
        args.warmup = -(-args.warmup // G) * G

        def step(i):
            if i % G == 0:
                pipe.submit_group(frames, hx, outs, seed=1000, check_weights=False)
        fin = pipe.flush
    else:
        frames = (0.1 * torch.randn(B, dn.n_fft, generator=g)).to(dev)
        pipe = HopPipeline(dn, B)
        pipe.set_depth(args.depth if args.depth > 0 else default_depth(B, dn.n_fft))
        mode = f"frames, depth {pipe.depth}"

        def step(i):
            pipe.submit(frames, hx, out, seed=1000, check_weights=False)
        fin = pipe.flush
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.5:
        for i in range(50):
            step(i)
        torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    # steady state, as the headline: K launches between synchronize brackets with the pipe primed on both sides (every launch carries one hop of
    # work for every stream); the drain (flush of the hops in flight, first-use allocations and the pageable copy of its result included) is
    # timed separately -- at K = 2,000 it used to add 6 us per step to the host-fed line
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    fin()
    torch.cuda.synchronize()
    drain = time.perf_counter() - t0 - el
    print(json.dumps({"metric": "frames/s (side measurement, not the headline)", "preset": args.preset, "mode": mode, "conv": args.conv,
                      "value": round(B * args.steps / el, 1), "unit": "frames/s", "streams": B, "ms_per_step": round(1e3 * el / args.steps, 4),
                      "drain_ms": round(1e3 * drain, 3), "timed_region": "steady state (pipe primed on both sides); drain timed separately",
                      "n_fft": dn.n_fft, "hop": dn.hop, "n_mels": dn.n_mels, "sample_rate": dn.sample_rate,
                      "realtime_streams_per_gpu": int(B * args.steps / el / (dn.sample_rate / dn.hop))}), flush=True)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks as a child torchrun job.  Nothing in this process has
    initialised the GPU yet (device_count() does not), and the ranks are fresh processes -- never an exec of this one."""
    backend = os.environ.get("DN_DIST_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    rehearsal = os.environ.get("DN_BENCH_REHEARSAL") == "1" and backend == "gloo"
    if n_dev < args.gpus and not rehearsal and not (backend == "gloo" and os.environ.get("DN_ALLOW_SHARED_GPU") == "1"):
        raise SystemExit(f"bench.py --gpus {args.gpus}: this node exposes {n_dev} GPU(s); one rank per GPU is required "
                         f"(a rehearsal of the rank plumbing on fewer GPUs needs DN_DIST_BACKEND=gloo DN_ALLOW_SHARED_GPU=1)")
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {args.gpus} without a launcher: starting {args.gpus} ranks ({' '.join(cmd[1:8])} ...)", file=sys.stderr, flush=True)
    raise SystemExit(subprocess.run(cmd).returncode)


class Rehearsal:
    """Stand-in for the hop when rehearsing the RANK PLUMBING on CPU-only ranks (DN_BENCH_REHEARSAL=1, gloo; tests/test_bench_ranks.py):
    `out = 2 * frames`, nothing else.  It exercises self-launch, sharding, barriers, the MAX reduction, the double-buffered
    scatter/gather loop and the JSON fields -- never a kernel, and its line says so (`data: rehearsal`)."""

    class _P:
        def __init__(self): self.done = []
        def submit(self, frames, hx, out, **kw): out.copy_(frames * 2.0)
        def submit_group(self, frames, hx, out, **kw): out.copy_(frames * 2.0)
        def set_group(self, hops): pass
        def flush(self): pass

    def __init__(self): self.device, self.n_fft = torch.device("cpu"), N_FFT
    def init_hx(self, B): return torch.zeros(B, 17, 5)
    def pipe(self, B): return Rehearsal._P()
    def process_frame_(self, frames, hx, out, **kw): out.copy_(frames * 2.0)


def ingress_variant(dn, pipe, B, world, rank, dev, lo, steps, warmup, dist, backend, fence, on_gpu, group=0):
    """Every step: scatter the N x 256 frames rank 0 holds -> hop -> gather the denoised frames on rank 0, on the HEADLINE's schedule:
    the unit is one launch = `group` consecutive hops of every stream (`group` = 0: one hop per launch, the one-hop pipe).  Transfers are issued on a
    second HIP stream, double-buffered: while launch i computes, the frames of launch i+1 arrive and the result of launch i-1 leaves (a launch
    completes the Griffin-Lim of the hops the previous launch fronted)."""
    from audio_denoising_amd.shard import gather_rows, scatter_rows
    total = B * world
    G = max(group, 1)
    cdev = dev if on_gpu else torch.device("cpu")
    g = torch.Generator().manual_seed(99)
    big_in = (0.1 * torch.randn(G, total, N_FFT, generator=g)).to(cdev) if rank == 0 else None
    big_out = torch.zeros(G, total, N_FFT, device=cdev) if rank == 0 else None
    in_buf = [torch.empty(G, B, N_FFT, device=cdev) for _ in range(2)]
    out_buf = [torch.zeros(G, B, N_FFT, device=cdev) for _ in range(2)]
    hx = dn.init_hx(B)
    if on_gpu:
        comm = torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream(dev)
        ev_in = [torch.cuda.Event() for _ in range(2)]
        ev_hop = [torch.cuda.Event() for _ in range(2)]
        ev_gth = [torch.cuda.Event() for _ in range(2)]

    def scatter(i):
        for h in range(G):
            scatter_rows(None if big_in is None else big_in[h], total, (N_FFT,), torch.float32, cdev, out=in_buf[i & 1][h])

    def gather(i):
        for h in range(G):
            gather_rows(out_buf[i & 1][h], total, out=None if big_out is None else big_out[h])

    def launch(s):
        if group > 0:
            pipe.submit_group(in_buf[s], hx, out_buf[s], seed=3000, stream_id0=lo, check_weights=False)
        else:
            pipe.submit(in_buf[s][0], hx, out_buf[s][0], seed=3000, stream_id0=lo, check_weights=False)

    def run(n_hops):
        n = -(-n_hops // G)                       # launches
        if on_gpu:
            with torch.cuda.stream(comm):
                scatter(0)
                ev_in[0].record(comm)
            for i in range(n):
                s = i & 1
                cur.wait_event(ev_in[s])
                if i >= 3:
                    cur.wait_event(ev_gth[s ^ 1])                    # launch i rewrites out_buf[(i-1)&1]: the egress of launch i-3 has left it
                launch(s)                                            # launch i: front halves of unit i, Griffin-Lim of unit i-1
                ev_hop[s].record(cur)
                with torch.cuda.stream(comm):
                    if i + 1 < n:
                        if i >= 1:
                            comm.wait_event(ev_hop[s ^ 1])          # launch i-1 has read in_buf[(i+1)&1] (its front halves are done)
                        scatter(i + 1)                               # overlaps launch i
                        ev_in[s ^ 1].record(comm)
                    if i >= 1:
                        comm.wait_event(ev_hop[s])                   # launch i completed unit i-1 into out_buf[(i-1)&1]
                        gather(i - 1)                                # overlaps launch i+1
                        ev_gth[s ^ 1].record(comm)
            pipe.flush()
            ev_hop[0].record(cur)
            with torch.cuda.stream(comm):
                comm.wait_event(ev_hop[0])
                gather(n - 1)
            cur.wait_stream(comm)
        else:
            for i in range(n):
                scatter(i)
                launch(i & 1)
                gather(i)
        return n * G
    run(max(2 * G, warmup))
    fence()
    t0 = time.perf_counter()
    done = run(steps)
    fence()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    ok = True
    if rank == 0:
        ok = bool(torch.isfinite(big_out).all()) and float(big_out.abs().max()) > 0.0
    return {"ingress": "scatter_gather", "value": round(total * done / el, 1), "unit": "frames/s", "ms_per_step": round(1e3 * el / done, 4),
            "hops_per_launch": G, "schedule": "the headline's: hop groups" if group > 0 else "one hop per launch (one-hop pipe)",
            "bytes_per_step_each_way": total * N_FFT * 4, "backend": backend, "overlap": "second HIP stream, double-buffered" if on_gpu else "none (host rehearsal)",
            "root_output_finite": ok}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=BATCH, help="streams per GPU (the metric is quoted at 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial", action="store_true", help="one hop at a time, nothing overlapped (dn_process_frame: no hop of added latency) instead of the software-pipelined hop")
    ap.add_argument("--preset", choices=sorted(PRESETS), default="S", help="S = the metric's config; R1/R2 = the reference's own parameters (extra measurements)")
    ap.add_argument("--conv", choices=["fp32", "bf16"], default="fp32", help="bf16 = BASELINE config 3: UNet convs on bf16 MFMA tiles (restated tolerance); side measurement")
    ap.add_argument("--pcie", action="store_true", help="side measurement: frames arrive in pinned host memory and results return to it every hop")
    ap.add_argument("--stream", action="store_true", help="streaming mode (BASELINE config 5): pipe-owned ring/overlap-add/hx state, one hop of new samples per step")
    ap.add_argument("--graph", action="store_true", help="with --stream: replay ONE hipGraph-captured push per step")
    ap.add_argument("--queues", type=int, default=1, help="side measurement: the batch as this many independent pipes on as many HIP streams")
    ap.add_argument("--pipes", type=int, default=0, help="with --queues: the batch as this many pipes (default: one per queue), pipe i on queue i %% queues")
    ap.add_argument("--graph-hops", type=int, default=1, help="with --stream --graph: consecutive pushes captured per graph")
    ap.add_argument("--depth", type=int, default=0, help="hops of one stream in flight (dn_pipe_set_depth, 1..4); 0 = the throughput default for the batch "
                                                            "(4 up to 384 streams, 2 below 1,024, else 1); 1 = output after the next hop.  Used when --group is 0")
    ap.add_argument("--group", type=int, default=-1, help="hops of every stream per launch (dn_pipe_set_group, 0..4: whole Griffin-Lim chains, nothing parked "
                                                            "between launches); -1 = the throughput default (4 up to 384 streams); 0 = one hop per launch (--depth)")
    ap.add_argument("--no-extras", action="store_true", help="skip the bounded side measurements carried in the line (config 3, config 5, R1, batch-1 latency)")
    args = ap.parse_args()

    rehearsal = os.environ.get("DN_BENCH_REHEARSAL") == "1"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    backend = os.environ.get("DN_DIST_BACKEND", "nccl")   # "nccl" = RCCL over xGMI; "gloo" only to rehearse the rank plumbing
    on_gpu = not rehearsal
    if on_gpu and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hop path has no CPU fallback")
    dev = None
    if on_gpu:
        n_dev = torch.cuda.device_count()
        if world > n_dev and not (backend == "gloo" and os.environ.get("DN_ALLOW_SHARED_GPU") == "1"):
            raise SystemExit(f"{world} ranks but {n_dev} GPU(s): one rank per GPU is required")
        dev = torch.device("cuda", local % n_dev)
        torch.cuda.set_device(dev)
    dist = None
    ranks_seen = 1
    # DN_BENCH_FORCE_DIST=1: a ONE-rank job (under a launcher) still builds its process group and runs every collective call of the multi-rank path --
    # how the RCCL branch (init with device_id, all_reduce, barrier with device_ids, the ingress loop's calls) is exercised on a one-GPU box
    force_dist = os.environ.get("DN_BENCH_FORCE_DIST") == "1" and "WORLD_SIZE" in os.environ
    if world > 1 or force_dist:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        t = torch.ones(1, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t)
        ranks_seen = int(t.item())
        if ranks_seen != args.gpus:
            raise SystemExit(f"the {backend} process group sees {ranks_seen} ranks, --gpus asked for {args.gpus}")

    from audio_denoising_amd.shard import shard_range
    B = args.batch
    dn = Rehearsal() if rehearsal else build_denoiser(dev, args.preset, args.conv)
    if on_gpu and (args.preset != "S" or args.stream or args.pcie or (B != BATCH and world == 1)):
        return side_measurement(args, dn, B, dev)
    lo, hi = shard_range(B * world, world, rank)          # this rank's global stream ids
    g = torch.Generator().manual_seed(1234 + rank)
    # Product configuration for throughput (n_fft 1024, about one stream per CU): HOP GROUPS (dn_pipe_set_group): ONE launch carries `group`
    # consecutive hops of every stream -- their front halves (analysis + model + inverse mel) in order, hx handed on inside the launch -- beside the
    # WHOLE Griffin-Lim chains of the hops the previous launch fronted, one wavefront each.  A "step" stays what it was: one hop (256 frames per
    # GPU); K steps are K / group launches.  --group 0: one launch per hop (the deep pipe of round 3, --depth).
    depth, group = 1, 0
    if rehearsal:
        pipe = dn.pipe(B)
    else:
        from audio_denoising_amd.pipeline import HopPipeline
        pipe = None if args.serial else HopPipeline(dn, B)
        if pipe is not None:
            group = args.group if args.group >= 0 else default_group(B, N_FFT)
            if group > 0:
                pipe.set_group(group)
            else:
                depth = args.depth if args.depth > 0 else default_depth(B, N_FFT)
                pipe.set_depth(depth)
    G = max(group, 1)
    # every hop of a group is a different batch of 256 synthetic frames (a launch reads G x 256 frames)
    frames = (0.1 * torch.randn(G, B, N_FFT, generator=g)).to(dev if on_gpu else "cpu")
    hx = dn.init_hx(B)
    out = torch.empty_like(frames)

    def advance(n):
        """n hops of work for every stream"""
        if pipe is None:
            for i in range(n):
                dn.process_frame_(frames[0], hx, out[0], seed=1000 + i, stream_id0=lo)
        elif group > 0:
            for j in range(0, n, group):
                k = min(group, n - j)
                pipe.submit_group(frames[:k], hx, out[:k], seed=1000, stream_id0=lo, check_weights=False)
        else:
            for i in range(n):
                pipe.submit(frames[0], hx, out[0], seed=1000, stream_id0=lo, check_weights=False)

    def sync():
        if on_gpu:
            torch.cuda.synchronize()

    def fence(drain=True):
        if pipe is not None and drain:
            pipe.flush()
        sync()
        if dist is not None:
            dist.barrier(device_ids=[dev.index]) if backend == "nccl" else dist.barrier()
        sync()

    # untimed: bring the GPU out of its idle power state first (a cold start ran the first few hundred hops at roughly
    # half clock: 124 us/step instead of 68), then the W warm-up steps of the contract
    t_pre = time.perf_counter()
    while on_gpu and time.perf_counter() - t_pre < 0.5:
        advance(48)
        sync()
    # Throughput is a steady-state quantity: the pipe stays PRIMED across both boundaries of the timed region (the hops in flight -- one group, or
    # `depth` hops -- are neither drained before t0 nor after the last step), so the launches between the two barrier + synchronize brackets are K
    # hops of work for every stream -- K front halves and K whole chains -- and nothing else.  The drain is timed right after and reported beside
    # it (`drain_ms`, `ms_per_step_with_drain`: what a finite job of K hops pays).
    warm = max(args.warmup, 2 * max(depth, G))
    advance(-(-warm // G) * G)
    fence(drain=False)
    t0 = time.perf_counter()
    advance(args.steps)
    fence(drain=False)
    elapsed = time.perf_counter() - t0
    td = time.perf_counter()
    fence(drain=True)
    drain_s = time.perf_counter() - td
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out).all()
    if rank == 0:
        print(f"[bench] {args.steps} steps x {B * world} frames in {elapsed:.4f} s", file=sys.stderr, flush=True)

    ingress = None
    if (world > 1 or force_dist) and pipe is not None:
        # a side measurement must never cost the headline line: any failure in it (on every rank alike: the loop is collective) is reported, not raised
        try:
            # the headline's schedule: one launch = `group` hops (its double buffering needs only that launch i completes the unit launch i-1 fronted)
            igroup = group if group > 0 else (int(os.environ.get("DN_REHEARSAL_GROUP", "0")) if rehearsal else 0)
            ipipe = dn.pipe(B) if rehearsal else HopPipeline(dn, B)
            if igroup > 0:
                ipipe.set_group(igroup)

            def ifence():
                ipipe.flush()
                fence()
            ingress = ingress_variant(dn, ipipe, B, world, rank, dev, lo, args.steps, args.warmup, dist, backend, ifence, on_gpu, group=igroup)
        except Exception as e:          # noqa: BLE001
            ingress = {"ingress": "scatter_gather", "error": f"{type(e).__name__}: {e}"[:300]}

    line = None
    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        value = B * world * args.steps / elapsed
        line = {
            "metric": METRIC,
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic" if on_gpu else "rehearsal (rank plumbing only: no kernels ran)",
            "ranks_seen": ranks_seen, "ingress": "local",
            "config": {"workload": "configs[1]: batch 256 synthetic 32 ms frames per GPU, n_fft=1024 hop=512 n_mels=80, GRUUNet2 fp32 "
                                   "(dari_tult weights, num_compressed_bins=5), 32-iter Griffin-Lim, device-RNG initial phases, hx carried; "
                                   "a step = one hop of 256 resident synthetic frames per GPU (the hops of a group are different frames; the same buffers "
                                   "are re-processed step after step: the path is compute-bound, no data-dependent work)",
                       "streams_per_gpu": B, "hops_per_launch": G, "frames_per_step": B * world, "sample_rate": SR, "n_fft": N_FFT, "hop": HOP,
                       "n_mels": N_MELS, "griffinlim_iters": GL_ITERS, "conv_precision": args.conv,
                       "parallelism": f"stream-sharded x{world} (no data-path collective)"},
        }
        if ingress is not None:
            line["ingress_variant"] = ingress
            line["ingress_ok"] = "error" not in ingress and bool(ingress.get("root_output_finite", False))
        line["ranks"] = {"seen": ranks_seen, "backend": backend if (world > 1 or force_dist) else None,
                         "rank0_device": (torch.cuda.get_device_name(dev) if on_gpu else "cpu"), "local_rank": local}
    if rank == 0 and on_gpu:
        # the same K steps strictly one after another, nothing overlapped (dn_process_frame: one launch, no added hop of latency)
        f0, o0 = frames[0], out[0]
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for i in range(args.steps):
            dn.process_frame_(f0, hx, o0, seed=5000 + i, stream_id0=lo)
        torch.cuda.synchronize()
        serial_ms = 1e3 * (time.perf_counter() - ts) / args.steps
        kt = staged_kernel_times(dn, f0, hx, 200)
        # the same workload one hop per launch: the one-hop pipe (output after the next submit: wavefront per column + head start) and the deep pipe
        # of round 3 (four hops of a stream in flight as chain segments parked between launches)
        depth1_ms = time_pipe(dn, B, dev, max(args.steps, 200), 1)[1] if (group > 0 or depth != 1 or pipe is None) else ms
        depth4_ms = time_pipe(dn, B, dev, max(args.steps, 200), default_depth(B, N_FFT))[1] if (group > 0 or pipe is None) else ms
        # dominant kernel of the timed region.  Mean launch duration from HIP events recorded on the launch stream around a run of back-to-back
        # launches (one event pair around the run: an event between every two launches adds ~4 us each).
        n_ev = min(args.steps, 100)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        per_launch = 1
        if pipe is not None and group > 0:
            per_launch = group
            n_ev = max(n_ev // group, 5)
            pipe.submit_group(frames, hx, out, seed=1, stream_id0=lo)
            e0.record()
            for i in range(n_ev):
                pipe.submit_group(frames, hx, out, seed=1, stream_id0=lo, check_weights=False)
            e1.record()
            pipe.flush()
            torch.cuda.synchronize()
            dom_ms = e0.elapsed_time(e1) / n_ev
            dom_name = (f"group_kernel (one launch = {group} consecutive hops of work for every stream: the WHOLE Griffin-Lim chains of the {group} hops "
                        f"the previous launch fronted, one wavefront each, + the analysis/model/inverse-mel blocks of {group} new hops, in order)")
            dom_note = ("fp32-VALU bound (78 % of the flops are FFT butterflies on the fp32 vector ALU, the convs run on fp32 MFMA; vector and matrix "
                        "fp32 peaks are both 157.3 TF, so one roof serves both); "
                        f"algorithmic = 6.50 MFLOP per frame (198 rFFT-1024 x 25,600 + mel + convs + inverse mel) x 256 frames x {group} hops per launch")
        elif pipe is not None:
            pipe.submit(f0, hx, o0, seed=1, stream_id0=lo)
            e0.record()
            for i in range(n_ev):
                pipe.submit(f0, hx, o0, seed=1, stream_id0=lo, check_weights=False)
            e1.record()
            pipe.flush()
            torch.cuda.synchronize()
            dom_ms = e0.elapsed_time(e1) / n_ev
            dom_name = ("hop_kernel (Griffin-Lim blocks of the hops in flight + this hop's analysis/model/inverse-mel blocks: "
                        "one launch = one hop of work for every stream)")
            dom_note = ("fp32-VALU bound (78 % of the flops are FFT butterflies on the fp32 vector ALU, the convs run on fp32 MFMA; vector and matrix "
                        "fp32 peaks are both 157.3 TF, so one roof serves both); "
                        "algorithmic = 6.50 MFLOP per frame (198 rFFT-1024 x 25,600 + mel + convs + inverse mel) x 256 frames per launch")
        else:
            e0.record()
            for i in range(n_ev):
                dn.process_frame_(f0, hx, o0, seed=1 + i, stream_id0=lo)
            e1.record()
            torch.cuda.synchronize()
            dom_ms = e0.elapsed_time(e1) / n_ev
            dom_name = "frame_kernel (P1-P12 of one stream per workgroup, nothing overlapped)"
            dom_note = "fp32-VALU bound; algorithmic = 6.50 MFLOP per frame x 256 frames per launch"
        dom_flop = TOTAL_FLOP_PER_FRAME * per_launch
        gl_s = dom_ms * 1e-3
        ach = dom_flop * B / gl_s / 1e12
        traffic = None
        pmc = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            traffic = json.load(open(pmc)).get("frame_kernel_hbm_bytes_per_launch" if pipe is None else
                                               "group_kernel_hbm_bytes_per_launch" if group > 0 else "hop_kernel_hbm_bytes_per_launch")
        bf16 = args.conv == "bf16"
        conv_peak = PEAK_BF16_TFLOPS if bf16 else PEAK_FP32_TFLOPS
        conv_ach = CONV_FLOP_PER_FRAME * B / (kt["cell"] * 1e-3) / 1e12
        line.update({
            "roofline": {"bound": "fp32-valu", "bound_detail": "fp32 vector ALU (FFT butterflies, 78 % of the flops) + fp32 MFMA (convs): the fp32 compute roof, "
                                                               "157.3 TFLOP/s either way; not HBM (hbm_frac) and not the matrix pipe",
                         "kernel": dom_name, "achieved": round(ach, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / PEAK_FP32_TFLOPS, 4), "traffic": traffic,
                         "traffic_source": None if traffic is None else "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes taken by "
                                                                        "tools/collect_profiles.sh beside the committed kernel statistics (2 x FETCH + WRITE, per launch); "
                                                                        "counters cannot be read from inside this process, so this is NOT a measurement of this run",
                         "note": dom_note,
                         "launch_ms": round(dom_ms, 4), "hops_per_launch": per_launch, "ms_per_hop": round(dom_ms / per_launch, 4),
                         "algorithmic_bytes_per_launch": HBM_BYTES_PER_FRAME * B * per_launch,
                         "hbm_frac": round(HBM_BYTES_PER_FRAME * B * per_launch / gl_s / 1e9 / PEAK_HBM_GBS, 6)},
            "kernel_ms": {k: round(v, 4) for k, v in kt.items()},
            # MFMA utilisation of the UNet convs: conv FLOPs / (stand-alone cell kernel time x MFMA peak of the conv dtype)
            "conv_mfma": {"kernel": "cell_kernel_bf16 (v_mfma_f32_16x16x32_bf16)" if bf16 else "cell_kernel (fp32 v_mfma_f32_16x16x4_f32)",
                          "achieved": round(conv_ach, 3), "peak": conv_peak, "unit": "TFLOP/s", "frac": round(conv_ach / conv_peak, 5),
                          "note": "0.24 GFLOP per batch-256 launch is 1.5 us at the fp32 peak: structurally latency-bound (SURVEY section 7)"},
            "schedule": "unpipelined, 1 launch per hop (dn_process_frame), no added latency" if pipe is None
                        else f"hop groups (dn_pipe_submit_group): 1 launch per {group} consecutive hops of every stream; their front halves run in order inside "
                             f"the launch beside the whole Griffin-Lim chains of the previous {group} hops (one wavefront each, never parked: bit-identical "
                             f"to the one-hop pipe); input arrives {group} hops at a time, the output of a group is complete one launch later" if group > 0
                        else f"software-pipelined, 1 launch per hop (dn_pipe_submit), depth {depth}: {depth} hops of every stream in flight, "
                             f"output complete {depth} launch(es) after its submit" + ("" if depth == 1 else
                             "; the Griffin-Lim chain of a frame runs as one segment per launch, one wavefront per stream and segment (bit-identical to depth 1)"),
            "pipeline_depth": depth if pipe is not None else 0,
            "hops_per_launch": G,
            "timed_region": "steady state: K hops of work for every stream between barrier + synchronize brackets, the pipe primed on both sides (K front "
                            "halves and K whole Griffin-Lim chains" + (f" in K / {group} launches" if group > 0 else "") + "); the drain is timed separately",
            "drain_ms": round(1e3 * drain_s, 4),
            "ms_per_step_with_drain": round(1e3 * (elapsed + drain_s) / args.steps, 4),
            "depth1_ms_per_step": round(depth1_ms, 4),
            "depth4_ms_per_step": round(depth4_ms, 4),
            "serial_ms_per_step": round(serial_ms, 4),
            "whole_path": {"tflops": round(TOTAL_FLOP_PER_FRAME * value / 1e12, 3),
                           "fp32_frac": round(TOTAL_FLOP_PER_FRAME * value / 1e12 / (PEAK_FP32_TFLOPS * world), 4),
                           "hbm_frac": round(HBM_BYTES_PER_FRAME * value / 1e9 / (PEAK_HBM_GBS * world), 6)},
        })
        if world == 1 and not args.no_extras:
            try:
                line.update(extra_measurements(args, dev))
            except Exception as e:          # noqa: BLE001  (a side measurement must never cost the headline line)
                line["extras_error"] = f"{type(e).__name__}: {e}"[:300]
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
    if dist is not None:
        dist.barrier(device_ids=[dev.index]) if backend == "nccl" else dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
