"""Kernel LOGIC checks without a GPU: the package's .hip sources are compiled for the host
against tests/emu/hip/hip_runtime.h (one std::thread per work-item) and driven through the same
C ABI, then compared with the oracle.  Small batches only (a work-item is an OS thread here).
The parity tests proper are the -m gpu tests; this tier exists so index maths and hand-offs can
be debugged in the GPU-less container."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import emu  # noqa: E402
from audio_denoising_amd._lib import DN_PEAK_NORMALIZE, DN_PRE_WINDOW, DnError, DspCfg, ModelCfg  # noqa: E402
from oracle import dsp_ref, model_ref, pipeline_ref  # noqa: E402

P = pipeline_ref.PARAMS_S


@pytest.fixture(scope="module")
def lib():
    return emu.load()


@pytest.fixture(scope="module")
def dsp(lib):
    fb = dsp_ref.melscale_fbanks(P.n_stft, P.n_mels, P.sample_rate).numpy()
    h = C.c_void_p()
    lib.check(lib.dn_dsp_create(C.byref(DspCfg(P.sample_rate, P.n_fft, P.hop, P.n_mels)), emu.ptr(emu.f32(fb)), None, None, C.byref(h)))
    yield h
    lib.dn_dsp_destroy(h)


def make_model(lib, C_bins=5, short="dari_tult"):
    w = np.fromfile(os.path.join(GOLDEN, f"weights_{short}.bin"), dtype=np.float32)
    h = C.c_void_p()
    lib.check(lib.dn_model_create(emu.ptr(w), w.size, C.byref(ModelCfg(C_bins, 1, 4, 17, 3, 2, 1, 6)), C.byref(h)))
    return h


def test_stft_matches_oracle(lib, dsp):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, P.n_fft, generator=g)
    spec = np.zeros((2, 3, P.n_stft, 2), np.float32)
    lib.check(lib.dn_stft(dsp, emu.ptr(emu.f32(x.numpy())), emu.ptr(spec), 2, 0, None))
    ref = dsp_ref.spectrogram(x, P.n_fft, P.hop).numpy()
    got = (spec[..., 0] + 1j * spec[..., 1]).transpose(0, 2, 1)
    assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max() + 1e-5


def test_analysis_matches_oracle_including_silent_frames(lib, dsp):
    g = load_golden("dsp_S.npz")
    frames = emu.f32(g["frames"])
    B = frames.shape[0]
    mel = np.zeros((B, 3, P.n_mels), np.float32)
    peak = np.zeros(B, np.float32)
    lib.check(lib.dn_stft_mel_log1p(dsp, emu.ptr(frames), emu.ptr(mel), emu.ptr(peak), B, DN_PEAK_NORMALIZE | DN_PRE_WINDOW, None))
    assert np.abs(mel - g["model_input"]).max() <= 2e-5
    assert np.array_equal(peak, g["peak"])


def test_plan_tables_native_fb_and_pinv(lib):
    """fb/pinv computed natively (NULL inputs) agree with the oracle's filterbank and numpy pinv."""
    h = C.c_void_p()
    lib.check(lib.dn_dsp_create(C.byref(DspCfg(P.sample_rate, P.n_fft, P.hop, P.n_mels)), None, None, None, C.byref(h)))
    fb = np.zeros((P.n_stft, P.n_mels), np.float32)
    pinv = np.zeros((P.n_stft, P.n_mels), np.float32)
    win = np.zeros(P.n_fft, np.float32)
    lib.check(lib.dn_dsp_get_tables(h, emu.ptr(fb), emu.ptr(pinv), emu.ptr(win)))
    lib.dn_dsp_destroy(h)
    ref_fb = dsp_ref.melscale_fbanks(P.n_stft, P.n_mels, P.sample_rate).numpy()
    assert np.abs(fb - ref_fb).max() <= 5e-5
    assert np.abs(pinv - np.linalg.pinv(fb.astype(np.float64).T)).max() <= 1e-5
    # the native default is the exactly rounded periodic Hann; torch.hann_window (fp32 cos) is within 2e-7 of it,
    # which is why the Python host passes torch's own window to dn_dsp_create
    assert np.abs(win - torch.hann_window(P.n_fft).numpy()).max() <= 2.5e-7


def test_residual_invmel_matches_oracle(lib, dsp):
    g = load_golden("dsp_S.npz")
    B = g["frames"].shape[0]
    lin = np.zeros((B, 3, P.n_stft), np.float32)
    lib.check(lib.dn_residual_invmel(dsp, emu.ptr(emu.f32(g["model_input"])), emu.ptr(emu.f32(g["predicted_diff"])), emu.ptr(lin), B, 3, None))
    ref = g["lin_mag"].transpose(0, 2, 1)
    assert np.abs(lin - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max())


def test_griffinlim_and_istft_match_oracle(lib, dsp):
    g = load_golden("dsp_S.npz")
    B = 2
    mag = emu.f32(g["lin_mag"][:B].transpose(0, 2, 1))
    init = g["init_angles"][:B].transpose(0, 2, 1)
    init_ri = emu.f32(np.stack([init.real, init.imag], axis=-1))
    wave = np.zeros((B, P.n_fft), np.float32)
    lib.check(lib.dn_griffinlim(dsp, emu.ptr(mag), emu.ptr(init_ri), 0, 0, None, emu.ptr(wave), B, 32, 0.99, None))
    ref = dsp_ref.griffinlim(torch.from_numpy(g["lin_mag"][:B]), P.n_fft, P.hop, init_angles=torch.from_numpy(g["init_angles"][:B])).numpy()
    rms = np.sqrt(np.mean((wave - ref) ** 2))
    assert rms <= 1e-3 * max(1.0, np.sqrt(np.mean(ref ** 2))), rms
    # istft (n_iter = 0 path): invert an actual STFT
    x = torch.from_numpy(g["frames"][:B])
    spec = dsp_ref.spectrogram(x, P.n_fft, P.hop).numpy().transpose(0, 2, 1)
    spec_ri = emu.f32(np.stack([spec.real, spec.imag], axis=-1))
    lib.check(lib.dn_istft(dsp, emu.ptr(spec_ri), emu.ptr(wave), B, None))
    assert np.abs(wave - x.numpy()).max() <= 1e-5


def test_griffinlim_device_rng_is_shard_invariant(lib, dsp):
    """With no init_angles the phases come from the counter-based generator: n_iter=0 with unit
    magnitude returns istft(angles), so two launches that place the same global stream ids in
    different batch slots must agree exactly.  (What the generator IS: test_device_rng_is_philox4x32_10...)"""
    ones = np.ones((2, 3, P.n_stft), np.float32)
    a = np.zeros((2, P.n_fft), np.float32)
    b = np.zeros((1, P.n_fft), np.float32)
    lib.check(lib.dn_griffinlim(dsp, emu.ptr(ones), None, 1234, 10, None, emu.ptr(a), 2, 0, 0.99, None))
    lib.check(lib.dn_griffinlim(dsp, emu.ptr(ones[:1]), None, 1234, 11, None, emu.ptr(b), 1, 0, 0.99, None))
    assert np.array_equal(a[1], b[0]) and not np.array_equal(a[0], a[1])


@pytest.mark.parametrize("name", ["cell_dari_tult_B4_T3_F80.npz", "cell_dari_tult_B4_T3_F64.npz", "cell_dari_tult2_B3_T7_F80.npz",
                                  "cell_dari_tult_B2_T1_F64.npz"])
def test_cell_matches_reference_golden(lib, name):
    g = load_golden(name)
    B, T, F = g["x"].shape
    Cb = F // 16
    m = make_model(lib, Cb, "dari_tult2" if "tult2" in name else "dari_tult")
    out = np.zeros((B, T, F), np.float32)
    hx1 = np.zeros((B, 17, Cb), np.float32)
    lib.check(lib.dn_cell_forward(m, emu.ptr(emu.f32(g["x"])), emu.ptr(emu.f32(g["hx0"])), emu.ptr(out), emu.ptr(hx1), B, T, F, Cb, None))
    lib.dn_model_destroy(m)
    assert np.abs(out - g["out"]).max() <= 1e-4          # north_star tolerance on the mel residual
    assert np.abs(hx1 - g["hx1"]).max() <= 1e-4


def test_cell_shape_mismatch_is_an_error(lib):
    m = make_model(lib, 4)
    x = np.zeros((1, 3, 80), np.float32)
    out = np.zeros_like(x)
    hx = np.zeros((1, 17, 4), np.float32)
    rc = lib.dn_cell_forward(m, emu.ptr(x), emu.ptr(hx), emu.ptr(out), emu.ptr(hx), 1, 3, 80, 4, None)
    assert rc == -1 and b"compress" in lib.dn_last_error()
    with pytest.raises(DnError):
        lib.check(rc)
    lib.dn_model_destroy(m)


def test_process_frame_and_stream_step_match_oracle(lib, dsp):
    g = load_golden("stream_S.npz")
    sig, inits = g["signal"][:2], g["init_angles"][:, :2]
    B, n_hops = 2, 3
    m = make_model(lib, 5)
    ws = np.zeros(lib.dn_workspace_bytes(dsp, B) // 4 + 16, np.float32)
    ring = np.zeros((B, P.n_fft), np.float32)
    ring[:, P.hop:] = sig[:, :P.n_fft - P.hop]
    ola = np.zeros((B, P.n_fft), np.float32)
    hx = np.zeros((B, 17, 5), np.float32)
    outs = []
    for h in range(n_hops):
        hop_in = emu.f32(sig[:, P.n_fft - P.hop + h * P.hop: P.n_fft + h * P.hop])
        ia = inits[h].transpose(0, 2, 1)
        ia = emu.f32(np.stack([ia.real, ia.imag], axis=-1))
        hop_out = np.zeros((B, P.hop), np.float32)
        lib.check(lib.dn_stream_step(m, dsp, emu.ptr(hop_in), emu.ptr(ring), emu.ptr(ola), emu.ptr(hx), emu.ptr(hop_out),
                                     emu.ptr(ia), 0, 0, 32, 0.99, emu.ptr(ws), B, 0, None))
        outs.append(hop_out)
    lib.dn_model_destroy(m)
    got = np.concatenate(outs, axis=1)
    ref = g["out"][:2, :n_hops * P.hop]
    rms = np.sqrt(np.mean((got - ref) ** 2))
    assert rms <= 1e-3, rms


def test_pipelined_hops_equal_serial_hops(lib, dsp):
    """dn_pipe_* (one launch = previous hop's Griffin-Lim blocks + this hop's front blocks) against dn_process_frame."""
    g = load_golden("stream_S.npz")
    B, n_hops = 2, 3
    m = make_model(lib, 5)
    ws = np.zeros(lib.dn_workspace_bytes(dsp, B) // 4 + 16, np.float32)
    frames = [emu.f32(g["signal"][:B, h * P.hop: h * P.hop + P.n_fft]) for h in range(n_hops)]
    hx_a = np.zeros((B, 17, 5), np.float32)
    outs_a = [np.zeros((B, P.n_fft), np.float32) for _ in range(n_hops)]
    for h in range(n_hops):
        lib.check(lib.dn_process_frame(m, dsp, emu.ptr(frames[h]), emu.ptr(hx_a), emu.ptr(outs_a[h]), None, None, 11 + h, 3, 8, 0.99,
                                       emu.ptr(ws), B, 0, None))
    pipe = C.c_void_p()
    lib.check(lib.dn_pipe_create(m, dsp, B, 0, C.byref(pipe)))
    hx_b = np.zeros((B, 17, 5), np.float32)
    outs_b = [np.zeros((B, P.n_fft), np.float32) for _ in range(n_hops)]
    for h in range(n_hops):
        lib.check(lib.dn_pipe_submit(pipe, emu.ptr(frames[h]), emu.ptr(hx_b), emu.ptr(outs_b[h]), None, 11, 3, 8, 0.99, None))     # frame h draws from seed + h
    lib.check(lib.dn_pipe_flush(pipe, 8, 0.99, None))
    lib.dn_pipe_destroy(pipe)
    lib.dn_model_destroy(m)
    assert np.array_equal(hx_a, hx_b)
    for a, b in zip(outs_a, outs_b):
        assert np.abs(a - b).max() <= 1e-5      # fused-prologue vs separate inverse-mel launch differ only in summation order


def test_streaming_pipe_matches_oracle_with_one_hop_delay(lib, dsp):
    """dn_pipe_stream_*: ring shift, frame, overlap-add and emission inside the one-launch hop; float and int16 I/O."""
    g = load_golden("stream_S.npz")
    B, n_frames = 2, 3
    sig, inits = g["signal"][:B], g["init_angles"][:, :B]
    m = make_model(lib, 5)
    pipe = C.c_void_p()
    lib.check(lib.dn_pipe_stream_create(m, dsp, B, 0, C.byref(pipe)))
    outs, keep = [], []
    for p_i in range(n_frames + 1):                      # push 0 primes the ring; push f+1 delivers frame f
        hop_in = emu.f32(sig[:, p_i * P.hop:(p_i + 1) * P.hop])
        ia = None
        if p_i >= 1:
            a = inits[p_i - 1].transpose(0, 2, 1)
            ia = emu.f32(np.stack([a.real, a.imag], axis=-1))
        keep.append((hop_in, ia))
        out = np.full((B, P.hop), 7.0, np.float32)
        lib.check(lib.dn_pipe_stream_push(pipe, emu.ptr(hop_in), 0, emu.ptr(out), 0, emu.ptr(ia), 0, 0, 32, 0.99, None))
        outs.append(out)
    last = np.zeros((B, P.hop), np.float32)
    lib.check(lib.dn_pipe_stream_flush(pipe, emu.ptr(last), 0, 32, 0.99, None))
    outs.append(last)
    assert np.all(outs[0] == 0) and np.all(outs[1] == 0)                      # nothing to emit yet
    got = np.concatenate(outs[2:], axis=1)                                    # segments of frames 0..n_frames-1
    ref = g["out"][:B, :n_frames * P.hop]
    assert np.sqrt(np.mean((got - ref) ** 2)) <= 1e-3
    lib.dn_pipe_destroy(pipe)
    # int16 in / int16 out: quantised input through the float path must give the same samples, quantised
    pipe = C.c_void_p()
    lib.check(lib.dn_pipe_stream_create(m, dsp, B, 0, C.byref(pipe)))
    pipe_f = C.c_void_p()
    lib.check(lib.dn_pipe_stream_create(m, dsp, B, 0, C.byref(pipe_f)))
    for p_i in range(3):
        q = np.clip(np.round(sig[:, p_i * P.hop:(p_i + 1) * P.hop] * 3.0 * 32767), -32768, 32767).astype(np.int16)
        qf = emu.f32(q.astype(np.float32) / np.float32(32767))
        o16 = np.zeros((B, P.hop), np.int16)
        of = np.zeros((B, P.hop), np.float32)
        lib.check(lib.dn_pipe_stream_push(pipe, emu.ptr(np.ascontiguousarray(q)), 1, emu.ptr(o16), 1, None, 5, 0, 4, 0.99, None))
        lib.check(lib.dn_pipe_stream_push(pipe_f, emu.ptr(qf), 0, emu.ptr(of), 0, None, 5, 0, 4, 0.99, None))
        assert np.array_equal(o16, (np.clip(of, -1, 1) * 32767).astype(np.int16))
    lib.dn_pipe_destroy(pipe)
    lib.dn_pipe_destroy(pipe_f)
    lib.dn_model_destroy(m)


def test_host_transports_emit_the_device_fed_samples(lib, dsp, monkeypatch):
    """dn_pipe_stream_push_host on the emulator: staged (pageable buffers), zero copy direct and zero copy deferred (DN_HOST_DEFER: the next
    launch carries the samples out; a wait on the newest push moves them itself), mixed on one pipe, against dn_pipe_stream_push.  The
    emulated runtime calls a pointer page-locked when DN_EMU_PINNED=1."""
    from audio_denoising_amd import _lib
    g = load_golden("stream_S.npz")
    B, n, n_iter = 2, 6, 1
    m = make_model(lib, 5)
    hops = [np.ascontiguousarray(np.clip(g["signal"][:B, h * P.hop:(h + 1) * P.hop] * 32767.0, -32767, 32767).astype(np.int16)) for h in range(n)]
    ref = C.c_void_p()
    lib.check(lib.dn_pipe_stream_create(m, dsp, B, 0, C.byref(ref)))
    want = []
    for h in hops:
        o = np.zeros((B, P.hop), np.int16)
        lib.check(lib.dn_pipe_stream_push(ref, emu.ptr(h), 1, emu.ptr(o), 1, None, 11, 3, n_iter, 0.99, None))
        want.append(o)
    lib.dn_pipe_destroy(ref)
    assert any(w.any() for w in want)
    D, Z, S = _lib.DN_HOST_DEFER, 0, _lib.DN_HOST_STAGED
    flags = [D, D, Z, D, S, D]
    wait_at_once = {1, 2}                 # (push 1: the wait on the newest deferred push has to move its samples itself)
    pipe = C.c_void_p()
    lib.check(lib.dn_pipe_stream_create(m, dsp, B, 0, C.byref(pipe)))
    outs = [np.full((B, P.hop), -7, np.int16) for _ in range(n)]
    for i in range(n):
        monkeypatch.setenv("DN_EMU_PINNED", "0" if flags[i] == S else "1")
        t = C.c_uint64()
        lib.check(lib.dn_pipe_stream_push_host(pipe, emu.ptr(hops[i]), 1, emu.ptr(outs[i]), 1, 11, 3, n_iter, 0.99, flags[i], None, C.byref(t)))
        assert t.value == i
        if flags[i] == D:
            assert np.all(outs[i] == -7)                     # still in the staging buffer
        if i in wait_at_once:
            lib.check(lib.dn_pipe_stream_host_wait(pipe, i))
            assert np.array_equal(outs[i], want[i]), f"push {i}"
    for i in reversed(range(n)):
        lib.check(lib.dn_pipe_stream_host_wait(pipe, i))
        assert np.array_equal(outs[i], want[i]), f"push {i}"
    assert lib.dn_pipe_stream_host_wait(pipe, n) != 0        # no such push
    lib.dn_pipe_destroy(pipe)
    lib.dn_model_destroy(m)


# ------------------------------------------------------------------ the app's own parameters: n_fft 1536 (768 = 4*4*4*12)
P1 = pipeline_ref.PARAMS_R1


@pytest.fixture(scope="module")
def dsp_r1(lib):
    fb = dsp_ref.melscale_fbanks(P1.n_stft, P1.n_mels, P1.sample_rate).numpy()
    h = C.c_void_p()
    lib.check(lib.dn_dsp_create(C.byref(DspCfg(P1.sample_rate, P1.n_fft, P1.hop, P1.n_mels)), emu.ptr(emu.f32(fb)), None,
                                emu.ptr(emu.f32(torch.hann_window(P1.n_fft).numpy())), C.byref(h)))
    yield h
    lib.dn_dsp_destroy(h)


def test_r1_stft_and_istft_mixed_radix(lib, dsp_r1):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, P1.n_fft, generator=g)
    spec = np.zeros((2, 3, P1.n_stft, 2), np.float32)
    lib.check(lib.dn_stft(dsp_r1, emu.ptr(emu.f32(x.numpy())), emu.ptr(spec), 2, 0, None))
    ref = dsp_ref.spectrogram(x, P1.n_fft, P1.hop).numpy()
    got = (spec[..., 0] + 1j * spec[..., 1]).transpose(0, 2, 1)
    assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max() + 1e-5
    wave = np.zeros((2, P1.n_fft), np.float32)
    lib.check(lib.dn_istft(dsp_r1, emu.ptr(spec), emu.ptr(wave), 2, None))
    assert np.abs(wave - x.numpy()).max() <= 2e-5


def test_r1_process_frame_matches_oracle(lib, dsp_r1):
    g = load_golden("dsp_R1.npz")
    B = 2
    m = make_model(lib, 4)
    ws = np.zeros(lib.dn_workspace_bytes(dsp_r1, B) // 4 + 16, np.float32)
    frames = emu.f32(g["frames"][:B])
    hx = np.zeros((B, 17, 4), np.float32)
    out = np.zeros((B, P1.n_fft), np.float32)
    resid = np.zeros((B, 3, P1.n_mels), np.float32)
    ia = g["init_angles"][:B].transpose(0, 2, 1)
    ia = emu.f32(np.stack([ia.real, ia.imag], axis=-1))
    lib.check(lib.dn_process_frame(m, dsp_r1, emu.ptr(frames), emu.ptr(hx), emu.ptr(out), emu.ptr(resid), emu.ptr(ia), 0, 0, 32, 0.99,
                                   emu.ptr(ws), B, 0, None))
    lib.dn_model_destroy(m)
    assert np.abs(resid - g["predicted_diff"][:B]).max() <= 1e-4
    assert np.abs(hx - g["hx"][:B]).max() <= 1e-4
    assert np.sqrt(np.mean((out - g["out"][:B]) ** 2)) <= 1e-3


@pytest.mark.parametrize("name", ["cell_dari_tult_B4_T3_F80.npz", "cell_dari_tult2_B4_T3_F64.npz"])
def test_cell_bf16_conv_tiles_within_restated_tolerance(lib, name):
    """BASELINE config 3 (bf16 MFMA conv tiles): tolerance restated -- <= 5e-1 max-abs, <= 1e-2 relative RMS (see tests/test_gpu_parity.py)."""
    g = load_golden(name)
    B, T, F = g["x"].shape
    Cb = F // 16
    m = make_model(lib, Cb, "dari_tult2" if "tult2" in name else "dari_tult")
    out = np.zeros((B, T, F), np.float32)
    hx1 = np.zeros((B, 17, Cb), np.float32)
    lib.check(lib.dn_cell_forward_bf16(m, emu.ptr(emu.f32(g["x"])), emu.ptr(emu.f32(g["hx0"])), emu.ptr(out), emu.ptr(hx1), B, T, F, Cb, None))
    lib.dn_model_destroy(m)
    err = out - g["out"]
    assert np.abs(err).max() <= 5e-1
    assert np.sqrt(np.mean(err ** 2)) / np.sqrt(np.mean(g["out"] ** 2)) <= 1e-2
    assert np.abs(err).max() > 1e-5            # it really is the reduced-precision path


def test_server_variant_matches_oracle(lib):
    """server.py:199-217 (R2 parameters, checkpoint GRUUNet2-good): general-length STFT, model over T columns with
    hx*0.9, relu*3 / exp-1 / inverse mel / polar with the noisy phase, general-length ISTFT."""
    from oracle import pipeline_ref
    p = pipeline_ref.PARAMS_R2
    g = load_golden("server_R2.npz")
    fb = dsp_ref.melscale_fbanks(p.n_stft, p.n_mels, p.sample_rate).numpy()
    d = C.c_void_p()
    lib.check(lib.dn_dsp_create(C.byref(DspCfg(p.sample_rate, p.n_fft, p.hop, p.n_mels)), emu.ptr(emu.f32(fb)), None,
                                emu.ptr(emu.f32(torch.hann_window(p.n_fft).numpy())), C.byref(d)))
    m = make_model(lib, 4, "good")
    B, L = g["chunks"].shape[1:]
    T = 1 + L // p.hop
    hx = np.zeros((B, 17, 4), np.float32)
    for c in range(2):
        x = emu.f32(g["chunks"][c])
        spec = np.zeros((B, T, p.n_stft, 2), np.float32)
        logmel = np.zeros((B, T, p.n_mels), np.float32)
        out = np.zeros_like(logmel)
        wave = np.zeros((B, p.hop * (T - 1)), np.float32)
        lib.check(lib.dn_stft_general(d, emu.ptr(x), emu.ptr(spec), emu.ptr(logmel), B, L, None))
        lib.check(lib.dn_cell_forward_ex(m, emu.ptr(logmel), emu.ptr(hx), emu.ptr(out), emu.ptr(hx), B, T, p.n_mels, 4, 0.9, None))
        if c == 0:
            assert np.abs(logmel - g["log_mel0"].transpose(0, 2, 1)).max() <= 2e-5
            assert np.abs(out - g["model_out0"]).max() <= 1e-4
        lib.check(lib.dn_server_rows(d, emu.ptr(logmel), emu.ptr(out), emu.ptr(spec), emu.ptr(spec), B * T, None))
        lib.check(lib.dn_istft_general(d, emu.ptr(spec), emu.ptr(wave), B, T, None))
        assert np.sqrt(np.mean((wave - g["out"][c]) ** 2)) <= 1e-3 * max(1.0, np.sqrt(np.mean(g["out"][c] ** 2)))
    lib.dn_model_destroy(m)
    lib.dn_dsp_destroy(d)


# ------------------------------------------------------------------ sibling model MOMO3 on the general conv tiles
@pytest.mark.parametrize("name", ["momo3_B4_T3_F22.npz", "momo3_B3_T7_F24.npz", "momo3_B2_T1_F23.npz"])
def test_momo3_matches_reference_golden(lib, name):
    from audio_denoising_amd._lib import MomoCfg
    g = load_golden(name)
    B, T, F = g["x"].shape
    Cb = g["hx0"].shape[2]
    w = np.fromfile(os.path.join(GOLDEN, "weights_momo3_4d4ea0.bin"), dtype=np.float32)
    h = C.c_void_p()
    lib.check(lib.dn_momo_create(emu.ptr(w), w.size, C.byref(MomoCfg(Cb, 1, 3, 16, 3, 2, (C.c_int32 * 3)(1, 0, 1), 6)), C.byref(h)))
    out = np.zeros((B, T, F), np.float32)
    hx1 = np.zeros((B, 16, Cb), np.float32)
    last = np.zeros((B, F), np.float32)
    prev = emu.f32(g["prev"].reshape(B, F)) if "prev" in g.files else None
    lib.check(lib.dn_momo_forward(h, emu.ptr(emu.f32(g["x"])), emu.ptr(emu.f32(g["hx0"])), emu.ptr(prev), emu.ptr(out), emu.ptr(hx1),
                                  emu.ptr(last), B, T, F, Cb, None))
    assert np.abs(out - g["out"]).max() <= 1e-4 and np.abs(hx1 - g["hx1"]).max() <= 1e-4
    assert np.array_equal(last, g["x"][:, -1, :])
    rc = lib.dn_momo_forward(h, emu.ptr(emu.f32(g["x"])), emu.ptr(emu.f32(g["hx0"])), None, emu.ptr(out), emu.ptr(hx1), None, B, T, F, Cb + 1, None)
    assert rc == -1 and b"compress" in lib.dn_last_error()
    lib.dn_momo_destroy(h)


def test_griffinlim_head_start_is_bit_identical(lib, dsp):
    """dn_pipe_set_head_start: the front workgroup runs the first iterations of its frame's Griffin-Lim chain and parks it in HBM; the
    next launch resumes.  Cutting the chain at the top of an iteration must not change a bit (ten iterations here; 32 and batch 256 in the gpu tier)."""
    g = load_golden("stream_S.npz")
    B, n_hops = 2, 3
    m = make_model(lib, 5)
    frames = [emu.f32(g["signal"][:B, h * P.hop: h * P.hop + P.n_fft]) for h in range(n_hops)]
    res = {}
    for split in (0, 3, 10):
        pipe = C.c_void_p()
        lib.check(lib.dn_pipe_create(m, dsp, B, 0, C.byref(pipe)))
        lib.check(lib.dn_pipe_set_head_start(pipe, split))
        hx = np.zeros((B, 17, 5), np.float32)
        outs = [np.zeros((B, P.n_fft), np.float32) for _ in range(n_hops)]
        for h in range(n_hops):
            lib.check(lib.dn_pipe_submit(pipe, emu.ptr(frames[h]), emu.ptr(hx), emu.ptr(outs[h]), None, 11, 3, 10, 0.99, None))
        lib.check(lib.dn_pipe_flush(pipe, 10, 0.99, None))
        lib.dn_pipe_destroy(pipe)
        res[split] = (hx, outs)
    for split in (3, 10):
        assert np.array_equal(res[0][0], res[split][0])
        for a, b in zip(res[0][1], res[split][1]):
            assert np.array_equal(a, b)
    lib.dn_model_destroy(m)


def test_pending_hop_is_finished_with_its_own_n_iter_momentum_and_destination(lib, dsp):
    """A hop submitted with n_iter = 12 and a head start of 5 iterations, then flushed with n_iter = 2 (the state the Python front ends
    reach when Denoiser.n_iter is lowered between submit() and flush()): the pending chain resumes at iteration 5 > 2 -- it must finish
    with the n_iter / momentum of ITS submit (they travel in the scratch slot) and terminate.  Likewise its destination: a submit to
    another buffer in between does not redirect it."""
    g = load_golden("stream_S.npz")
    B = 2
    m = make_model(lib, 5)
    f0 = emu.f32(g["signal"][:B, :P.n_fft])
    f1 = emu.f32(g["signal"][:B, P.hop:P.hop + P.n_fft])
    ref, got = [], []
    for mode in ("same", "changed"):
        pipe = C.c_void_p()
        lib.check(lib.dn_pipe_create(m, dsp, B, 0, C.byref(pipe)))
        lib.check(lib.dn_pipe_set_head_start(pipe, 5))
        hx = np.zeros((B, 17, 5), np.float32)
        o0, o1 = np.zeros((B, P.n_fft), np.float32), np.zeros((B, P.n_fft), np.float32)
        lib.check(lib.dn_pipe_submit(pipe, emu.ptr(f0), emu.ptr(hx), emu.ptr(o0), None, 11, 3, 12, 0.99, None))
        if mode == "same":
            lib.check(lib.dn_pipe_submit(pipe, emu.ptr(f1), emu.ptr(hx), emu.ptr(o1), None, 11, 3, 12, 0.99, None))
            lib.check(lib.dn_pipe_flush(pipe, 12, 0.99, None))
            ref = [o0.copy(), o1.copy()]
        else:
            # second hop submitted with other settings, flushed with yet others: each hop keeps the (12, 0.99) of its own submit
            lib.check(lib.dn_pipe_submit(pipe, emu.ptr(f1), emu.ptr(hx), emu.ptr(o1), None, 11, 3, 12, 0.99, None))
            lib.check(lib.dn_pipe_flush(pipe, 2, 0.5, None))
            got = [o0.copy(), o1.copy()]
        lib.dn_pipe_destroy(pipe)
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)
    # n_iter really is per frame: a hop submitted with n_iter = 3 differs from one with 12 and equals the unpipelined hop with 3
    pipe = C.c_void_p()
    lib.check(lib.dn_pipe_create(m, dsp, B, 0, C.byref(pipe)))
    lib.check(lib.dn_pipe_set_head_start(pipe, 5))          # more head start than the frame has iterations
    hx = np.zeros((B, 17, 5), np.float32)
    o0 = np.zeros((B, P.n_fft), np.float32)
    lib.check(lib.dn_pipe_submit(pipe, emu.ptr(f0), emu.ptr(hx), emu.ptr(o0), None, 11, 3, 3, 0.99, None))
    lib.check(lib.dn_pipe_flush(pipe, 12, 0.99, None))
    lib.dn_pipe_destroy(pipe)
    ws = np.zeros(int(lib.lib.dn_workspace_bytes(dsp, B)), np.uint8)
    hx2 = np.zeros((B, 17, 5), np.float32)
    o2 = np.zeros((B, P.n_fft), np.float32)
    lib.check(lib.dn_process_frame(m, dsp, emu.ptr(f0), emu.ptr(hx2), emu.ptr(o2), None, None, 11, 3, 3, 0.99, emu.ptr(ws), B, 0, None))
    assert np.array_equal(o0, o2) and not np.array_equal(o0, ref[0])
    lib.dn_model_destroy(m)


def test_mel_stages_at_a_filter_count_that_is_not_a_multiple_of_16(lib):
    """n_mels = 40 through the public transform entry points (dn_dsp_create accepts 0..128): the packed mel schedule covers the partial last
    group of filters, and the dense inverse mel (explicit pinv) walks zero-padded rows instead of reading past the matrix."""
    M = 40
    fb = dsp_ref.melscale_fbanks(P.n_stft, M, P.sample_rate).numpy()
    g = torch.Generator().manual_seed(5)
    frames = (0.1 * torch.randn(2, P.n_fft, generator=g)).numpy()
    ref_spec = dsp_ref.spectrogram(torch.from_numpy(frames), P.n_fft, P.hop)
    ref_mel = torch.log1p(torch.matmul(ref_spec.abs().transpose(-1, -2), torch.from_numpy(fb))).numpy()      # (B, 3, M)
    for pinv in (None, np.linalg.pinv(fb.astype(np.float64).T).astype(np.float32)):
        h = C.c_void_p()
        lib.check(lib.dn_dsp_create(C.byref(DspCfg(P.sample_rate, P.n_fft, P.hop, M)), emu.ptr(emu.f32(fb)),
                                    None if pinv is None else emu.ptr(emu.f32(pinv)), None, C.byref(h)))
        mel = np.full((2, 3, M), np.nan, np.float32)
        lib.check(lib.dn_stft_mel_log1p(h, emu.ptr(emu.f32(frames)), emu.ptr(mel), None, 2, 0, None))
        assert np.abs(mel - ref_mel).max() <= 2e-5 * max(1.0, np.abs(ref_mel).max())
        mag = np.expm1(ref_mel).astype(np.float32)
        lin = np.full((2, 3, P.n_stft), np.nan, np.float32)
        lib.check(lib.dn_invmel(h, emu.ptr(emu.f32(mag)), emu.ptr(lin), 2, 3, None))
        ref_lin = np.maximum(np.einsum("km,btm->btk", np.linalg.pinv(fb.astype(np.float64).T), mag.astype(np.float64)), 0.0)
        assert np.isfinite(lin).all()
        assert np.abs(lin - ref_lin).max() <= 2e-4 * max(1.0, np.abs(ref_lin).max())
        lib.dn_dsp_destroy(h)


def _run_pipe(lib, dsp, m, schedule, B, n_hops, g, head_start=0, stream=False, s16=False, init=None, n_iter=6, depth=1, flush_after=None, split=None, P=P):
    """n_hops pipelined hops (frame mode or streaming mode) under one Griffin-Lim schedule; returns everything a hop leaves behind"""
    from audio_denoising_amd._lib import DN_GL_WAVE_PER_STREAM  # noqa: F401
    pipe = C.c_void_p()
    create = lib.dn_pipe_stream_create if stream else lib.dn_pipe_create
    lib.check(create(m, dsp, B, 0, C.byref(pipe)))
    lib.check(lib.dn_pipe_set_depth(pipe, depth))
    lib.check(lib.dn_pipe_set_gl_schedule(pipe, schedule))
    lib.check(lib.dn_pipe_set_head_start(pipe, head_start))
    if split is not None:
        lib.check(lib.dn_pipe_set_split(pipe, split))
    outs = []
    if not stream:
        hx = np.zeros((B, 17, P.num_compressed_bins), np.float32)
        frames = [emu.f32(g["signal"][:B, h * P.hop: h * P.hop + P.n_fft]) for h in range(n_hops)]
        outs = [np.zeros((B, P.n_fft), np.float32) for _ in range(n_hops)]
        for h in range(n_hops):
            lib.check(lib.dn_pipe_submit(pipe, emu.ptr(frames[h]), emu.ptr(hx), emu.ptr(outs[h]), None if init is None else emu.ptr(init[h]), 11, 3, n_iter, 0.99, None))
            if flush_after is not None and h == flush_after:
                lib.check(lib.dn_pipe_flush(pipe, n_iter, 0.99, None))       # a drain in the middle of the sequence
        lib.check(lib.dn_pipe_flush(pipe, n_iter, 0.99, None))
        outs.append(hx)
    else:
        dt = np.int16 if s16 else np.float32
        for h in range(n_hops + 1):
            x = g["signal"][:B, h * P.hop:(h + 1) * P.hop]
            hop_in = np.ascontiguousarray(np.clip(x * 32767.0, -32767, 32767).astype(np.int16)) if s16 else emu.f32(x)
            o = np.zeros((B, P.hop), dt)
            lib.check(lib.dn_pipe_stream_push(pipe, emu.ptr(hop_in), int(s16), emu.ptr(o), int(s16), None, 11, 3, n_iter, 0.99, None))
            outs.append(o)
        for _ in range(depth):
            o = np.zeros((B, P.hop), dt)
            lib.check(lib.dn_pipe_stream_flush(pipe, emu.ptr(o), int(s16), n_iter, 0.99, None))
            outs.append(o)
        ring, ola, hx = np.zeros((B, P.n_fft), np.float32), np.zeros((B, P.n_fft), np.float32), np.zeros((B, 17, P.num_compressed_bins), np.float32)
        lib.check(lib.dn_pipe_stream_get_state(pipe, emu.ptr(ring), emu.ptr(ola), emu.ptr(hx), None))
        outs += [ring, ola, hx]
    lib.dn_pipe_destroy(pipe)
    return outs


@pytest.mark.parametrize("case", ["frames+head_start", "stream+s16"])
def test_griffinlim_one_wavefront_per_stream_is_bit_identical_to_one_per_column(lib, dsp, case):
    """dn_pipe_set_gl_schedule: the wavefront-per-stream Griffin-Lim (four streams a workgroup, the three columns interleaved in one wave,
    overlap-add in registers) must reproduce the three-wave chain bit for bit -- frames, overlap-add lines, emitted hops, hx.  B = 5: a
    full workgroup and one with a single live wave; six iterations (the emulator runs a work-item per OS thread; the gpu tier runs 32)."""
    from audio_denoising_amd._lib import DN_GL_WAVE_PER_COLUMN, DN_GL_WAVE_PER_STREAM, DN_SPLIT_OFF, DN_SPLIT_ON
    sig = load_golden("stream_S.npz")["signal"]
    g = {"signal": np.concatenate([sig, 0.5 * sig[:1, ::-1]], axis=0)}          # a fifth stream
    B, n_hops = 5, 2
    m = make_model(lib, 5)
    kw = dict(stream=case.startswith("stream"), s16="s16" in case, head_start=4 if "head_start" in case else 0)
    if "init" in case:
        rg = np.random.default_rng(7)
        kw["init"] = [emu.f32(rg.random((B, 3, P.n_stft, 2))) for _ in range(n_hops)]
    a = _run_pipe(lib, dsp, m, DN_GL_WAVE_PER_COLUMN, B, n_hops, g, **kw)
    # (the streaming case as a SPLIT hop: the chains' launch, then the front halves' launch with four workgroups a CU and unstaged weights)
    b = _run_pipe(lib, dsp, m, DN_GL_WAVE_PER_STREAM, B, n_hops, g, split=DN_SPLIT_ON if kw["stream"] else DN_SPLIT_OFF, **kw)
    lib.dn_model_destroy(m)
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert sum(int(np.abs(x.astype(np.float64)).max() > 0) for x in a) >= 3           # (the comparison is of real output)


@pytest.mark.parametrize("depth", [2, 4])
def test_deep_pipe_runs_the_chain_in_segments_bit_identically(lib, dsp, depth):
    """dn_pipe_set_depth: a frame's Griffin-Lim chain is cut into `depth` segments that run in the launches after its submit (one wavefront per
    stream and segment, parked in HBM in between), `depth` hops of every stream in flight.  Frames, hx, overlap-add lines and emitted hops must
    equal the depth-1 pipe bit for bit -- the emitted hops `depth - 1` pushes later -- also with injected phases and a drain in mid-sequence.
    n_iter = 7 does not divide evenly (segments of 2/2/3, 1/2/2/2 ...); n_iter = 2 < depth leaves empty segments.  (Depth 3, 32 iterations and
    batch 256 run in the gpu tier; the emulator runs a work-item per OS thread.)"""
    from audio_denoising_amd._lib import DN_GL_WAVE_PER_COLUMN, DN_GL_AUTO, DN_SPLIT_ON
    sig = load_golden("stream_S.npz")["signal"]
    g = {"signal": sig}
    B, n_hops = 2, depth + 1
    m = make_model(lib, 5)
    rg = np.random.default_rng(17)
    init = [emu.f32(rg.random((B, 3, P.n_stft, 2))) for _ in range(n_hops)]
    for kw in (dict(n_iter=7 if depth == 2 else 3, init=init, flush_after=1),):       # (depth 4 with 3 iterations: an empty segment)
        a = _run_pipe(lib, dsp, m, DN_GL_WAVE_PER_COLUMN, B, n_hops, g, **kw)
        b = _run_pipe(lib, dsp, m, DN_GL_AUTO, B, n_hops, g, depth=depth, split=DN_SPLIT_ON if depth == 4 else None, **kw)   # (depth 4: as two launches per hop)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        assert np.abs(a[0]).max() > 0
    if depth != 2:          # (the streaming form at depth 3 and 4: gpu tier)
        lib.dn_model_destroy(m)
        return
    # streaming: the same samples, depth - 1 pushes later; the state after the drain is the same
    a = _run_pipe(lib, dsp, m, DN_GL_WAVE_PER_COLUMN, B, n_hops, g, stream=True, n_iter=5)
    b = _run_pipe(lib, dsp, m, DN_GL_AUTO, B, n_hops, g, stream=True, n_iter=5, depth=depth)          # (device-RNG phases)
    lib.dn_model_destroy(m)
    ea, eb = np.concatenate(a[:-3], axis=1), np.concatenate(b[:-3], axis=1)
    lag = (depth - 1) * P.hop
    assert eb.shape[1] == ea.shape[1] + lag
    assert np.array_equal(eb[:, lag:], ea) and not eb[:, :lag].any() and np.abs(ea).max() > 0
    for x, y in zip(a[-3:], b[-3:]):
        assert np.array_equal(x, y)


def _run_groups(lib, dsp, m, B, n_hops, g, H, sizes=None, stream=False, s16=False, init=None, n_iter=4, P=P):
    """the same hops as _run_pipe, submitted as groups (dn_pipe_set_group): `sizes` = hops per submit (frame mode; default H each)"""
    pipe = C.c_void_p()
    create = lib.dn_pipe_stream_create if stream else lib.dn_pipe_create
    lib.check(create(m, dsp, B, 0, C.byref(pipe)))
    lib.check(lib.dn_pipe_set_group(pipe, H))
    outs = []
    if not stream:
        hx = np.zeros((B, 17, P.num_compressed_bins), np.float32)
        frames = emu.f32(np.stack([g["signal"][:B, h * P.hop: h * P.hop + P.n_fft] for h in range(n_hops)]))
        out = np.zeros((n_hops, B, P.n_fft), np.float32)
        ia = None if init is None else emu.f32(np.stack(init))
        h = 0
        for k in (sizes or [H] * ((n_hops + H - 1) // H)):
            k = min(k, n_hops - h)
            if k == 0:
                break
            lib.check(lib.dn_pipe_submit_group(pipe, emu.ptr(frames[h:]), B * P.n_fft, emu.ptr(hx), emu.ptr(out[h:]), B * P.n_fft,
                                               None if ia is None else emu.ptr(ia[h:]), 0 if ia is None else ia[0].size, 11, 3, k, n_iter, 0.99, None))
            h += k
        assert h == n_hops
        lib.check(lib.dn_pipe_flush(pipe, n_iter, 0.99, None))
        outs = [out[i] for i in range(n_hops)] + [hx]
    else:
        dt = np.int16 if s16 else np.float32
        assert (n_hops + 1) % H == 0
        for h0 in range(0, n_hops + 1, H):
            x = np.stack([g["signal"][:B, h * P.hop:(h + 1) * P.hop] for h in range(h0, h0 + H)])
            hop_in = np.ascontiguousarray(np.clip(x * 32767.0, -32767, 32767).astype(np.int16)) if s16 else emu.f32(x)
            o = np.zeros((H, B, P.hop), dt)
            lib.check(lib.dn_pipe_stream_push_group(pipe, emu.ptr(hop_in), B * P.hop, int(s16), emu.ptr(o), B * P.hop, int(s16), None, 0, 11, 3, n_iter, 0.99, None))
            outs += [o[i] for i in range(H)]
        o = np.zeros((H, B, P.hop), dt)
        valid = C.c_int32(-1)
        lib.check(lib.dn_pipe_stream_flush_group(pipe, emu.ptr(o), B * P.hop, int(s16), C.byref(valid), None))
        outs += [o[i] for i in range(H)]
        ring, ola, hx = np.zeros((B, P.n_fft), np.float32), np.zeros((B, P.n_fft), np.float32), np.zeros((B, 17, P.num_compressed_bins), np.float32)
        lib.check(lib.dn_pipe_stream_get_state(pipe, emu.ptr(ring), emu.ptr(ola), emu.ptr(hx), None))
        outs += [valid.value, ring, ola, hx]
    lib.dn_pipe_destroy(pipe)
    return outs


@pytest.mark.parametrize("H", [2, 4])
def test_hop_groups_run_whole_chains_bit_identically(lib, dsp, H):
    """dn_pipe_set_group: a launch carries H consecutive hops of every stream (front halves in order, hx handed on) beside the WHOLE Griffin-Lim
    chains of the previous group, one wavefront each -- nothing is parked.  Frames and hx must equal the one-hop pipe's bit for bit (device-RNG
    and injected phases; full groups, a short last group, uneven submits, a single-hop submit through dn_pipe_submit's route)."""
    from audio_denoising_amd._lib import DN_GL_WAVE_PER_COLUMN
    g = {"signal": load_golden("stream_S.npz")["signal"]}
    B, n_hops = 2, 5
    m = make_model(lib, 5)
    rg = np.random.default_rng(23)
    init = [emu.f32(rg.random((B, 3, P.n_stft, 2))) for _ in range(n_hops)]
    # (the emulator runs a work-item per OS thread: H = 2 takes the device-RNG phases and full groups -- two streams a chain workgroup --, H = 4 the
    # injected phases and uneven submits; the gpu tier runs every combination at batch 256 and 32 iterations)
    kw, sizes = (dict(init=None), None) if H == 2 else (dict(init=init), [1, H, 1, H])
    a = _run_pipe(lib, dsp, m, DN_GL_WAVE_PER_COLUMN, B, n_hops, g, n_iter=4, **kw)
    b = _run_groups(lib, dsp, m, B, n_hops, g, H, sizes=sizes, n_iter=4, **kw)
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert np.abs(a[0]).max() > 0
    lib.dn_model_destroy(m)


@pytest.mark.parametrize("s16", [True])
def test_streaming_hop_groups_emit_the_one_hop_pipes_samples_later(lib, dsp, s16):
    """dn_pipe_stream_push_group: H hops in, H hops out per launch; the chains of one stream finish in the same launch and fold into its
    overlap-add line IN ORDER.  The emitted stream is the one-hop pipe's, H - 1 hops later (zeros first); ring, overlap-add line and hx after the
    drain are the same."""
    from audio_denoising_amd._lib import DN_GL_WAVE_PER_COLUMN
    g = {"signal": load_golden("stream_S.npz")["signal"]}
    H, B, n_hops = 3, 2, 5          # 6 pushes = two groups of three
    m = make_model(lib, 5)
    a = _run_pipe(lib, dsp, m, DN_GL_WAVE_PER_COLUMN, B, n_hops, g, stream=True, s16=s16, n_iter=3)
    b = _run_groups(lib, dsp, m, B, n_hops, g, H, stream=True, s16=s16, n_iter=3)
    lib.dn_model_destroy(m)
    ea = np.concatenate(a[:-3], axis=1)                      # n_hops + 1 pushes and one flush
    eb = np.concatenate(b[:-4], axis=1)                      # two groups and one flush group
    lag = (H - 1) * P.hop
    assert np.array_equal(eb[:, lag:lag + ea.shape[1]], ea) and not eb[:, :lag].any() and np.abs(ea).max() > 0
    assert not eb[:, lag + ea.shape[1]:].any()               # zero hops behind the drained frames
    assert b[-4] == H                                        # frames 2, 3, 4 were pending: three hops of the flush carry samples
    for x, y in zip(a[-3:], b[-3:]):
        assert np.array_equal(x, y)


def test_device_rng_is_philox4x32_10_and_matches_the_published_known_answers(lib, dsp):
    """Integer work, bit-exact: the device generator behind rand_init=True (app3.py:149-153) against an independent numpy restatement of
    Philox4x32-10 that is itself pinned by the three known-answer vectors Random123 publishes; the first of them is reachable through the
    public draw (seed 0, stream 0, column 0, bin 0).  Then: a launch with init_angles = NULL uses exactly this draw."""
    from oracle import philox_ref
    for ctr, key, out in philox_ref.KAT:
        w = philox_ref.philox4x32_10(*ctr, *key)
        assert tuple(int(x) for x in w) == out
    seed, sid0, B = 0, 0, 2
    ang = np.zeros((B, 3, P.n_stft, 2), np.float32)
    lib.check(lib.dn_griffinlim_draw_phases(dsp, seed, sid0, emu.ptr(ang), B, None))
    assert ang[0, 0, 0, 0] == np.float32((0x6627e8d5 >> 8) / 16777216.0) and ang[0, 0, 0, 1] == np.float32((0xe169c58d >> 8) / 16777216.0)
    for seed, sid0 in ((0, 0), (0x9E3779B97F4A7C15, (1 << 40) + 12345)):
        lib.check(lib.dn_griffinlim_draw_phases(dsp, seed, sid0, emu.ptr(ang), B, None))
        ref = philox_ref.draw_phases(seed, sid0, B, P.n_stft)
        got = (ang[..., 0] + 1j * ang[..., 1]).transpose(0, 2, 1)
        assert np.array_equal(got, ref)
        assert ang.min() >= 0.0 and ang.max() < 1.0
    mag = np.abs(np.random.default_rng(3).standard_normal((B, 3, P.n_stft))).astype(np.float32)
    a = np.zeros((B, P.n_fft), np.float32)
    b = np.zeros((B, P.n_fft), np.float32)
    lib.check(lib.dn_griffinlim(dsp, emu.ptr(mag), None, seed, sid0, None, emu.ptr(a), B, 3, 0.99, None))
    lib.check(lib.dn_griffinlim(dsp, emu.ptr(mag), emu.ptr(ang), 0, 0, None, emu.ptr(b), B, 3, 0.99, None))
    assert np.array_equal(a, b) and np.abs(a).max() > 0
