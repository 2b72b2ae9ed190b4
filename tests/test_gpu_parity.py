"""Parity tests proper (run on the MI355X box: ``pytest -m gpu``).  Everything goes through the
product path -- Python host -> ctypes -> libdn_denoise.so (HIP) -- and is compared with
  * golden vectors produced by the reference's own GRUUNet2 (model stage: PINNED), and
  * the CPU oracle (DSP stages: restatement of torchaudio, parity unpinned by the reference).
Tolerances are the north_star's: <= 1e-4 max-abs on the mel residual (predicted_diff_mel),
<= 1e-3 RMS on the reconstructed waveform given shared Griffin-Lim initial phases.
"""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu

TOL_RESIDUAL = 1e-4
TOL_WAVE_RMS = 1e-3
# Next to every RMS check there is a max-abs bound, so that a single bad sample per frame cannot hide in the mean: 32 Griffin-Lim
# iterations amplify fp32 rounding differences ~100x (1e-6 relative on the magnitudes -> 1e-4 on the waveform),
# measured max-abs on the MI355X is 1e-5 .. 3e-4 on signals of RMS 0.05 .. 0.1; the bound is 20x the RMS bar.
TOL_WAVE_MAX = 2e-2
# hx after a chain of hops fed by the fp32 DSP front end (the model input itself differs from the oracle's by ~2e-5, P5): measured
# <= 6e-5 after 30 chained steps (stream_S), bound 2e-4.  (The model stage alone holds 1e-4: test_gruunet2_chain_of_20_hops.)
TOL_HX_STREAM = 2e-4
# Guard bands next to the north_star bars: 10x what is measured on the MI355X (tools/parity_margins.py, profiles/r03_parity_margins.txt), so that a
# regression two orders below the bars still fails.  On the committed goldens (S / R2 / R1, 8 frames each, shared Griffin-Lim phases) the hop
# measures: residual <= 2.9e-6, hx <= 4.2e-7, waveform max-abs <= 2.0e-6, RMS <= 3.7e-7.
GUARD_RESIDUAL = 5e-5
GUARD_HX = 1e-5
GUARD_WAVE_MAX = 5e-5
GUARD_WAVE_RMS = 1e-5
# All 256 random frames of the metric's batch: residual 5.8e-6, hx 6.1e-7; the waveform error is NOT uniform over streams -- 32 Griffin-Lim iterations
# amplify rounding differences by a factor that depends on the frame (~100x typical), a few streams in 256 end two orders above
# the median -- measured over the batch RMS 5.2e-5 and max-abs 1.4e-3 (signal RMS 8.9e-3), median per-stream RMS ~1e-6.
GUARD_B256_WAVE_RMS = 5e-4
GUARD_B256_WAVE_MAX = 1e-2
GUARD_B256_STREAM_MEDIAN_RMS = 2e-5


def _wave_close(got, ref, scale=1.0):
    """RMS and max-abs waveform bounds together (numpy arrays)."""
    err = np.asarray(got, dtype=np.float64) - np.asarray(ref, dtype=np.float64)
    rms, mx = float(np.sqrt(np.mean(err ** 2))), float(np.abs(err).max())
    assert rms <= TOL_WAVE_RMS * scale and mx <= TOL_WAVE_MAX * scale, (rms, mx)
    return rms, mx


def _f64_frames(hops, inits, p, short="dari_tult", hx0=None, n_iter=32):
    """Float64 yardstick for a chain of hops (oracle/pipeline_np64.py: float64 numpy DSP on the fp32 window / filterbank constants, the model stage
    = oracle/model_ref.forward on float64 weights, which oracle/make_f64_golden.py checks against the reference's own class in double to 1e-12).
    hops: list of (B, n_fft) fp32 tensors, inits: list of (B, K, 3) complex64 -> (list of float64 waveforms, float64 hx)."""
    from oracle import dsp_ref, model_ref, pipeline_np64
    sd64 = {k: v.double() for k, v in _state_dict(short).items()}

    def model64(x, hx):
        with torch.no_grad():
            o, h = model_ref.forward(sd64, torch.from_numpy(x), torch.from_numpy(hx))
        return o.numpy(), h.numpy()
    fb = dsp_ref.melscale_fbanks(p.n_stft, p.n_mels, p.sample_rate).numpy()
    window = torch.hann_window(p.n_fft).numpy()
    B = hops[0].shape[0]
    hx = np.zeros((B, 17, p.num_compressed_bins)) if hx0 is None else hx0
    outs = []
    for f, ia in zip(hops, inits):
        r = pipeline_np64.process_frame64(f.numpy(), hx, model64, window, fb, ia.numpy(), p.n_fft, p.hop, n_iter=n_iter)
        hx = r["hx"]
        outs.append(r["out"])
    return outs, hx


# Random frames against the float64 yardstick: the batch-256 guard bands (profiles/r04_parity_margins.txt: the GPU's own error has the same heavy tail
# as the CPU oracle's -- it is the Griffin-Lim chain's amplification of ANY fp32 rounding -- so float64 does not allow tighter bands than these)
GUARD_F64_WAVE_RMS = GUARD_B256_WAVE_RMS
GUARD_F64_WAVE_MAX = GUARD_B256_WAVE_MAX
CFG = dict(in_size=1, hidden_sizes=(17, 17, 17, 17), kernel_sizes=(3, 3, 3, 3), strides=(2, 2, 2, 2), paddings=(1, 1, 1, 1), num_gaussians=6)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the GPU box"
    return torch.device("cuda:0")


def _state_dict(short):
    from oracle import model_ref
    return model_ref.unflatten_weights(np.fromfile(os.path.join(GOLDEN, f"weights_{short}.bin"), dtype=np.float32))


def _model(dev, C, short="dari_tult"):
    from gruunet2 import GRUUNet2          # the reference's import line (app3.py:38)
    m = GRUUNet2(num_compressed_bins=C, **CFG)
    m.load_state_dict(_state_dict(short))   # reference checkpoint keys load unchanged
    return m.eval().to(dev)


def test_native_library_is_the_in_tree_hip_build(dev):
    from audio_denoising_amd import _lib
    lib = _lib.get_lib()
    assert lib.path.endswith(os.path.join("audio-denoising_amd", "lib", "libdn_denoise.so"))
    with open("/proc/self/maps") as f:
        assert any("libdn_denoise.so" in ln for ln in f)


CELL_CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "cell_*_B*_T*_F*.npz")))


@pytest.mark.parametrize("name", CELL_CASES)
def test_gruunet2_forward_matches_reference_golden(dev, name):
    g = load_golden(name)
    F = g["x"].shape[2]
    m = _model(dev, F // 16, "dari_tult2" if "dari_tult2" in name else "dari_tult")
    with torch.no_grad():
        out, hx = m(torch.from_numpy(g["x"]).to(dev), torch.from_numpy(g["hx0"]).to(dev))
    assert out.shape == g["out"].shape and hx.shape == g["hx1"].shape
    assert np.abs(out.cpu().numpy() - g["out"]).max() <= TOL_RESIDUAL
    assert np.abs(hx.cpu().numpy() - g["hx1"]).max() <= TOL_RESIDUAL


def test_gruunet2_conventions_2d_input_default_hx_and_no_mutation(dev):
    g = load_golden("cell_dari_tult_conventions.npz")
    m = _model(dev, 4)
    o2, h2 = m(torch.from_numpy(g["x2"]).to(dev))                 # (T,F) input, hx=None   gruunet2.py:291-305
    assert o2.shape == (3, 64) and h2.shape == (1, 17, 4)
    assert np.abs(o2.cpu().numpy() - g["out2"]).max() <= TOL_RESIDUAL
    o3, h3 = m(torch.from_numpy(g["x3"]).to(dev))
    assert np.abs(o3.cpu().numpy() - g["out3"]).max() <= TOL_RESIDUAL and np.abs(h3.cpu().numpy() - g["hx3"]).max() <= TOL_RESIDUAL
    hx = torch.randn(2, 17, 4, device=dev)
    keep = hx.clone()
    m(torch.from_numpy(g["x3"]).to(dev), hx)
    assert torch.equal(hx, keep)                                  # forward does not mutate its hx argument
    with pytest.raises(RuntimeError):                             # 80 bins vs 4 compressed bins: shape error as in the reference
        m(torch.zeros(1, 3, 80, device=dev))
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 64))                                  # CPU tensor: no fallback


def test_gruunet2_chain_of_20_hops_and_time_split(dev):
    g = load_golden("cell_dari_tult_chain20_F80.npz")
    m = _model(dev, 5)
    hx = None
    for h in range(20):
        o, hx = m(torch.from_numpy(g["x"][h]).to(dev), hx)
        assert np.abs(o.cpu().numpy() - g["out"][h]).max() <= TOL_RESIDUAL
    assert np.abs(hx.cpu().numpy() - g["hx_final"]).max() <= TOL_RESIDUAL
    # forward(x[:, :2]) then forward(x[:, 2:], hx) == forward(x): bit-exact (same kernels, same order)
    x = torch.from_numpy(g["x"][0]).to(dev)
    full, hfull = m(x)
    a, ha = m(x[:, :2].contiguous())
    b, hb = m(x[:, 2:].contiguous(), ha)
    assert torch.equal(torch.cat([a, b], 1), full) and torch.equal(hb, hfull)


def test_reloading_weights_rebuilds_the_native_handle(dev):
    g1, g2 = load_golden("cell_dari_tult_B4_T3_F64.npz"), load_golden("cell_dari_tult2_B4_T3_F64.npz")
    m = _model(dev, 4, "dari_tult")
    x, h = torch.from_numpy(g1["x"]).to(dev), torch.from_numpy(g1["hx0"]).to(dev)
    assert np.abs(m(x, h)[0].cpu().numpy() - g1["out"]).max() <= TOL_RESIDUAL
    m.load_state_dict(_state_dict("dari_tult2"))
    assert np.abs(m(x, h)[0].cpu().numpy() - g2["out"]).max() <= TOL_RESIDUAL


# ------------------------------------------------------------------ transforms (DSP stages)
def _transforms(dev, p):
    from audio_denoising_amd import transforms as T
    return (T.Spectrogram(power=None, n_fft=p.n_fft, win_length=p.n_fft, hop_length=p.hop, window_fn=torch.hann_window).to(dev),
            T.MelScale(n_mels=p.n_mels, n_stft=p.n_stft, sample_rate=p.sample_rate).to(dev),
            T.InverseMelScale(n_mels=p.n_mels, n_stft=p.n_stft, sample_rate=p.sample_rate).to(dev),
            T.GriffinLim(n_fft=p.n_fft, win_length=p.n_fft, hop_length=p.hop, window_fn=torch.hann_window, power=1.0).to(dev))


def _params(tag):
    from oracle import pipeline_ref
    return {"S": pipeline_ref.PARAMS_S, "R1": pipeline_ref.PARAMS_R1, "R2": pipeline_ref.PARAMS_R2}[tag]


@pytest.mark.parametrize("tag", ["S", "R2", "R1"])
def test_transform_chain_as_the_app_calls_it(dev, tag):
    """app3.py:188-213 written with this package's transforms in place of torchaudio's (R1 = the app's own
    STFT_PARAMS, app3.py:29-33: n_fft 1536 -> the mixed-radix 768-point one-wave FFT)."""
    p = _params(tag)
    g = load_golden(f"dsp_{tag}.npz")
    T0, M0T, M0I, GL = _transforms(dev, p)
    assert torch.equal(M0T.fb.cpu(), torch.from_numpy(g["fb"]))          # filterbank is bit-identical to the oracle's
    frames = torch.from_numpy(g["frames"])
    peak = torch.from_numpy(g["peak"])
    windowed = (frames / peak[:, None]) * torch.hann_window(p.n_fft)     # P1, P2 done on the host as the app does
    spec = T0(windowed.to(dev))                                          # P4
    assert spec.shape == (frames.shape[0], p.n_stft, 3) and spec.dtype == torch.complex64
    from oracle import dsp_ref
    ref_spec = dsp_ref.spectrogram(windowed, p.n_fft, p.hop).numpy()
    assert np.abs(spec.cpu().numpy() - ref_spec).max() <= 2e-6 * np.abs(ref_spec).max() + 1e-6
    mel = M0T(spec.abs()).log1p()                                        # P5
    model_input = mel.transpose(-1, -2)                                  # P6
    assert np.abs(model_input.cpu().numpy() - g["model_input"]).max() <= 2e-5
    lin = torch.clamp(M0I(torch.from_numpy(g["mel_mag"]).to(dev)), min=0)   # P10
    assert np.abs(lin.cpu().numpy() - g["lin_mag"]).max() <= 2e-4 * max(1.0, float(np.abs(g["lin_mag"]).max()))
    y = GL(torch.from_numpy(g["lin_mag"]).to(dev), init_angles=torch.from_numpy(g["init_angles"]).to(dev))   # P11
    ref_y = g["out"] / g["peak"][:, None]
    _wave_close(y.cpu().numpy(), ref_y)


@pytest.mark.parametrize("tag", ["S", "R1"])
def test_inverse_mel_factored_and_dense_forms_agree(dev, tag):
    """A plan that builds the pseudo-inverse itself runs the inverse mel in factors (fb, banded (fb^T fb)^-1); a plan handed an explicit
    pseudo-inverse runs the dense contraction.  Same operator: both against the float64 least-squares oracle and against each other,
    at batch 256 with ragged tails."""
    from audio_denoising_amd import transforms as T
    from audio_denoising_amd.transforms import DspPlan
    from oracle import dsp_np64, dsp_ref
    p = _params(tag)
    fb = dsp_ref.melscale_fbanks(p.n_stft, p.n_mels, p.sample_rate)
    fac = DspPlan(dev, p.sample_rate, p.n_fft, p.hop, p.n_mels, fb=fb)
    _, pinv, _ = fac.tables()
    dense = DspPlan(dev, p.sample_rate, p.n_fft, p.hop, p.n_mels, fb=fb, pinv=pinv)
    g = torch.Generator().manual_seed(31)
    for rows in (256 * 3, 7):                                  # 7 rows: the last workgroup of three rows is ragged
        mel = (torch.rand(rows, p.n_mels, generator=g) * 20).to(dev)
        outs = []
        for plan in (fac, dense):
            lin = torch.empty(rows, p.n_stft, device=dev)
            plan.lib.check(plan.lib.dn_invmel(plan.handle, mel.data_ptr(), lin.data_ptr(), rows, 1, None))
            torch.cuda.synchronize()
            outs.append(lin.cpu().numpy())
        ref = dsp_np64.inverse_mel_scale(mel.cpu().numpy().reshape(rows, p.n_mels, 1), fb.numpy())[..., 0]
        scale = max(1.0, float(np.abs(ref).max()))
        assert np.abs(outs[0] - ref).max() <= 2e-5 * scale and np.abs(outs[1] - ref).max() <= 2e-5 * scale
        assert np.abs(outs[0] - outs[1]).max() <= 2e-5 * scale


def test_inverse_spectrogram_round_trip_and_linearity_at_full_batch(dev):
    """Size-independent properties at BASELINE's batch (256): istft(stft(x)) == x, STFT linearity, Parseval."""
    from audio_denoising_amd import transforms as T
    from oracle import pipeline_ref
    p = pipeline_ref.PARAMS_S
    T0 = T.Spectrogram(power=None, n_fft=p.n_fft, win_length=p.n_fft, hop_length=p.hop).to(dev)
    I0 = T.InverseSpectrogram(n_fft=p.n_fft, win_length=p.n_fft, hop_length=p.hop).to(dev)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(256, p.n_fft, generator=g).to(dev)
    y = torch.randn(256, p.n_fft, generator=g).to(dev)
    sx, sy = T0(x), T0(y)
    assert (I0(sx) - x).abs().max().item() <= 2e-5
    lin = T0(2.0 * x - 3.0 * y)
    assert (lin - (2.0 * sx - 3.0 * sy)).abs().max().item() <= 1e-4 * sx.abs().max().item()
    # Parseval on the centre column (window applied): sum |w x|^2 == (|X0|^2 + 2 sum |Xk|^2 + |Xn|^2) / N
    w = torch.hann_window(p.n_fft, device=dev)
    e_time = ((x * w) ** 2).sum(1)
    c = sx[:, :, 1].abs() ** 2
    e_freq = (c[:, 0] + 2 * c[:, 1:-1].sum(1) + c[:, -1]) / p.n_fft
    assert ((e_time - e_freq).abs() / e_time).max().item() <= 1e-5


def test_griffinlim_is_idempotent_on_a_consistent_spectrogram_and_seeded_rng_repeats(dev):
    from audio_denoising_amd import transforms as T
    from oracle import pipeline_ref
    p = pipeline_ref.PARAMS_S
    T0 = T.Spectrogram(power=None, n_fft=p.n_fft, win_length=p.n_fft, hop_length=p.hop).to(dev)
    GL = T.GriffinLim(n_fft=p.n_fft, win_length=p.n_fft, hop_length=p.hop, power=1.0).to(dev)
    g = torch.Generator().manual_seed(6)
    x = 0.1 * torch.randn(64, p.n_fft, generator=g).to(dev)
    s = T0(x)
    # start from the true phases: the magnitude is consistent, every iteration must reproduce x
    y = GL(s.abs(), init_angles=(s / (s.abs() + 1e-16)))
    assert (y - x).abs().max().item() <= 1e-4
    torch.manual_seed(11)
    a = GL(s.abs())
    torch.manual_seed(11)
    b = GL(s.abs())
    assert torch.equal(a, b) and not torch.equal(a, GL(s.abs()))


# ------------------------------------------------------------------ the fused hop
@pytest.mark.parametrize("tag", ["S", "R2", "R1"])
def test_process_frame_matches_oracle_golden(dev, tag):
    from audio_denoising_amd.pipeline import Denoiser
    p = _params(tag)
    g = load_golden(f"dsp_{tag}.npz")
    dn = Denoiser(_model(dev, p.num_compressed_bins), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    frames = torch.from_numpy(g["frames"]).to(dev)
    out, hx, resid = dn.process_frame(frames, None, init_angles=torch.from_numpy(g["init_angles"]).to(dev), return_residual=True)
    assert np.abs(resid.cpu().numpy() - g["predicted_diff"]).max() <= GUARD_RESIDUAL <= TOL_RESIDUAL
    assert np.abs(hx.cpu().numpy() - g["hx"]).max() <= GUARD_HX
    rms, mx = _wave_close(out.cpu().numpy(), g["out"])
    assert rms <= GUARD_WAVE_RMS and mx <= GUARD_WAVE_MAX, (rms, mx)
    # silent / sub-threshold streams (peak <= 1e-6 -> no normalisation, app3.py:182-186) stay finite and match the oracle
    assert torch.isfinite(out).all()
    assert np.abs(out.cpu().numpy()[4:6] - g["out"][4:6]).max() <= 1e-3


@pytest.mark.parametrize("n_mels", [16, 48])
def test_process_frame_at_other_filter_counts_matches_the_oracle(dev, n_mels):
    """Filter counts the goldens do not cover: 16 mels (bands of up to 142 bins: longer than the packed mel schedule takes, so the
    analysis falls back to a lane per filter; 1 compressed bin) and 48 (3 compressed bins; schedule).  Both run the model with run-time
    lengths (no compile-time bin count), the factored inverse mel at a different width, pipelined == unpipelined bit for bit."""
    from audio_denoising_amd.pipeline import Denoiser, HopPipeline
    from oracle import dsp_ref, pipeline_ref
    p = pipeline_ref.Params(16000, 1024, 512, n_mels)
    C = n_mels // 16
    dn = Denoiser(_model(dev, C), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    g = torch.Generator().manual_seed(77 + n_mels)
    frames = 0.1 * torch.randn(12, p.n_fft, generator=g)
    init = torch.rand(12, p.n_stft, 3, dtype=torch.complex64, generator=g)
    out, hx, resid = dn.process_frame(frames.to(dev), None, init_angles=init.to(dev), return_residual=True)
    fb = dsp_ref.melscale_fbanks(p.n_stft, p.n_mels, p.sample_rate)
    with torch.no_grad():
        ref = pipeline_ref.process_frame(_state_dict("dari_tult"), frames, torch.zeros(12, 17, C), p, fb, init_angles=init)
    assert (resid.cpu() - ref["predicted_diff"]).abs().max().item() <= TOL_RESIDUAL
    assert (hx.cpu() - ref["hx"]).abs().max().item() <= TOL_RESIDUAL
    _wave_close(out.cpu().numpy(), ref["out"].numpy())
    # the pipelined hop (256-thread inverse mel) against the unpipelined one (inverse mel as the Griffin-Lim prologue, 192 threads)
    fd = frames.to(dev)
    hs, hp = dn.init_hx(12), dn.init_hx(12)
    serial = [torch.empty_like(fd) for _ in range(3)]
    piped = [torch.empty_like(fd) for _ in range(3)]
    pipe = HopPipeline(dn, 12)
    for i in range(3):
        dn.process_frame_(fd, hs, serial[i], seed=5 + i, stream_id0=0)
        pipe.submit(fd, hp, piped[i], seed=5, stream_id0=0)
    pipe.flush()
    torch.cuda.synchronize()
    assert torch.equal(hs, hp)
    for a, b in zip(serial, piped):
        assert torch.equal(a, b)


def test_process_frame_full_batch_vs_oracle_sample_and_shard_invariance(dev):
    """BASELINE config 2: batch 256, S params.  EVERY stream is checked against the oracle (guard bands beside the north_star bars);
    the device-RNG path by the shard property (two half-batches with global stream ids == one batch)."""
    from audio_denoising_amd.pipeline import Denoiser
    from oracle import dsp_ref, pipeline_ref
    p = pipeline_ref.PARAMS_S
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    g = torch.Generator().manual_seed(1234)
    frames = 0.1 * torch.randn(256, p.n_fft, generator=g)
    init = torch.rand(256, p.n_stft, 3, dtype=torch.complex64, generator=torch.Generator().manual_seed(4321))
    out, hx, resid = dn.process_frame(frames.to(dev), None, init_angles=init.to(dev), return_residual=True)
    fb = dsp_ref.melscale_fbanks(p.n_stft, p.n_mels, p.sample_rate)
    with torch.no_grad():       # ALL 256 streams (the CPU oracle takes ~0.3 s for them)
        ref = pipeline_ref.process_frame(_state_dict("dari_tult"), frames, torch.zeros(256, 17, 5), p, fb, init_angles=init)
    assert (resid.cpu() - ref["predicted_diff"]).abs().max().item() <= GUARD_RESIDUAL
    assert (hx.cpu() - ref["hx"]).abs().max().item() <= GUARD_HX
    rms, mx = _wave_close(out.cpu().numpy(), ref["out"].numpy())
    per_stream = (out.cpu() - ref["out"]).pow(2).mean(dim=1).sqrt().numpy()
    print(f"batch 256, all streams: waveform RMS {rms:.2e} max-abs {mx:.2e}; per-stream RMS median {np.median(per_stream):.2e} "
          f"p90 {np.quantile(per_stream, 0.9):.2e} max {per_stream.max():.2e}")
    assert rms <= GUARD_B256_WAVE_RMS and mx <= GUARD_B256_WAVE_MAX and np.median(per_stream) <= GUARD_B256_STREAM_MEDIAN_RMS
    # device-RNG path: sharding must not change a single bit
    fd = frames.to(dev)
    whole, hw = dn.process_frame(fd, None, seed=99, stream_id0=0)
    lo, hl = dn.process_frame(fd[:128].contiguous(), None, seed=99, stream_id0=0)
    hi, hh = dn.process_frame(fd[128:].contiguous(), None, seed=99, stream_id0=128)
    assert torch.equal(torch.cat([lo, hi]), whole) and torch.equal(torch.cat([hl, hh]), hw)
    assert torch.isfinite(whole).all()


def test_batch_256_waveform_error_is_attributed_against_float64(dev):
    """Which side of the batch-256 comparison carries its long tail?  GPU and fp32 CPU oracle differ by up to 1.4e-3 on the worst of the 256
    metric frames (signal RMS 8.9e-3) because 32 Griffin-Lim iterations amplify rounding by a frame-dependent factor.  Here both are measured
    against the SAME algorithm in float64 (tests/golden/metric_f64_B256.npz: oracle/pipeline_np64.py with the model stage run by the
    reference's own class in double, oracle/make_f64_golden.py): per stream the GPU must be as close to the float64 result as the reference's
    own fp32 arithmetic is -- |gpu - f64| <= 2 |cpu_fp32 - f64| + a small floor, on the waveform (RMS and max-abs) and on the mel residual --
    and over the batch the GPU's error distribution must not sit above the CPU's."""
    from audio_denoising_amd.pipeline import Denoiser
    from oracle import dsp_ref, pipeline_ref
    p = pipeline_ref.PARAMS_S
    f64 = load_golden("metric_f64_B256.npz")
    g = torch.Generator().manual_seed(int(f64["frames_seed"]))
    frames = 0.1 * torch.randn(256, p.n_fft, generator=g)
    init = torch.rand(256, p.n_stft, 3, dtype=torch.complex64, generator=torch.Generator().manual_seed(int(f64["init_seed"])))
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    out, hx, resid = dn.process_frame(frames.to(dev), None, init_angles=init.to(dev), return_residual=True)
    fb = dsp_ref.melscale_fbanks(p.n_stft, p.n_mels, p.sample_rate)
    with torch.no_grad():
        ref = pipeline_ref.process_frame(_state_dict("dari_tult"), frames, torch.zeros(256, 17, 5), p, fb, init_angles=init)
    eg = out.cpu().numpy().astype(np.float64) - f64["out"]
    ec = ref["out"].numpy().astype(np.float64) - f64["out"]
    g_rms, c_rms = np.sqrt((eg ** 2).mean(axis=1)), np.sqrt((ec ** 2).mean(axis=1))
    g_max, c_max = np.abs(eg).max(axis=1), np.abs(ec).max(axis=1)
    rg = np.abs(resid.cpu().numpy() - f64["predicted_diff"]).reshape(256, -1).max(axis=1)
    rc = np.abs(ref["predicted_diff"].numpy() - f64["predicted_diff"]).reshape(256, -1).max(axis=1)
    q = lambda a: f"median {np.median(a):.2e} p90 {np.quantile(a, .9):.2e} max {a.max():.2e}"
    print(f"vs float64, per-stream waveform RMS: gpu {q(g_rms)} | cpu fp32 {q(c_rms)}")
    print(f"vs float64, per-stream waveform max-abs: gpu {q(g_max)} | cpu fp32 {q(c_max)}")
    print(f"vs float64, per-stream residual max-abs: gpu {q(rg)} | cpu fp32 {q(rc)}")
    # What can be asserted per stream and what cannot.  A frame's error is its rounding times the amplification of 32 Griffin-Lim iterations that
    # start from RANDOM phases, and that amplification is not a property of the frame alone: the float64 pipeline itself, fed the same frames
    # perturbed by 6e-8 relative (one fp32 rounding), moves by a per-stream amount that differs by up to 300x between two draws of the
    # perturbation (profiles/r04_parity_margins.txt).  So the per-stream RATIO of two fp32 implementations' errors is heavy-tailed in both
    # directions (measured 0.004 .. 290, median 1.06) and "gpu <= 2 x cpu for every stream" is not satisfiable by any implementation, the
    # reference's own included.  Asserted instead: the two error DISTRIBUTIONS over the 256 streams coincide -- every quantile of the GPU's
    # within 2x of the CPU's, about half of the streams on either side, the GPU's batch RMS not above the CPU's -- and the mel residual and hx,
    # which no chain amplifies, per stream.
    for e_g, e_c, floor in ((g_rms, c_rms, 2e-8), (g_max, c_max, 2e-7)):
        for x in (0.1, 0.25, 0.5, 0.75, 0.9, 0.99, 1.0):
            assert np.quantile(e_g, x) <= 2 * np.quantile(e_c, x) + floor, (x, np.quantile(e_g, x), np.quantile(e_c, x))
    worse = float((g_rms > c_rms).mean())
    assert 0.3 <= worse <= 0.7, worse
    assert np.sqrt((eg ** 2).mean()) <= 1.25 * np.sqrt((ec ** 2).mean())
    for x in (0.5, 0.9, 1.0):
        assert np.quantile(rg, x) <= 2 * np.quantile(rc, x), (x, np.quantile(rg, x), np.quantile(rc, x))
    assert np.abs(hx.cpu().numpy() - f64["hx"]).max() <= GUARD_HX and rg.max() <= GUARD_RESIDUAL          # (every stream, absolute)
    assert np.sqrt((eg ** 2).mean()) <= GUARD_B256_WAVE_RMS and g_max.max() <= GUARD_B256_WAVE_MAX and np.median(g_rms) <= GUARD_B256_STREAM_MEDIAN_RMS
    # Without the chain's amplification the arithmetic can be pinned per stream: two iterations (four transforms each way, the phase update with
    # momentum, the istft) against float64, every stream, at a band the tail cannot hide in
    dn2 = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels, n_iter=2)
    o2, _ = dn2.process_frame(frames.to(dev), None, init_angles=init.to(dev))
    y2 = _f64_frames([frames], [init], p, n_iter=2)[0][0]
    e2 = o2.cpu().numpy().astype(np.float64) - y2
    print(f"two iterations vs float64: per-stream max-abs {q(np.abs(e2).max(axis=1))}")
    assert np.abs(e2).max() <= GUARD_WAVE_MAX


@pytest.mark.parametrize("batch", [1, 3, 17, 255, 257, 1000])
def test_odd_batch_sizes_pipelined_equal_serial_and_match_the_oracle(dev, batch):
    """Batch sizes around the switches in the launch logic: a single stream, sizes that are no multiple of anything, 255 / 257 on either side of
    the head start's limit (on up to 256 streams, off above), more streams than CUs.  Pipelined hops == unpipelined hops bit for bit;
    the first and the last stream against the oracle."""
    from audio_denoising_amd.pipeline import Denoiser, HopPipeline
    from oracle import dsp_ref, pipeline_ref
    p = pipeline_ref.PARAMS_S
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    g = torch.Generator().manual_seed(1000 + batch)
    hops = [(0.1 * torch.randn(batch, p.n_fft, generator=g)) for _ in range(3)]
    hd = [h.to(dev) for h in hops]
    hs, hp = dn.init_hx(batch), dn.init_hx(batch)
    serial = [torch.empty(batch, p.n_fft, device=dev) for _ in hd]
    piped = [torch.empty(batch, p.n_fft, device=dev) for _ in hd]
    pipe = HopPipeline(dn, batch)
    for i, f in enumerate(hd):
        dn.process_frame_(f, hs, serial[i], seed=9 + i, stream_id0=3)
        pipe.submit(f, hp, piped[i], seed=9, stream_id0=3)
    pipe.flush()
    torch.cuda.synchronize()
    assert torch.equal(hs, hp)
    for a, b in zip(serial, piped):
        assert torch.equal(a, b) and torch.isfinite(a).all()
    # oracle, shared phases, first hop, streams 0 and batch - 1
    idx = torch.tensor(sorted({0, batch - 1}))
    init = torch.rand(batch, p.n_stft, 3, dtype=torch.complex64, generator=g)
    out, hx, resid = dn.process_frame(hd[0], None, init_angles=init.to(dev), return_residual=True)
    fb = dsp_ref.melscale_fbanks(p.n_stft, p.n_mels, p.sample_rate)
    with torch.no_grad():
        ref = pipeline_ref.process_frame(_state_dict("dari_tult"), hops[0][idx], torch.zeros(len(idx), 17, 5), p, fb, init_angles=init[idx])
    assert (resid.cpu()[idx] - ref["predicted_diff"]).abs().max().item() <= GUARD_RESIDUAL
    _wave_close(out.cpu()[idx].numpy(), ref["out"].numpy())
    # at the guard bands: against the float64 yardstick (two fp32 results of random frames differ by what Griffin-Lim makes of EITHER one's rounding)
    y64, _ = _f64_frames([hops[0][idx]], [init[idx]], p)
    e = out.cpu()[idx].numpy().astype(np.float64) - y64[0]
    assert np.sqrt((e ** 2).mean()) <= GUARD_F64_WAVE_RMS and np.abs(e).max() <= GUARD_F64_WAVE_MAX, (np.sqrt((e ** 2).mean()), np.abs(e).max())


def test_pipelined_hops_equal_serial_hops_bit_for_bit(dev):
    """dn_pipe_* runs hop n's Griffin-Lim blocks next to hop n+1's analysis+model blocks in one launch per hop; the
    results (8 chained hops, batch 256, device RNG) must equal the serial dn_process_frame sequence exactly."""
    from audio_denoising_amd.pipeline import Denoiser, HopPipeline
    from oracle import pipeline_ref
    p = pipeline_ref.PARAMS_S
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    g = torch.Generator().manual_seed(31)
    hops = [(0.1 * torch.randn(256, p.n_fft, generator=g)).to(dev) for _ in range(8)]
    hx_a = dn.init_hx(256)
    outs_a = [torch.empty(256, p.n_fft, device=dev) for _ in hops]
    for i, f in enumerate(hops):
        dn.process_frame_(f, hx_a, outs_a[i], seed=50 + i, stream_id0=7)
    hx_b = dn.init_hx(256)
    outs_b = [torch.empty(256, p.n_fft, device=dev) for _ in hops]
    pipe = HopPipeline(dn, 256)
    for i, f in enumerate(hops):
        pipe.submit(f, hx_b, outs_b[i], seed=50, stream_id0=7)          # frame i draws from seed + i
    pipe.flush()
    torch.cuda.synchronize()
    assert torch.equal(hx_a, hx_b)
    for a, b in zip(outs_a, outs_b):
        assert torch.equal(a, b)


def test_streaming_matches_oracle_golden(dev):
    """10 hops, 4 streams: ring buffer, hx carry and overlap-add (app3.py:178-226) against oracle/pipeline_ref.StreamRef."""
    from audio_denoising_amd.pipeline import Denoiser, DenoiserStream
    from oracle import pipeline_ref
    p = pipeline_ref.PARAMS_S
    g = load_golden("stream_S.npz")
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    st = DenoiserStream(dn, 4)
    sig = torch.from_numpy(g["signal"]).to(dev)
    inits = [torch.from_numpy(a).to(dev) for a in g["init_angles"]]
    # ragged arrival: the chunks do not line up with hops
    cuts = [0, 300, 1024, 1500, 3000, sig.shape[1]]
    outs = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        outs.append(st.push(sig[:, a:b].contiguous(), init_angles_per_hop=inits[st.hops:]))
    y = torch.cat(outs, 1).cpu().numpy()
    assert y.shape == g["out"].shape
    _wave_close(y, g["out"])
    _wave_close(st.ola.cpu().numpy(), g["ola"])
    assert np.abs(st.hx.cpu().numpy() - g["hx"]).max() <= TOL_HX_STREAM    # hx after 30 chained steps fed by the fp32 DSP front end
    assert st.push(torch.zeros(4, 0, device=dev)).shape == (4, 0)  # empty push: nothing emitted


def test_pipelined_stream_matches_oracle_golden_with_one_hop_delay(dev):
    """BASELINE config 5 path: state owned by the native pipe, one launch per hop (dn_pipe_stream_*)."""
    from audio_denoising_amd.pipeline import Denoiser, PipelinedStream
    from oracle import pipeline_ref
    p = pipeline_ref.PARAMS_S
    g = load_golden("stream_S.npz")
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    ps = PipelinedStream(dn, 4)
    sig = torch.from_numpy(g["signal"]).to(dev)
    inits = [torch.from_numpy(a).to(dev) for a in g["init_angles"]]
    n_frames = len(inits)
    outs = []
    for i in range(n_frames + 1):
        outs.append(ps.push(sig[:, i * p.hop:(i + 1) * p.hop].contiguous(), init_angles=inits[i - 1] if i >= 1 else None))
    outs.append(ps.flush())
    torch.cuda.synchronize()
    assert float(outs[0].abs().max()) == 0.0 and float(outs[1].abs().max()) == 0.0
    y = torch.cat(outs[2:], 1).cpu().numpy()
    assert y.shape == g["out"].shape
    _wave_close(y, g["out"])
    ring, ola, hx, frames_done = ps.state()
    assert frames_done == n_frames and ps.counters() == (n_frames + 1, n_frames, False)
    _wave_close(ola.cpu().numpy(), g["ola"])
    assert np.abs(hx.cpu().numpy() - g["hx"]).max() <= TOL_HX_STREAM
    assert torch.equal(ring.cpu(), torch.from_numpy(g["signal"])[:, -p.n_fft:])
    # checkpoint / resume of live streams: a second pipe restored from the snapshot continues identically
    ps2 = PipelinedStream(dn, 4)
    ps2.load_state(ring, ola, hx, frames_done)          # the resumed stream continues the seed sequence (frame index restored)
    nxt = (0.05 * torch.randn(4, p.hop, generator=torch.Generator().manual_seed(8))).to(dev)
    ps.seed, ps2.seed = 100, 100
    a1, b1 = ps.push(nxt), ps2.push(nxt)
    a2, b2 = ps.flush(), ps2.flush()
    assert torch.equal(a2, b2) and float(b1.abs().max()) == 0.0      # ps had nothing pending either: both emit ola[:hop] at flush
    # int16 PCM in / out (app3.py:168-172, 244-245)
    ps16 = PipelinedStream(dn, 4, seed=3)
    psf = PipelinedStream(dn, 4, seed=3)
    for i in range(4):
        q = torch.clamp(torch.round(sig[:, i * p.hop:(i + 1) * p.hop] * 3.0 * 32767), -32768, 32767).to(torch.int16).contiguous()
        o16 = ps16.push(q)
        # numpy's true division, as app3.py:172 (torch's GPU `tensor / python_scalar` multiplies by the reciprocal instead)
        of = psf.push(torch.from_numpy(q.cpu().numpy().astype(np.float32) / np.float32(32767)).to(dev))
        assert o16.dtype == torch.int16
        assert torch.equal(o16, (torch.clamp(of, -1, 1) * 32767).to(torch.int16))


def test_app_parameters_streaming_serial_and_pipelined(dev):
    """The reference app's own configuration (app3.py:13-33): 48 kHz, n_fft 1536, hop 768, 64 mels, checkpoint
    GRUUNet2-dari_tult2 -- streamed through DenoiserStream (serial hops) and PipelinedStream (one launch per hop)."""
    from audio_denoising_amd.pipeline import Denoiser, DenoiserStream, PipelinedStream
    p = _params("R1")
    g = load_golden("stream_R1.npz")
    dn = Denoiser(_model(dev, 4, "dari_tult2"), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    sig = torch.from_numpy(g["signal"]).to(dev)
    inits = [torch.from_numpy(a).to(dev) for a in g["init_angles"]]
    st = DenoiserStream(dn, 3)
    y = st.push(sig, init_angles_per_hop=inits).cpu().numpy()
    assert y.shape == g["out"].shape
    _wave_close(y, g["out"])
    assert np.abs(st.hx.cpu().numpy() - g["hx"]).max() <= TOL_HX_STREAM
    ps = PipelinedStream(dn, 3)
    outs = [ps.push(sig[:, i * p.hop:(i + 1) * p.hop].contiguous(), init_angles=inits[i - 1] if i >= 1 else None) for i in range(len(inits) + 1)]
    outs.append(ps.flush())
    y2 = torch.cat(outs[2:], 1).cpu().numpy()
    _wave_close(y2, g["out"])
    # size-independent property at the app's parameters: istft(stft(x)) == x for a full batch
    from audio_denoising_amd import transforms as T
    T0 = T.Spectrogram(power=None, n_fft=p.n_fft, win_length=p.n_fft, hop_length=p.hop).to(dev)
    I0 = T.InverseSpectrogram(n_fft=p.n_fft, win_length=p.n_fft, hop_length=p.hop).to(dev)
    x = torch.randn(256, p.n_fft, generator=torch.Generator().manual_seed(3)).to(dev)
    assert (I0(T0(x)) - x).abs().max().item() <= 3e-5


def test_empty_batch_is_a_no_op(dev):
    from audio_denoising_amd.pipeline import Denoiser
    p = _params("S")
    m = _model(dev, 5)
    out, hx = m(torch.zeros(0, 3, 80, device=dev))
    assert out.shape == (0, 3, 80) and hx.shape == (0, 17, 5)
    dn = Denoiser(m, p.sample_rate, p.n_fft, p.hop, p.n_mels)
    o, h = dn.process_frame(torch.zeros(0, p.n_fft, device=dev))
    assert o.shape == (0, p.n_fft) and h.shape == (0, 17, 5)


def test_one_model_shared_by_concurrent_threads(dev):
    """The reference shares ONE model object between all WebRTC worker threads (st.cache_resource, app3.py:46,499-505),
    each session with private hx/buffers.  Handles are immutable, so concurrent forwards from threads on their own
    HIP streams must give each thread exactly its single-threaded result."""
    import threading
    g = load_golden("cell_dari_tult_chain20_F80.npz")
    m = _model(dev, 5)
    xs = torch.from_numpy(g["x"]).to(dev)                       # (20, 8, 3, 80)
    m(xs[0])                                                    # build the native handle once
    results, errors = {}, []

    def worker(tid):
        try:
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                hx, outs = None, []
                for h in range(20):
                    o, hx = m(xs[h, tid:tid + 2].contiguous(), hx)
                    outs.append(o)
                st.synchronize()
            results[tid] = (torch.stack(outs).cpu().numpy(), hx.cpu().numpy())
        except Exception as e:          # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in (0, 2, 4, 6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for tid, (o, hx) in results.items():
        assert np.abs(o - g["out"][:, tid:tid + 2]).max() <= TOL_RESIDUAL
        assert np.abs(hx - g["hx_final"][tid:tid + 2]).max() <= TOL_RESIDUAL


def test_pipelined_hops_are_graph_capturable(dev):
    """The pipelined hop is one plain kernel launch on the caller's stream (no allocation, no sync, no events) whose
    hop-to-hop state lives on the device: ONE captured submit replays as consecutive hops (frame mode, batch 64)."""
    from audio_denoising_amd.pipeline import Denoiser, HopPipeline
    p = _params("S")
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    B, n = 64, 5
    gen = torch.Generator().manual_seed(17)
    frames = [(0.1 * torch.randn(B, p.n_fft, generator=gen)).to(dev) for _ in range(n)]
    outs = [torch.empty(B, p.n_fft, device=dev) for _ in range(n)]
    hx = dn.init_hx(B)
    pipe = HopPipeline(dn, B)
    for i in range(n):
        pipe.submit(frames[i], hx, outs[i], seed=40, stream_id0=0)
    pipe.flush()
    torch.cuda.synchronize()
    # the same five hops as replays of ONE captured submit: fixed input / output buffers, rewritten / read between replays
    f_buf, o_buf, hx2 = torch.empty(B, p.n_fft, device=dev), torch.empty(B, p.n_fft, device=dev), dn.init_hx(B)
    pipe2 = HopPipeline(dn, B)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        pipe2.submit(f_buf, hx2, o_buf, seed=40, stream_id0=0, check_weights=False)
    got = []
    for i in range(n):
        f_buf.copy_(frames[i])
        graph.replay()
        if i >= 1:
            got.append(o_buf.clone())           # the replay that takes hop i completes hop i-1
    pipe2.flush()
    got.append(o_buf.clone())
    torch.cuda.synchronize()
    assert torch.equal(hx, hx2) and pipe2.counters() == (n, n, False)
    for a, b in zip(outs, got):
        assert torch.equal(a, b)


def test_captured_streaming_step_replays_at_1024_streams(dev):
    """BASELINE config 5 in its stated form (per GPU): 1,024 streams in streaming mode -- pipe-owned ring / overlap-add / hx,
    persistent across hops -- driven by ONE hipGraph-captured push that is replayed for every hop (app3.py:167-250 steady state).
    The replays must equal eager pushes bit for bit (outputs, overlap-add lines, hx) and match the oracle golden (the 4 golden
    streams tiled 256 times, shared initial phases) within the waveform tolerance."""
    from audio_denoising_amd.pipeline import Denoiser, PipelinedStream
    p = _params("S")
    g = load_golden("stream_S.npz")
    B, rep = 1024, 256
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    sig = torch.from_numpy(g["signal"]).to(dev).repeat(rep, 1)                               # (1024, n_fft + 9 hops)
    inits = [torch.from_numpy(a).to(dev).repeat(rep, 1, 1) for a in g["init_angles"]]       # per frame (1024, K, 3)
    n_frames = len(inits)
    # eager pushes
    pa = PipelinedStream(dn, B, seed=5)
    eager = [pa.push(sig[:, i * p.hop:(i + 1) * p.hop].contiguous(), init_angles=inits[i - 1] if i >= 1 else None) for i in range(n_frames + 1)]
    eager.append(pa.flush())
    # one captured push, replayed
    pb = PipelinedStream(dn, B, seed=5)
    hop_buf = torch.zeros(B, p.hop, device=dev)
    out_buf = torch.zeros(B, p.hop, device=dev)
    init_buf = inits[0].clone()
    torch.cuda.synchronize()
    graph = pb.graph_step(hop_buf, out_buf, init_angles=init_buf)
    replayed = []
    for i in range(n_frames + 1):
        hop_buf.copy_(sig[:, i * p.hop:(i + 1) * p.hop])
        if i >= 1:
            init_buf.copy_(inits[i - 1])
        graph.replay()
        replayed.append(out_buf.clone())
    replayed.append(pb.flush())
    torch.cuda.synchronize()
    assert pb.counters() == (n_frames + 1, n_frames, False)
    for a, b in zip(eager, replayed):
        assert torch.equal(a, b)
    for x, y in zip(pa.state()[:3], pb.state()[:3]):
        assert torch.equal(x, y)
    y = torch.cat(replayed[2:], 1).cpu().numpy().reshape(rep, 4, -1)
    err = y - g["out"][None]
    assert np.sqrt(np.mean(err ** 2)) <= TOL_WAVE_RMS and np.abs(err).max() <= TOL_WAVE_MAX
    assert np.array_equal(y[0], y[rep - 1])                                                # copies of a stream agree exactly
    # device-RNG phases under replay: 12 more replays continue the seed sequence exactly as eager pushes do
    nxt = (0.05 * torch.randn(B, p.hop, generator=torch.Generator().manual_seed(8))).to(dev)
    pc, pd = PipelinedStream(dn, B, seed=77), PipelinedStream(dn, B, seed=77)
    g2 = pd.graph_step(hop_buf, out_buf)
    for i in range(12):
        e = pc.push(nxt * (i + 1))
        hop_buf.copy_(nxt * (i + 1))
        g2.replay()
        assert torch.equal(e, out_buf)
    assert torch.equal(pc.flush(), pd.flush())


def test_reloading_weights_between_pipelined_hops(dev):
    """A pipe keeps launching with device memory it holds a reference on, and follows the model object: weights replaced
    between two hops (load_state_dict) are used from the next hop on, exactly as the unpipelined path does."""
    from audio_denoising_amd.pipeline import Denoiser, HopPipeline, PipelinedStream
    p = _params("R2")
    m = _model(dev, 4, "dari_tult")
    dn = Denoiser(m, p.sample_rate, p.n_fft, p.hop, p.n_mels)
    B = 8
    gen = torch.Generator().manual_seed(3)
    frames = [(0.1 * torch.randn(B, p.n_fft, generator=gen)).to(dev) for _ in range(4)]
    hx_a, hx_b = dn.init_hx(B), dn.init_hx(B)
    outs_a = [torch.empty(B, p.n_fft, device=dev) for _ in range(4)]
    outs_b = [torch.empty(B, p.n_fft, device=dev) for _ in range(4)]
    pipe = HopPipeline(dn, B)
    ps = PipelinedStream(dn, B)
    for i in range(4):
        if i == 2:
            torch.cuda.synchronize()
            m.load_state_dict(_state_dict("dari_tult2"))      # replaces the native handle (the old one is released)
            m(torch.zeros(1, 3, 64, device=dev))               # ... and forces the rebuild before the pipes launch again
        dn.process_frame_(frames[i], hx_a, outs_a[i], seed=9 + i, stream_id0=0)
        pipe.submit(frames[i], hx_b, outs_b[i], seed=9, stream_id0=0)
        ps.push(frames[i][:, :p.hop].contiguous())
    pipe.flush()
    ps.flush()
    torch.cuda.synchronize()
    assert torch.equal(hx_a, hx_b)
    for a, b in zip(outs_a, outs_b):
        assert torch.equal(a, b)
    # and it really switched: the same hops with the first weights throughout give something else
    m.load_state_dict(_state_dict("dari_tult"))
    hx_c, out_c = dn.init_hx(B), torch.empty(B, p.n_fft, device=dev)
    for i in range(4):
        dn.process_frame_(frames[i], hx_c, out_c, seed=9 + i, stream_id0=0)
    assert not torch.equal(hx_c, hx_a)


@pytest.mark.parametrize("name", ["cell_dari_tult_B256_T3_F80.npz", "cell_dari_tult_B256_T3_F64.npz", "cell_dari_tult2_B3_T7_F80.npz"])
def test_config3_bf16_mfma_conv_tiles_restated_tolerance(dev, name):
    """BASELINE config 3: batch 256, GRUUNet2 with bf16 MFMA conv tiles (v_mfma_f32_16x16x32_bf16, fp32 accumulate).
    Tolerance restated on the mel residual against the reference's fp32 golden: relative RMS <= 1e-2 and max-abs <= 5e-1
    (outputs span +-6).  Measured on the MI355X (tools/bf16_error_probe.py): rel-RMS 3.7e-3 .. 4.2e-3 on every case --
    the 2.9e-3 .. 4.4e-3 SURVEY.md 8d predicts for bf16-rounded conv inputs + weights -- and max-abs 0.07 .. 0.31 (the tail
    of 49k outputs at batch 256 with random non-zero hx; 0.15 on the 20-hop chain that starts from hx = 0)."""
    g = load_golden(name)
    F = g["x"].shape[2]
    m = _model(dev, F // 16, "dari_tult2" if "dari_tult2" in name else "dari_tult")
    m.conv_precision = "bf16"
    out, hx = m(torch.from_numpy(g["x"]).to(dev), torch.from_numpy(g["hx0"]).to(dev))
    err = out.cpu().numpy() - g["out"]
    assert np.abs(err).max() <= 5e-1
    assert np.sqrt(np.mean(err ** 2)) / np.sqrt(np.mean(g["out"] ** 2)) <= 1e-2
    assert np.abs(hx.cpu().numpy() - g["hx1"]).max() <= 5e-2
    assert np.abs(err).max() > 1e-5                       # really the reduced-precision path
    m.conv_precision = "fp32"
    out32, _ = m(torch.from_numpy(g["x"]).to(dev), torch.from_numpy(g["hx0"]).to(dev))
    assert np.abs(out32.cpu().numpy() - g["out"]).max() <= TOL_RESIDUAL


def test_server_variant_matches_oracle_golden(dev):
    """The socket server's request loop (server.py:199-217): R2 parameters, checkpoint GRUUNet2-good, three consecutive
    chunks with hx carried (and decayed by 0.9), resynthesis with the noisy phase."""
    from audio_denoising_amd.pipeline import ServerDenoiser
    g = load_golden("server_R2.npz")
    p = _params("R2")
    sd = ServerDenoiser(_model(dev, 4, "good"), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    hx = None
    for c in range(3):
        wave, hx = sd.process(torch.from_numpy(g["chunks"][c]).to(dev), hx)
        assert wave.shape == g["out"][c].shape
        _wave_close(wave.cpu().numpy(), g["out"][c])
    assert np.abs(hx.cpu().numpy() - g["hx"]).max() <= TOL_HX_STREAM
    # ragged: a chunk that is not a multiple of the hop (the client of server.py sends e.g. 4800 samples)
    x = torch.randn(2, 4800, generator=torch.Generator().manual_seed(2)).to(dev) * 0.1
    w, _ = sd.process(x, None)
    assert w.shape == (2, p.hop * (4800 // p.hop)) and torch.isfinite(w).all()
    from oracle import model_ref, server_ref
    ref = server_ref.process_chunk(_state_dict("good"), x.cpu(), None, p)
    _wave_close(w.cpu().numpy(), ref["out"].numpy())


def test_maximum_batch_streams_are_independent(dev):
    """BASELINE config 5's total (8192 streams) on ONE GPU: every stream's result must equal what it gets in a small
    batch (streams are independent; the device RNG is keyed by global stream id) -- checked on a spread of streams."""
    from audio_denoising_amd.pipeline import Denoiser, HopPipeline
    p = _params("S")
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    B = 8192
    gen = torch.Generator().manual_seed(23)
    frames = [(0.1 * torch.randn(B, p.n_fft, generator=gen)).to(dev) for _ in range(2)]
    outs = [torch.empty(B, p.n_fft, device=dev) for _ in range(2)]
    hx = dn.init_hx(B)
    pipe = HopPipeline(dn, B)
    for i in range(2):
        pipe.submit(frames[i], hx, outs[i], seed=9, stream_id0=0)
    pipe.flush()
    torch.cuda.synchronize()
    assert all(torch.isfinite(o).all() for o in outs)
    for lo in (0, 4096, 8192 - 64):                      # three 64-stream slices, re-run alone with their global ids
        hx_s = dn.init_hx(64)
        small = [torch.empty(64, p.n_fft, device=dev) for _ in range(2)]
        ps = HopPipeline(dn, 64)
        for i in range(2):
            ps.submit(frames[i][lo:lo + 64].contiguous(), hx_s, small[i], seed=9, stream_id0=lo)
        ps.flush()
        torch.cuda.synchronize()
        for i in range(2):
            assert torch.equal(small[i], outs[i][lo:lo + 64])
        assert torch.equal(hx_s, hx[lo:lo + 64])


@pytest.mark.parametrize("tag", ["S", "R1"])
def test_special_signals_match_oracle(dev, tag):
    """Edge-case frames through the whole hop vs the oracle: impulses at the frame edges and centre, DC, the Nyquist
    alternation, a bin-centred sine, sub-threshold and very large amplitudes, exact silence (app3.py:181-186 branches)."""
    from audio_denoising_amd.pipeline import Denoiser
    from oracle import dsp_ref, pipeline_ref
    p = _params(tag)
    N = p.n_fft
    n = torch.arange(N, dtype=torch.float32)
    frames = torch.zeros(10, N)
    frames[0, 0] = 1.0
    frames[1, N - 1] = -0.5
    frames[2, N // 2] = 0.25
    frames[3] = 0.3                                            # DC
    frames[4] = 0.2 * (1 - 2 * (n % 2))                        # Nyquist alternation
    frames[5] = 0.7 * torch.sin(2 * torch.pi * 37 * n / N)     # bin-centred tone
    frames[6] = 5e-7 * torch.sin(2 * torch.pi * 5 * n / N)     # below the 1e-6 peak threshold: not normalised
    frames[7] = 1e4 * torch.randn(N, generator=torch.Generator().manual_seed(1))
    frames[8] = 0.0                                            # silence
    frames[9] = 2e-6 * torch.randn(N, generator=torch.Generator().manual_seed(2))   # just above the threshold
    C = p.num_compressed_bins
    short = "dari_tult2" if tag == "R1" else "dari_tult"
    dn = Denoiser(_model(dev, C, short), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    init = torch.rand(10, p.n_stft, 3, dtype=torch.complex64, generator=torch.Generator().manual_seed(4321))
    out, hx, resid = dn.process_frame(frames.to(dev), None, init_angles=init.to(dev), return_residual=True)
    fb = dsp_ref.melscale_fbanks(p.n_stft, p.n_mels, p.sample_rate)
    with torch.no_grad():
        ref = pipeline_ref.process_frame(_state_dict(short), frames, torch.zeros(10, 17, C), p, fb, init_angles=init)
    assert torch.isfinite(out).all()
    assert (resid.cpu() - ref["predicted_diff"]).abs().max().item() <= TOL_RESIDUAL
    assert (hx.cpu() - ref["hx"]).abs().max().item() <= TOL_RESIDUAL
    # waveform: per-frame RMS error relative to that frame's scale (outputs are multiplied back by the frame's peak)
    scale = torch.maximum(ref["peak"], torch.tensor(1.0))
    rel = (out.cpu() - ref["out"]) / scale[:, None]
    err = rel.pow(2).mean(1).sqrt()
    assert err.max().item() <= TOL_WAVE_RMS and rel.abs().max().item() <= TOL_WAVE_MAX, (err, rel.abs().max())


def test_c_abi_host_without_python_gives_the_same_samples(dev, tmp_path):
    """examples/denoise_hop.cpp drives the path through include/dn_denoise.h only (no Python, no torch): built here with
    hipcc, run as a child process, its output must equal the Python host's bit for bit (same seeds, device RNG)."""
    import math
    import shutil
    import subprocess
    from audio_denoising_amd import _lib
    from audio_denoising_amd.pipeline import Denoiser
    from conftest import REPO
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe, outf = str(tmp_path / "denoise_hop"), str(tmp_path / "out.f32")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run([hipcc, "-O2", os.path.join(REPO, "examples", "denoise_hop.cpp"), "-I", os.path.join(REPO, "include"), "-L", libdir,
                    "-ldn_denoise", f"-Wl,-rpath,{libdir}", "-o", exe], check=True)
    B, hops = 4, 3
    r = subprocess.run([exe, os.path.join(GOLDEN, "weights_dari_tult.bin"), outf, str(B), str(hops)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(outf, dtype=np.float32).reshape(B, 1024)
    # the same thing through the Python host with the library's NATIVE tables (fb/window NULL) to match the C program
    p = _params("S")
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    from audio_denoising_amd.transforms import DspPlan
    dn.plan = DspPlan(dev, p.sample_rate, p.n_fft, p.hop, p.n_mels)          # native HTK filterbank / Hann
    hx = dn.init_hx(B)
    out = torch.empty(B, 1024, device=dev)
    for hop in range(hops):
        h = np.zeros((B, 1024), np.float32)
        for b in range(B):
            t = (hop * 512 + np.arange(1024)) / 16000.0
            h[b] = (0.3 * np.sin(2 * math.pi * (220.0 + 110.0 * b) * t) + 0.05 * np.sin(2 * math.pi * 3300.0 * t + b)).astype(np.float32)
        dn.process_frame_(torch.from_numpy(h).to(dev), hx, out, seed=2024 + hop, stream_id0=0)
    torch.cuda.synchronize()
    assert np.array_equal(got, out.cpu().numpy())
    # the same hops in groups of two per launch (dn_pipe_set_group / dn_pipe_submit_group through the C ABI): the same bits again
    r = subprocess.run([exe, os.path.join(GOLDEN, "weights_dari_tult.bin"), outf, str(B), str(hops), "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(np.fromfile(outf, dtype=np.float32).reshape(B, 1024), got)


def test_bench_two_ranks_on_one_gpu(dev):
    """bench.py's multi-rank path with the real kernels: `python bench.py --gpus 2` self-launches two ranks (sharing this box's
    one GPU, gloo for the collectives because RCCL refuses two ranks on one device), each runs its 256-stream shard, and
    rank 0 reports both ingress variants.  The RCCL path itself is exercised by the driver's multi-GPU run."""
    import json
    import subprocess
    import sys
    from conftest import REPO
    env = dict(os.environ, DN_DIST_BACKEND="gloo", DN_ALLOW_SHARED_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["data"] == "synthetic" and d["config"]["frames_per_step"] == 512
    assert d["ingress_variant"]["ingress"] == "scatter_gather" and d["ingress_variant"]["root_output_finite"] is True
    assert d["roofline"]["frac"] > 0 and d["value"] > 0


def test_bench_one_rank_over_rccl_runs_every_collective_call(dev):
    """The RCCL branch of bench.py on this one-GPU box: ONE rank under torch.distributed.run with the `nccl` backend (DN_BENCH_FORCE_DIST=1 keeps the
    process group although the world is 1): init_process_group(device_id=...), the rank-count all_reduce, barrier(device_ids=...), the MAX reduction of
    the timing and the ingress loop's scatter_rows / gather_rows calls on a second HIP stream -- everything the multi-GPU run executes except a
    transfer between two devices, which needs the driver's node (RCCL refuses two ranks on one device)."""
    import json
    import socket
    import subprocess
    import sys
    from conftest import REPO
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, DN_DIST_BACKEND="nccl", DN_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "8", "--warmup", "4", "--no-cpu-baseline", "--no-extras"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["ranks"]["backend"] == "nccl" and d["value"] > 0
    iv = d["ingress_variant"]
    assert "error" not in iv, iv
    assert d["ingress_ok"] is True and iv["backend"] == "nccl" and iv["hops_per_launch"] == 4 and iv["root_output_finite"] is True


def test_real_clip_3s_batch1_streams_match_oracle_golden(dev):
    """BASELINE configs[0] at its stated size: ONE 3 s / 16 kHz clip (93 hops, batch 1, dari_tult weights) of real audio from the
    reference's data tree (tests/golden/clip_S.npz, made by oracle/make_clip_golden.py), streamed hop by hop as the app's recv()
    does -- through DenoiserStream (no added latency), PipelinedStream (one hop later) and the int16 transport of app3.py:168-172."""
    from audio_denoising_amd.pipeline import Denoiser, DenoiserStream, PipelinedStream
    from oracle.make_clip_golden import N_HOPS, clip_init_angles
    p = _params("S")
    g = load_golden("clip_S.npz")
    s16 = torch.from_numpy(g["signal_s16"])[None]
    sig = torch.from_numpy(g["signal_s16"].astype(np.float32) / np.float32(32767))[None].to(dev)       # app3.py:172
    inits = [clip_init_angles(f).to(dev) for f in range(N_HOPS)]
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    ref_rms = float(np.sqrt(np.mean(g["out"] ** 2)))
    # (a) recv()-sized chunks of 480 samples (30 ms WebRTC frames) through the unpipelined stream
    st = DenoiserStream(dn, 1)
    outs = []
    for a in range(0, sig.shape[1], 480):
        outs.append(st.push(sig[:, a:a + 480].contiguous(), init_angles_per_hop=inits[st.hops:]))
    y = torch.cat(outs, 1).cpu().numpy()
    assert y.shape == g["out"].shape == (1, N_HOPS * p.hop)
    rms, mx = _wave_close(y, g["out"])
    assert rms <= 2e-2 * ref_rms                         # and relative to the (quiet) output itself
    _wave_close(st.ola.cpu().numpy(), g["ola"])
    assert np.abs(st.hx.cpu().numpy() - g["hx"]).max() <= TOL_HX_STREAM
    # (b) one launch per hop, float samples
    ps = PipelinedStream(dn, 1)
    outs = [ps.push(sig[:, i * p.hop:(i + 1) * p.hop].contiguous(), init_angles=inits[i - 1] if i >= 1 else None) for i in range(N_HOPS + 1)]
    outs.append(ps.flush())
    y2 = torch.cat(outs[2:], 1).cpu().numpy()
    _wave_close(y2, g["out"])
    assert np.array_equal(y2, y)                         # the two schedules run the same arithmetic
    # (c) int16 PCM in, int16 PCM out
    ps16 = PipelinedStream(dn, 1)
    q = s16.to(dev)
    outs = [ps16.push(q[:, i * p.hop:(i + 1) * p.hop].contiguous(), init_angles=inits[i - 1] if i >= 1 else None) for i in range(N_HOPS + 1)]
    outs.append(ps16.flush(s16=True))
    y16 = torch.cat(outs[2:], 1).cpu().numpy()
    want = (np.clip(g["out"], -1, 1) * 32767).astype(np.int16)
    assert y16.dtype == np.int16 and np.abs(y16.astype(np.int32) - want.astype(np.int32)).max() <= int(TOL_WAVE_MAX * 32767) + 1


def test_config3_bf16_conv_tiles_inside_the_whole_hop_batch256(dev):
    """BASELINE config 3 as a whole hop: batch 256, S parameters, UNet convs on bf16 MFMA tiles (DN_CONV_BF16 /
    GRUUNet2.conv_precision = "bf16") in the one-launch hop and in the software-pipelined hop.  Tolerance restated against the
    ORACLE (fp32 reference arithmetic): mel residual rel-RMS <= 1e-2 and max-abs <= 5e-1 (as for dn_cell_forward_bf16); waveform
    (shared initial phases; Griffin-Lim turns the ~3e-3 relative magnitude error into a different phase path, so the bound is set
    from measurement on the MI355X -- RMS 1.2e-3, max-abs 0.012 on denoised frames of RMS 0.009, i.e. ~13 % of the signal where the
    fp32 path holds 0.03 % -- with a 4x margin): RMS <= 5e-3, max-abs <= 5e-2."""
    from audio_denoising_amd.pipeline import Denoiser, HopPipeline
    from oracle import dsp_ref, pipeline_ref
    p = pipeline_ref.PARAMS_S
    m = _model(dev, 5)
    m.conv_precision = "bf16"
    dn = Denoiser(m, p.sample_rate, p.n_fft, p.hop, p.n_mels)
    g = torch.Generator().manual_seed(1234)
    frames = 0.1 * torch.randn(256, p.n_fft, generator=g)
    init = torch.rand(256, p.n_stft, 3, dtype=torch.complex64, generator=torch.Generator().manual_seed(4321))
    out, hx, resid = dn.process_frame(frames.to(dev), None, init_angles=init.to(dev), return_residual=True)
    idx = torch.arange(0, 256)           # every stream
    fb = dsp_ref.melscale_fbanks(p.n_stft, p.n_mels, p.sample_rate)
    with torch.no_grad():
        ref = pipeline_ref.process_frame(_state_dict("dari_tult"), frames[idx], torch.zeros(len(idx), 17, 5), p, fb, init_angles=init[idx])
    e = (resid.cpu()[idx] - ref["predicted_diff"]).numpy()
    rel = float(np.sqrt(np.mean(e ** 2)) / np.sqrt(np.mean(ref["predicted_diff"].numpy() ** 2)))
    w = (out.cpu()[idx] - ref["out"]).numpy()
    w_rms, w_max = float(np.sqrt(np.mean(w ** 2))), float(np.abs(w).max())
    print(f"bf16 hop: residual rel-RMS {rel:.2e} max-abs {np.abs(e).max():.3f}; waveform RMS {w_rms:.2e} max-abs {w_max:.3f} "
          f"(ref RMS {float(ref['out'].pow(2).mean().sqrt()):.3f}); hx max-abs {float((hx.cpu()[idx] - ref['hx']).abs().max()):.2e}")
    assert rel <= 1e-2 and np.abs(e).max() <= 5e-1 and np.abs(e).max() > 1e-5
    assert w_rms <= 5e-3 and w_max <= 5e-2
    assert float((hx.cpu()[idx] - ref["hx"]).abs().max()) <= 5e-2
    # the pipelined hop runs the same bf16 front half: bit-equal to the one-launch hop
    fd = frames.to(dev)
    hx_a, hx_b = dn.init_hx(256), dn.init_hx(256)
    oa, ob = torch.empty(256, p.n_fft, device=dev), torch.empty(256, p.n_fft, device=dev)
    dn.process_frame_(fd, hx_a, oa, seed=5, stream_id0=0)
    pipe = HopPipeline(dn, 256)
    pipe.submit(fd, hx_b, ob, seed=5, stream_id0=0)
    pipe.flush()
    torch.cuda.synchronize()
    assert torch.equal(hx_a, hx_b) and torch.equal(oa, ob)
    m.conv_precision = "fp32"
    with pytest.raises(RuntimeError):
        pipe.submit(fd, hx_b, ob)                  # the precision of a pipe is fixed at creation


# ------------------------------------------------------------------ sibling model MOMO3
MOMO_CFG = dict(num_compressed_bins=3, in_size=1, hidden_sizes=(16, 16, 16), kernel_sizes=(3, 3, 3), strides=(2, 2, 2), paddings=(1, 0, 1), num_gaussians=6)


def _momo(dev, C=3):
    from momo3 import MOMO3                  # the reference's import line for this model
    from oracle import momo_ref
    m = MOMO3(**dict(MOMO_CFG, num_compressed_bins=C))
    m.load_state_dict(momo_ref.unflatten_weights(np.fromfile(os.path.join(GOLDEN, "weights_momo3_4d4ea0.bin"), dtype=np.float32)))
    return m.eval().to(dev)


@pytest.mark.parametrize("name", sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "momo3_B*_T*_F*.npz"))))
def test_momo3_forward_matches_reference_golden(dev, name):
    """MOMO3 (frame-delta channel, position code at the input only, paddings (1,0,1): 22 -> 11 -> 5 -> 3) on the general fp32 MFMA
    conv tiles against vectors from the reference's own class, incl. batch 256 and a case that continues a sequence (prev given)."""
    g = load_golden(name)
    m = _momo(dev, g["hx0"].shape[2])
    prev = torch.from_numpy(g["prev"]).to(dev) if "prev" in g.files else None
    with torch.no_grad():
        out, hx = m(torch.from_numpy(g["x"]).to(dev), torch.from_numpy(g["hx0"]).to(dev), prev=prev)
    assert out.shape == g["out"].shape and hx.shape == g["hx1"].shape
    assert np.abs(out.cpu().numpy() - g["out"]).max() <= TOL_RESIDUAL
    assert np.abs(hx.cpu().numpy() - g["hx1"]).max() <= TOL_RESIDUAL


def test_momo3_conventions_chain_and_errors(dev):
    g = load_golden("momo3_conventions.npz")
    m = _momo(dev)
    o2, h2 = m(torch.from_numpy(g["x2"]).to(dev))                     # (T,F) input, hx=None (momo3.py:301-316)
    assert o2.shape == (3, 22) and h2.shape == (1, 16, 3)
    assert np.abs(o2.cpu().numpy() - g["out2"]).max() <= TOL_RESIDUAL
    hx, prev = None, None
    for h in range(12):                                               # hx and prev carried by the caller, as the reference's API has it
        x = torch.from_numpy(g["xs"][h]).to(dev)
        o, hx = m(x, hx, prev=prev)
        prev = m.last_frame(x)
        assert np.abs(o.cpu().numpy() - g["outs"][h]).max() <= TOL_RESIDUAL
    assert np.abs(hx.cpu().numpy() - g["hx_final"]).max() <= TOL_RESIDUAL
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 44, device=dev))                          # 44 bins compress to 5, hx default has 3: shape error as in the reference
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 22))                                      # CPU tensor: no fallback


def test_griffinlim_head_start_is_bit_identical_at_batch_256(dev):
    """The pipelined hop's head start (front workgroups run the first iterations of their own frame's Griffin-Lim chain and park it in
    HBM; on by default up to 256 streams) must not change a bit: off, default and a long one give the same frames and the same hx,
    in frame mode at batch 256 and in streaming mode."""
    import ctypes as C
    from audio_denoising_amd.pipeline import Denoiser, HopPipeline, PipelinedStream
    p = _params("S")
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    g = torch.Generator().manual_seed(77)
    hops = [(0.1 * torch.randn(256, p.n_fft, generator=g)).to(dev) for _ in range(4)]
    res = []
    for split in (0, None, 13):
        pipe = HopPipeline(dn, 256)
        if split is not None:
            dn.lib.check(dn.lib.dn_pipe_set_head_start(pipe.handle, split))
        hx = dn.init_hx(256)
        outs = [torch.empty(256, p.n_fft, device=dev) for _ in hops]
        for i, f in enumerate(hops):
            pipe.submit(f, hx, outs[i], seed=5, stream_id0=100)
        pipe.flush()
        torch.cuda.synchronize()
        res.append((hx, outs))
    for hx, outs in res[1:]:
        assert torch.equal(hx, res[0][0])
        for a, b in zip(outs, res[0][1]):
            assert torch.equal(a, b)
    # fewer Griffin-Lim iterations than the head start: the front workgroup runs ALL of them, the next launch only the final istft
    dn2 = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels, n_iter=2)
    outs2 = []
    for split in (0, 7):
        pipe = HopPipeline(dn2, 64)
        dn2.lib.check(dn2.lib.dn_pipe_set_head_start(pipe.handle, split))
        hx = dn2.init_hx(64)
        o = [torch.empty(64, p.n_fft, device=dev) for _ in range(3)]
        for i in range(3):
            pipe.submit(hops[i][:64].contiguous(), hx, o[i], seed=5)
        pipe.flush()
        torch.cuda.synchronize()
        outs2.append(torch.stack(o))
    assert torch.equal(outs2[0], outs2[1]) and torch.isfinite(outs2[0]).all()
    sres = []
    for split in (0, 9):
        ps = PipelinedStream(dn, 8, seed=3)
        dn.lib.check(dn.lib.dn_pipe_set_head_start(ps.handle, split))
        o = [ps.push(hops[i][:8, :p.hop].contiguous()) for i in range(4)] + [ps.flush()]
        sres.append(torch.cat(o, 1))
    assert torch.equal(sres[0], sres[1])


@pytest.mark.parametrize("batch", [5, 1027])
def test_griffinlim_wavefront_per_stream_is_bit_identical_to_per_column(dev, batch):
    """dn_pipe_set_gl_schedule: one wavefront per stream (the three columns interleaved in one wave, the overlap-add in registers, four
    streams a workgroup, no workgroup barrier; the default from 1,024 streams on) against one wavefront per column -- the same frames,
    hx, overlap-add lines and emitted hops bit for bit: frame mode with device-RNG phases, with injected phases, resuming a head start,
    and the int16 streaming transport.  Batch 5 / 1,027: a workgroup with one live wave; more streams than the machine holds at once."""
    from audio_denoising_amd import _lib
    from audio_denoising_amd.pipeline import Denoiser, HopPipeline, PipelinedStream
    p = _params("S")
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    g = torch.Generator().manual_seed(4000 + batch)
    hops = [(0.1 * torch.randn(batch, p.n_fft, generator=g)).to(dev) for _ in range(3)]
    inits = [torch.rand(batch, p.n_stft, 3, dtype=torch.complex64, generator=g).to(dev) for _ in range(3)]
    for variant in ("rng", "init", "head_start"):
        res = []
        for sched in (_lib.DN_GL_WAVE_PER_COLUMN, _lib.DN_GL_WAVE_PER_STREAM, _lib.DN_GL_AUTO):
            pipe = HopPipeline(dn, batch)
            pipe.set_gl_schedule(sched)
            pipe.set_head_start(5 if variant == "head_start" else 0)
            hx = dn.init_hx(batch)
            outs = [torch.empty(batch, p.n_fft, device=dev) for _ in hops]
            for i, f in enumerate(hops):
                pipe.submit(f, hx, outs[i], seed=21, stream_id0=9, init_angles=inits[i] if variant == "init" else None)
            pipe.flush()
            torch.cuda.synchronize()
            res.append((hx, outs))
        for hx, outs in res[1:]:
            assert torch.equal(hx, res[0][0])
            for a, b in zip(outs, res[0][1]):
                assert torch.equal(a, b) and torch.isfinite(a).all()
        assert res[0][1][0].abs().max().item() > 1e-3
    # streaming transport, int16 in and out
    sig = (0.3 * torch.randn(batch, 6 * p.hop, generator=g)).clamp(-1, 1)
    pcm = (sig * 32767.0).to(torch.int16).to(dev)
    sres = []
    for sched in (_lib.DN_GL_WAVE_PER_COLUMN, _lib.DN_GL_WAVE_PER_STREAM):
        ps = PipelinedStream(dn, batch, seed=3, stream_id0=40)
        ps.set_gl_schedule(sched)
        o = [ps.push(pcm[:, i * p.hop:(i + 1) * p.hop].contiguous()) for i in range(6)] + [ps.flush(s16=True)]
        ring, ola, hx, frames = ps.state()
        torch.cuda.synchronize()
        sres.append((torch.cat(o, 1), ola, hx))
    for a, b in zip(sres[0], sres[1]):
        assert torch.equal(a, b)
    assert sres[0][0].dtype == torch.int16 and sres[0][0].abs().max().item() > 0


def test_wavefront_per_stream_schedule_matches_the_oracle_at_1024_streams(dev):
    """The saturated regime's own schedule (BASELINE configs[4]: 1,024 streams per GPU) directly against the oracle: two pipelined hops with
    injected Griffin-Lim phases, a sample of streams spread over the batch (first, last, one per XCD residue), residual and waveform."""
    from audio_denoising_amd import _lib
    from audio_denoising_amd.pipeline import Denoiser, HopPipeline
    from oracle import dsp_ref, pipeline_ref
    p = pipeline_ref.PARAMS_S
    B = 1024
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    g = torch.Generator().manual_seed(88)
    hops = [0.1 * torch.randn(B, p.n_fft, generator=g) for _ in range(2)]
    inits = [torch.rand(B, p.n_stft, 3, dtype=torch.complex64, generator=g) for _ in range(2)]
    pipe = HopPipeline(dn, B)
    pipe.set_gl_schedule(_lib.DN_GL_WAVE_PER_STREAM)
    hx = dn.init_hx(B)
    outs = [torch.empty(B, p.n_fft, device=dev) for _ in hops]
    for i in range(2):
        pipe.submit(hops[i].to(dev), hx, outs[i], seed=0, init_angles=inits[i].to(dev))
    pipe.flush()
    torch.cuda.synchronize()
    idx = torch.tensor([0, 1, 2, 3, 129, 258, 387, 516, 645, 774, 903, 1020, 1021, 1022, 1023])
    fb = dsp_ref.melscale_fbanks(p.n_stft, p.n_mels, p.sample_rate)
    sd = _state_dict("dari_tult")
    h = torch.zeros(len(idx), 17, 5)
    with torch.no_grad():
        for i in range(2):
            ref = pipeline_ref.process_frame(sd, hops[i][idx], h, p, fb, init_angles=inits[i][idx])
            h = ref["hx"]
            _wave_close(outs[i].cpu()[idx].numpy(), ref["out"].numpy())
    assert (hx.cpu()[idx] - h).abs().max().item() <= TOL_HX_STREAM
    # at the guard bands: both hops against the float64 yardstick
    y64, h64 = _f64_frames([hh[idx] for hh in hops], [ia[idx] for ia in inits], p)
    for i in range(2):
        e = outs[i].cpu()[idx].numpy().astype(np.float64) - y64[i]
        assert np.sqrt((e ** 2).mean()) <= GUARD_F64_WAVE_RMS and np.abs(e).max() <= GUARD_F64_WAVE_MAX, (i, np.sqrt((e ** 2).mean()), np.abs(e).max())
    assert np.abs(hx.cpu()[idx].numpy() - h64).max() <= GUARD_HX


@pytest.mark.parametrize("depth", [2, 3, 4])
def test_deep_pipe_is_bit_identical_to_depth_one_at_batch_256(dev, depth):
    """dn_pipe_set_depth (the headline's schedule): `depth` hops of every stream in flight, a frame's Griffin-Lim chain cut into `depth` segments
    that run one wavefront per stream in consecutive launches.  Six chained hops at batch 256 with device-RNG phases, then again with injected
    phases and a drain in mid-sequence: frames and hx equal the depth-1 pipe (wavefront per column + head start) bit for bit; streaming
    mode emits the same int16 samples `depth - 1` pushes later and leaves the same overlap-add lines and hx."""
    from audio_denoising_amd.pipeline import Denoiser, HopPipeline, PipelinedStream
    p = _params("S")
    B = 256
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    g = torch.Generator().manual_seed(600 + depth)
    hops = [(0.1 * torch.randn(B, p.n_fft, generator=g)).to(dev) for _ in range(6)]
    inits = [torch.rand(B, p.n_stft, 3, dtype=torch.complex64, generator=g).to(dev) for _ in range(6)]
    for variant in ("rng", "init+drain", "rng+head_start"):
        res = []
        for d in (1, depth):
            pipe = HopPipeline(dn, B)
            pipe.set_depth(d)
            if variant == "rng+head_start" and d > 1:
                pipe.set_head_start(3)          # the front workgroup runs three iterations first; segment 0 resumes from what it parked
            hx = dn.init_hx(B)
            outs = [torch.empty(B, p.n_fft, device=dev) for _ in hops]
            for i, f in enumerate(hops):
                pipe.submit(f, hx, outs[i], seed=77, stream_id0=5, init_angles=inits[i] if variant == "init+drain" else None)
                if variant == "init+drain" and i == 2:
                    pipe.flush()
            pipe.flush()
            torch.cuda.synchronize()
            assert pipe.counters()[2] is False
            res.append((hx, outs))
        assert torch.equal(res[0][0], res[1][0])
        for a, b in zip(res[0][1], res[1][1]):
            assert torch.equal(a, b) and torch.isfinite(a).all()
    # fewer iterations than segments: some (n_iter 1) or all (n_iter 0: Griffin-Lim is the istft of the drawn phases) of the segments are empty
    for n_iter in (0, 1):
        dn_few = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels, n_iter=n_iter)
        few = []
        for d in (1, depth):
            pipe = HopPipeline(dn_few, 8)
            pipe.set_depth(d)
            hx = dn_few.init_hx(8)
            outs = [torch.empty(8, p.n_fft, device=dev) for _ in range(depth + 1)]
            for i in range(depth + 1):
                pipe.submit(hops[i][:8].contiguous(), hx, outs[i], seed=4)
            pipe.flush()
            torch.cuda.synchronize()
            few.append(torch.stack(outs))
        assert torch.equal(few[0], few[1]) and torch.isfinite(few[0]).all() and few[0].abs().max().item() > 0
    sig = (0.3 * torch.randn(8, 8 * p.hop, generator=g)).clamp(-1, 1)
    pcm = (sig * 32767.0).to(torch.int16).to(dev)
    sres = []
    for d in (1, depth):
        ps = PipelinedStream(dn, 8, seed=3, stream_id0=40)
        ps.set_depth(d)
        o = [ps.push(pcm[:, i * p.hop:(i + 1) * p.hop].contiguous()) for i in range(8)] + [ps.flush(s16=True)]
        ring, ola, hx, frames = ps.state()
        torch.cuda.synchronize()
        sres.append((torch.cat(o, 1), ola, hx))
    lag = (depth - 1) * p.hop
    assert torch.equal(sres[1][0][:, lag:], sres[0][0]) and not sres[1][0][:, :lag].any() and sres[0][0].abs().max().item() > 0
    assert torch.equal(sres[0][1], sres[1][1]) and torch.equal(sres[0][2], sres[1][2])


@pytest.mark.parametrize("depth", [1, 4])
def test_pipelined_hop_at_batch_256_directly_against_the_oracle(dev, depth):
    """The configuration the bench times, compared with the oracle DIRECTLY (not through pipelined == serial): HopPipeline at batch 256 with
    its defaults (depth 1: a wavefront per column and the head start; depth 4: the bench's default -- chain segments, a wavefront per stream),
    three chained hops with injected Griffin-Lim phases (dn_pipe_reserve_parity), every stream: waveform and carried hx."""
    from audio_denoising_amd.pipeline import Denoiser, HopPipeline
    from oracle import dsp_ref, pipeline_ref
    p = pipeline_ref.PARAMS_S
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    g = torch.Generator().manual_seed(2024)
    hops = [0.1 * torch.randn(256, p.n_fft, generator=g) for _ in range(3)]
    inits = [torch.rand(256, p.n_stft, 3, dtype=torch.complex64, generator=g) for _ in range(3)]
    pipe = HopPipeline(dn, 256)
    pipe.set_depth(depth)
    hx = dn.init_hx(256)
    outs = [torch.empty(256, p.n_fft, device=dev) for _ in hops]
    for i in range(3):
        pipe.submit(hops[i].to(dev), hx, outs[i], seed=0, init_angles=inits[i].to(dev))
    pipe.flush()
    torch.cuda.synchronize()
    fb = dsp_ref.melscale_fbanks(p.n_stft, p.n_mels, p.sample_rate)
    sd = _state_dict("dari_tult")
    h = torch.zeros(256, 17, 5)
    with torch.no_grad():
        for i in range(3):
            ref = pipeline_ref.process_frame(sd, hops[i], h, p, fb, init_angles=inits[i])
            h = ref["hx"]
            rms, mx = _wave_close(outs[i].cpu().numpy(), ref["out"].numpy())
            per_stream = (outs[i].cpu() - ref["out"]).pow(2).mean(dim=1).sqrt().numpy()
            assert rms <= GUARD_B256_WAVE_RMS and mx <= GUARD_B256_WAVE_MAX and np.median(per_stream) <= GUARD_B256_STREAM_MEDIAN_RMS, (i, rms, mx)
    assert (hx.cpu() - h).abs().max().item() <= GUARD_HX


@pytest.mark.parametrize("H", [2, 4])
def test_hop_groups_are_bit_identical_to_the_one_hop_pipe_at_batch_256(dev, H):
    """dn_pipe_set_group (the headline's schedule since round 4): a launch carries H consecutive hops of every stream -- their front halves in
    order, hx handed on inside the launch -- beside the WHOLE Griffin-Lim chains of the previous group, one wavefront each; nothing is parked
    between launches.  Nine chained hops at batch 256 (full groups and a short last one), device-RNG phases, then injected phases with uneven
    submits and a drain in mid-sequence: frames and hx equal the depth-1 pipe (a wavefront per column + head start) bit for bit.  Streaming: the
    same int16 samples H - 1 pushes later, the same overlap-add lines, rings and hx.  Also 32 > n_iter in {0, 1}."""
    from audio_denoising_amd.pipeline import Denoiser, HopPipeline, PipelinedStream
    p = _params("S")
    B, n = 256, 9
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    g = torch.Generator().manual_seed(900 + H)
    frames = (0.1 * torch.randn(n, B, p.n_fft, generator=g)).to(dev)
    inits = torch.rand(n, B, p.n_stft, 3, dtype=torch.complex64, generator=g).to(dev)
    for variant in ("rng", "init+uneven+drain"):
        ia = inits if variant != "rng" else None
        pipe = HopPipeline(dn, B)
        hx_a = dn.init_hx(B)
        out_a = torch.empty_like(frames)
        for i in range(n):
            pipe.submit(frames[i], hx_a, out_a[i], seed=77, stream_id0=5, init_angles=None if ia is None else ia[i])
        pipe.flush()
        grp = HopPipeline(dn, B)
        grp.set_group(H)
        hx_b = dn.init_hx(B)
        out_b = torch.empty_like(frames)
        sizes = [H] * ((n + H - 1) // H) if variant == "rng" else [1, H, 1, H, H, H]
        i = 0
        for k in sizes:
            k = min(k, n - i)
            if k == 0:
                break
            grp.submit_group(frames[i:i + k], hx_b, out_b[i:i + k], seed=77, stream_id0=5, init_angles=None if ia is None else ia[i:i + k])
            i += k
            if variant != "rng" and i == 1 + H:
                grp.flush()
        grp.flush()
        torch.cuda.synchronize()
        assert grp.counters()[1] == n and grp.counters()[2] is False
        assert torch.equal(hx_a, hx_b) and torch.equal(out_a, out_b) and torch.isfinite(out_a).all() and out_a.abs().max().item() > 0
    for n_iter in (0, 1):
        dn_few = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels, n_iter=n_iter)
        few = []
        for grouped in (False, True):
            pipe = HopPipeline(dn_few, 8)
            hx = dn_few.init_hx(8)
            f8 = frames[:H + 1, :8].contiguous()
            outs = torch.empty_like(f8)
            if grouped:
                pipe.set_group(H)
                pipe.submit_group(f8[:H], hx, outs[:H], seed=4)
                pipe.submit_group(f8[H:], hx, outs[H:], seed=4)
            else:
                for i in range(H + 1):
                    pipe.submit(f8[i], hx, outs[i], seed=4)
            pipe.flush()
            torch.cuda.synchronize()
            few.append(outs)
        assert torch.equal(few[0], few[1]) and torch.isfinite(few[0]).all() and few[0].abs().max().item() > 0
    n_push = 3 * H
    sig = (0.3 * torch.randn(8, n_push * p.hop, generator=g)).clamp(-1, 1)
    pcm = (sig * 32767.0).to(torch.int16).to(dev)
    ps = PipelinedStream(dn, 8, seed=3, stream_id0=40)
    o = [ps.push(pcm[:, i * p.hop:(i + 1) * p.hop].contiguous()) for i in range(n_push)] + [ps.flush(s16=True)]
    ring_a, ola_a, hx_a, frames_a = ps.state()
    ea = torch.cat(o, 1)
    pg = PipelinedStream(dn, 8, seed=3, stream_id0=40)
    pg.set_group(H)
    o = []
    for i in range(0, n_push, H):
        hops = torch.stack([pcm[:, j * p.hop:(j + 1) * p.hop] for j in range(i, i + H)]).contiguous()
        o += list(pg.push_group(hops))
    tail, valid = pg.flush_group(s16=True)
    o += list(tail)
    ring_b, ola_b, hx_b, frames_b = pg.state()
    torch.cuda.synchronize()
    eb = torch.cat(o, 1)
    lag = (H - 1) * p.hop
    assert valid == H and frames_a == frames_b == n_push - 1
    assert torch.equal(eb[:, lag:lag + ea.shape[1]], ea) and not eb[:, :lag].any() and ea.abs().max().item() > 0
    assert torch.equal(ring_a, ring_b) and torch.equal(ola_a, ola_b) and torch.equal(hx_a, hx_b)


@pytest.mark.parametrize("tag,conv", [("R2", "fp32"), ("S", "bf16")])
def test_hop_groups_other_instantiations_equal_the_one_hop_pipe(dev, tag, conv):
    """The group kernel's other instantiations: 64 mels (server.py's parameters: four compressed bins, run-time lengths in the model stage) and the
    bf16 conv tiles of config 3 -- frames and hx equal the one-hop pipe of the same precision bit for bit (groups of three: a short last group)."""
    from audio_denoising_amd.pipeline import Denoiser, HopPipeline
    p = _params(tag)
    m = _model(dev, p.num_compressed_bins)
    m.conv_precision = conv
    dn = Denoiser(m, p.sample_rate, p.n_fft, p.hop, p.n_mels)
    B, n, H = 40, 5, 3
    frames = (0.1 * torch.randn(n, B, p.n_fft, generator=torch.Generator().manual_seed(77))).to(dev)
    one, grp = HopPipeline(dn, B), HopPipeline(dn, B)
    grp.set_group(H)
    hx_a, hx_b = dn.init_hx(B), dn.init_hx(B)
    out_a, out_b = torch.empty_like(frames), torch.empty_like(frames)
    for i in range(n):
        one.submit(frames[i], hx_a, out_a[i], seed=5, stream_id0=2)
    one.flush()
    for i in range(0, n, H):
        grp.submit_group(frames[i:i + H], hx_b, out_b[i:i + H], seed=5, stream_id0=2)
    grp.flush()
    torch.cuda.synchronize()
    assert torch.equal(hx_a, hx_b) and torch.equal(out_a, out_b) and torch.isfinite(out_a).all() and out_a.abs().max().item() > 0


@pytest.mark.parametrize("batch,H", [(1, 4), (3, 2), (17, 1), (255, 2), (257, 4), (1001, 1), (1001, 2)])
def test_hop_groups_at_odd_batch_sizes_equal_the_one_hop_pipe(dev, batch, H):
    """Batch sizes that are no multiple of the streams a chain workgroup holds (4 / H): a last workgroup with idle wavefronts, a single stream, more
    streams than CUs.  Frames, hx and -- streaming, int16 -- the emitted samples and the stream state equal the one-hop pipe bit for bit."""
    from audio_denoising_amd.pipeline import Denoiser, HopPipeline, PipelinedStream
    p = _params("S")
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    g = torch.Generator().manual_seed(31 * batch + H)
    n = 2 * H + 1
    frames = (0.1 * torch.randn(n, batch, p.n_fft, generator=g)).to(dev)
    one, grp = HopPipeline(dn, batch), HopPipeline(dn, batch)
    grp.set_group(H)
    hx_a, hx_b = dn.init_hx(batch), dn.init_hx(batch)
    out_a, out_b = torch.empty_like(frames), torch.empty_like(frames)
    for i in range(n):
        one.submit(frames[i], hx_a, out_a[i], seed=3, stream_id0=11)
    one.flush()
    for i in range(0, n, H):
        grp.submit_group(frames[i:i + H], hx_b, out_b[i:i + H], seed=3, stream_id0=11)
    grp.flush()
    torch.cuda.synchronize()
    assert torch.equal(hx_a, hx_b) and torch.equal(out_a, out_b) and torch.isfinite(out_a).all() and out_a.abs().max().item() > 0
    n_push = 4 * H          # (the first push primes the ring and a hop is emitted a push after its frame was folded: enough pushes for samples to come out)
    pcm = ((0.3 * torch.randn(batch, n_push * p.hop, generator=g)).clamp(-1, 1) * 32767.0).to(torch.int16).to(dev)
    ps = PipelinedStream(dn, batch, seed=3, stream_id0=40)
    ea = torch.cat([ps.push(pcm[:, i * p.hop:(i + 1) * p.hop].contiguous()) for i in range(n_push)] + [ps.flush(s16=True)], 1)
    st_a = ps.state()
    pg = PipelinedStream(dn, batch, seed=3, stream_id0=40)
    pg.set_group(H)
    o = []
    for i in range(0, n_push, H):
        o += list(pg.push_group(torch.stack([pcm[:, j * p.hop:(j + 1) * p.hop] for j in range(i, i + H)]).contiguous()))
    tail, valid = pg.flush_group(s16=True)
    st_b = pg.state()
    torch.cuda.synchronize()
    eb = torch.cat(o + list(tail), 1)
    lag = (H - 1) * p.hop
    assert torch.equal(eb[:, lag:lag + ea.shape[1]], ea) and not eb[:, :lag].any() and ea.abs().max().item() > 0
    for x, y in zip(st_a[:3], st_b[:3]):
        assert torch.equal(x, y)


def test_hop_groups_at_batch_256_directly_against_the_oracle(dev):
    """The configuration the bench times since round 4 (groups of four hops, whole chains), compared with the oracle DIRECTLY: batch 256, two
    groups (five chained hops: a full group and a short one) with injected Griffin-Lim phases; every fourth stream of the 256 (the CPU oracle takes
    ~0.1 s per stream and hop; all 256 streams of this schedule are compared bit for bit with the one-hop pipe, which IS compared with the oracle
    on all of them): waveform at the batch-256 guard bands and the carried hx."""
    from audio_denoising_amd.pipeline import Denoiser, HopPipeline
    from oracle import dsp_ref, pipeline_ref
    p = pipeline_ref.PARAMS_S
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    g = torch.Generator().manual_seed(2025)
    n = 5
    hops = 0.1 * torch.randn(n, 256, p.n_fft, generator=g)
    inits = torch.rand(n, 256, p.n_stft, 3, dtype=torch.complex64, generator=g)
    pipe = HopPipeline(dn, 256)
    pipe.set_group(4)
    hx = dn.init_hx(256)
    fd, out = hops.to(dev), torch.empty(n, 256, p.n_fft, device=dev)
    pipe.submit_group(fd[:4], hx, out[:4], seed=0, init_angles=inits[:4].to(dev))
    pipe.submit_group(fd[4:], hx, out[4:], seed=0, init_angles=inits[4:].to(dev))
    pipe.flush()
    torch.cuda.synchronize()
    fb = dsp_ref.melscale_fbanks(p.n_stft, p.n_mels, p.sample_rate)
    sd = _state_dict("dari_tult")
    idx = torch.arange(1, 256, 4)
    h = torch.zeros(len(idx), 17, 5)
    with torch.no_grad():
        for i in range(n):
            ref = pipeline_ref.process_frame(sd, hops[i][idx], h, p, fb, init_angles=inits[i][idx])
            h = ref["hx"]
            got = out[i].cpu()[idx]
            rms, mx = _wave_close(got.numpy(), ref["out"].numpy())
            per_stream = (got - ref["out"]).pow(2).mean(dim=1).sqrt().numpy()
            assert rms <= GUARD_B256_WAVE_RMS and mx <= GUARD_B256_WAVE_MAX and np.median(per_stream) <= GUARD_B256_STREAM_MEDIAN_RMS, (i, rms, mx)
    assert (hx.cpu()[idx] - h).abs().max().item() <= GUARD_HX


def test_hop_groups_refuse_what_they_do_not_run_and_replay_under_a_graph(dev):
    """Errors of the group API (n_fft 1536, depth > 1, more hops than the group holds, overlapping outputs, single-hop stream calls on a group pipe),
    and one captured dn_pipe_stream_push_group replayed: the control block advances on the device, so replays equal eager pushes bit for bit."""
    from audio_denoising_amd._lib import DnError
    from audio_denoising_amd.pipeline import Denoiser, HopPipeline, PipelinedStream
    p = _params("S")
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    r1 = _params("R1")
    with pytest.raises(DnError, match="1024"):
        HopPipeline(Denoiser(_model(dev, 4), r1.sample_rate, r1.n_fft, r1.hop, r1.n_mels), 4).set_group(2)
    pipe = HopPipeline(dn, 4)
    pipe.set_depth(2)
    with pytest.raises(DnError, match="deeper"):
        pipe.set_group(2)
    pipe.set_depth(1)
    pipe.set_group(2)
    with pytest.raises(DnError, match="group pipe"):
        pipe.set_depth(2)
    f = torch.zeros(3, 4, p.n_fft, device=dev)
    with pytest.raises(DnError, match="hops must be"):
        pipe.submit_group(f, dn.init_hx(4), torch.empty_like(f))
    with pytest.raises(DnError, match="overlap"):
        pipe.lib.check(pipe.lib.dn_pipe_submit_group(pipe.handle, f.data_ptr(), 0, dn.init_hx(4).data_ptr(), f.data_ptr(), 0, None, 0, 0, 0, 2, 32, 0.99, None))
    hs = HopPipeline(dn, 4)              # (up to 256 streams a one-hop pipe runs a head start: a split hop cannot)
    with pytest.raises(DnError, match="head start"):
        hs.set_split(1)
    hs.set_head_start(0)
    hs.set_gl_schedule(2)
    hs.set_split(1)
    with pytest.raises(DnError, match="splits every hop"):
        hs.set_head_start(3)
    ps = PipelinedStream(dn, 4)
    ps.set_group(2)
    with pytest.raises(DnError, match="groups of hops"):
        ps.push(torch.zeros(4, p.hop, device=dev))
    # replay
    g = torch.Generator().manual_seed(31)
    sig = (0.2 * torch.randn(64, 12 * p.hop, generator=g)).to(dev)
    groups = [torch.stack([sig[:, (2 * k + j) * p.hop:(2 * k + j + 1) * p.hop] for j in range(2)]).contiguous() for k in range(6)]
    eager = PipelinedStream(dn, 64, seed=9)
    eager.set_group(2)
    ref = [eager.push_group(x) for x in groups]
    rep = PipelinedStream(dn, 64, seed=9)
    rep.set_group(2)
    hop_buf, out_buf = torch.empty_like(groups[0]), torch.empty_like(groups[0])
    rep._bind()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        rep.push_group_(hop_buf, out_buf, check_weights=False)
    got = []
    for x in groups:
        hop_buf.copy_(x)
        graph.replay()
        got.append(out_buf.clone())
    torch.cuda.synchronize()
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    assert torch.stack(ref).abs().max().item() > 0


@pytest.mark.parametrize("short,F", [("dari_tult", 64), ("dari_tult", 80), ("dari_tult2", 64)])
def test_checkpoint_in_the_reference_format_loads_and_runs_on_the_gpu(dev, tmp_path, short, F):
    """SURVEY 8(f)-3 on the GPU tier: a `.pth` written with torch.save in the reference's own layout (app.py:75-91: config,
    model_state_dict, optimizer_state_dict, loss_record ...) from the committed weight blob -> checkpoint.load_model(path, device) as the
    app's loader does it (app3.py:59-116) -> HIP forward, against the golden produced by the reference's own GRUUNet2.  F = 80 goes through
    `num_compressed_bins=5` on a checkpoint that stores 4; the other spelling the loader accepts (hparams /
    state_dict) gives the same native weights."""
    from audio_denoising_amd import checkpoint as ck
    stored = dict(CFG, num_compressed_bins=4)
    path = str(tmp_path / "checkpoint.pth")
    torch.save({"config": stored, "model_state_dict": _state_dict(short), "optimizer_state_dict": {}, "last_epoch": 7,
                "loss_record": {"train": [7.43], "test": []}, "total_training_iters": 92637}, path)
    m = ck.load_model(path, device=dev, num_compressed_bins=None if F == 64 else 5)
    assert not m.training and next(m.parameters()).is_cuda and m.num_compressed_bins == F // 16
    g = load_golden(f"cell_{short}_B4_T3_F{F}.npz")
    x, hx0 = torch.from_numpy(g["x"]).to(dev), torch.from_numpy(g["hx0"]).to(dev)
    with torch.no_grad():
        out, hx1 = m(x, hx0)
    assert (out.cpu() - torch.from_numpy(g["out"])).abs().max().item() <= GUARD_RESIDUAL
    assert (hx1.cpu() - torch.from_numpy(g["hx1"])).abs().max().item() <= GUARD_HX * 10
    path2 = str(tmp_path / "other_spelling.pth")
    torch.save({"hparams": stored, "state_dict": _state_dict(short)}, path2)
    m2 = ck.load_model(path2, device=dev, num_compressed_bins=F // 16)
    with torch.no_grad():
        out2, _ = m2(x, hx0)
    assert torch.equal(out, out2)
    # the flat export is what the C ABI takes: byte-equal to the committed blob
    b, _ = ck.export_flat(path, str(tmp_path / "flat"))
    assert np.array_equal(np.fromfile(b, dtype=np.float32), np.fromfile(os.path.join(GOLDEN, f"weights_{short}.bin"), dtype=np.float32))


def test_device_rng_known_answers_and_statistics(dev):
    """rand_init=True (app3.py:149-153 -> torch.rand(complex64)) is replaced by an on-device Philox4x32-10.  (1) Bit-exact: 700 streams x 3 x 513
    complex draws against the independent numpy restatement (oracle/philox_ref.py, pinned by Random123's published known-answer vectors; the
    first vector is bin 0 of stream 0 at seed 0), with a seed and stream ids that use the high words.  (2) Statistics over 1.08 M complex
    draws: mean, variance and a Kolmogorov-Smirnov test of real and imaginary parts against U[0,1), and no correlation between real and
    imaginary part, neighbouring bins, columns, neighbouring streams and consecutive frames (seed + 1).  (3) A seeded hop == the same hop fed
    these phases."""
    from scipy import stats
    from audio_denoising_amd.pipeline import Denoiser
    from oracle import philox_ref
    p = _params("S")
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    for ctr, key, out in philox_ref.KAT:
        assert tuple(int(x) for x in philox_ref.philox4x32_10(*ctr, *key)) == out
    z0 = dn.draw_phases(1, 0, 0).cpu().numpy()
    assert z0[0, 0, 0].real == np.float32((0x6627e8d5 >> 8) / 16777216.0) and z0[0, 0, 0].imag == np.float32((0xe169c58d >> 8) / 16777216.0)
    B, seed, sid0 = 700, 0xC0FFEE1234567, (1 << 33) + 5
    z = dn.draw_phases(B, seed, sid0).cpu().numpy()
    assert np.array_equal(z[:24], philox_ref.draw_phases(seed, sid0, 24, p.n_stft))
    assert np.array_equal(z[-3:], philox_ref.draw_phases(seed, sid0 + B - 3, 3, p.n_stft))
    z1 = dn.draw_phases(B, seed + 1, sid0).cpu().numpy()              # the next frame of the same streams
    re, im = z.real.astype(np.float64), z.imag.astype(np.float64)
    n = re.size
    assert n >= 1_000_000 and re.min() >= 0.0 and re.max() < 1.0 and im.min() >= 0.0 and im.max() < 1.0
    for x in (re, im):
        assert abs(x.mean() - 0.5) <= 5 * np.sqrt(1 / 12 / n)                     # 5 sigma of the sample mean
        assert abs(x.var() - 1 / 12) <= 5 * np.sqrt(1 / 180 / n)                  # var of the sample variance of U[0,1) ~ 1/(180 n)
        assert stats.kstest(x.ravel(), "uniform").pvalue > 1e-4
    def corr(a, b):
        return float(np.corrcoef(a.ravel(), b.ravel())[0, 1])
    bound = 5 / np.sqrt(n / 2)
    assert abs(corr(re, im)) <= bound
    assert abs(corr(re[:, :-1, :], re[:, 1:, :])) <= bound                        # neighbouring bins
    assert abs(corr(re[:, :, 0], re[:, :, 1])) <= 5 / np.sqrt(n / 3) and abs(corr(im[:, :, 1], im[:, :, 2])) <= 5 / np.sqrt(n / 3)   # columns
    assert abs(corr(re[:-1], re[1:])) <= bound                                    # neighbouring streams
    assert abs(corr(re, z1.real.astype(np.float64))) <= bound and abs(corr(im, z1.imag.astype(np.float64))) <= bound                 # consecutive frames
    assert len(np.unique(z.view(np.uint64))) > 0.999 * n                          # (24-bit mantissas: a handful of repeated pairs at most)
    # a seeded hop is the hop fed exactly these phases
    g = torch.Generator().manual_seed(8)
    frames = (0.1 * torch.randn(16, p.n_fft, generator=g)).to(dev)
    a, ha = dn.process_frame(frames, None, seed=seed, stream_id0=sid0)
    b, hb = dn.process_frame(frames, None, init_angles=dn.draw_phases(16, seed, sid0))
    assert torch.equal(a, b) and torch.equal(ha, hb)


@pytest.mark.parametrize("batch,s16,depth,staged,defer", [(256, True, 1, False, True), (1024, True, 1, False, True), (7, False, 1, False, True),
                                                          (256, True, 4, False, True), (5, True, 4, False, True),
                                                          (256, True, 1, False, False), (7, False, 1, False, False), (256, True, 4, False, False),
                                                          (256, True, 1, True, False), (1024, True, 1, True, False), (5, False, 4, True, False)])
def test_host_fed_stream_equals_device_fed_stream(dev, batch, s16, depth, staged, defer):
    """dn_pipe_stream_push_host (app3.py:168-172,189,215,244-250: hops cross the host/device boundary): zero copy (the launch reads page-locked
    host memory itself; its emitted hop goes straight to host memory, or -- deferred -- waits in a device buffer and is carried out by the next
    launch) or staged (uploads and downloads on two copy queues beside the hop, device staging double-buffered).
    Whatever overlaps, the samples must be those of the device-fed stream bit for bit (int16 transport at 256 and 1,024 streams, float32
    at an odd batch, a deep pipe), and so must the stream state left behind."""
    from audio_denoising_amd.pipeline import Denoiser, HostFedStream, PipelinedStream
    p = _params("S")
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    g = torch.Generator().manual_seed(321 + batch)
    n = 9
    sig = (0.3 * torch.randn(batch, n * p.hop, generator=g)).clamp(-1, 1)
    host = (sig * 32767.0).to(torch.int16) if s16 else sig
    hops = [host[:, i * p.hop:(i + 1) * p.hop].contiguous() for i in range(n)]
    ref = PipelinedStream(dn, batch, seed=11, stream_id0=2)
    ref.set_depth(depth)
    a = torch.cat([ref.push(h.to(dev)) for h in hops] + [ref.flush(s16=s16)], 1).cpu()
    ra = [t.cpu() for t in ref.state()[:3]]
    hs = HostFedStream(dn, batch, seed=11, stream_id0=2, s16=s16, depth=depth, staged=staged, defer=defer)
    assert hs.LAG == (3 if defer else 2)
    outs = [hs.push(h) for h in hops]
    assert not torch.cat(outs[:hs.LAG], 1).any()    # a push hands out what was emitted LAG pushes earlier
    b = torch.cat(outs[hs.LAG:] + [hs.drain()], 1)
    rb = [t.cpu() for t in hs.state()[:3]]
    assert a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b) and a.abs().max().item() > 0
    for x, y in zip(ra, rb):
        assert torch.equal(x, y)


@pytest.mark.parametrize("wait_each", [False, True])
def test_host_pushes_of_every_transport_mix_and_no_result_is_stranded(dev, wait_each):
    """dn_pipe_stream_push_host with the transports mixed on one pipe -- deferred, direct and staged pushes in turn, each into its own page-locked
    buffer -- and dn_pipe_stream_host_wait either on every push at once (the wait on the NEWEST deferred push has to enqueue the move of its
    samples itself: no later launch will) or only at the end, newest first.  Every push's buffer must hold what the device-fed stream emitted."""
    import ctypes as C
    from audio_denoising_amd import _lib
    from audio_denoising_amd.pipeline import Denoiser, PipelinedStream
    p = _params("S")
    B, n = 37, 10
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    g = torch.Generator().manual_seed(77)
    host = ((0.3 * torch.randn(B, n * p.hop, generator=g)).clamp(-1, 1) * 32767.0).to(torch.int16)
    hops = [host[:, i * p.hop:(i + 1) * p.hop].contiguous().pin_memory() for i in range(n)]
    ref = PipelinedStream(dn, B, seed=4, stream_id0=9)
    want = [ref.push(h.to(dev)).cpu() for h in hops]
    ps = PipelinedStream(dn, B, seed=4, stream_id0=9)
    D, Z, S = _lib.DN_HOST_DEFER, 0, _lib.DN_HOST_STAGED
    flags = [D, D, Z, D, S, D, D, Z, S, D]
    outs = [torch.full((B, p.hop), -7, dtype=torch.int16).pin_memory() for _ in range(n)]
    tickets = []
    with torch.cuda.device(dev):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for i in range(n):
            t = C.c_uint64()
            ps.lib.check(ps.lib.dn_pipe_stream_push_host(ps.handle, hops[i].data_ptr(), 1, outs[i].data_ptr(), 1, 4, 9, dn.n_iter, dn.momentum,
                                                         flags[i], st, C.byref(t)))
            tickets.append(t.value)
            if wait_each:
                ps.lib.check(ps.lib.dn_pipe_stream_host_wait(ps.handle, t.value))
                assert torch.equal(outs[i], want[i]), f"push {i}"
        assert tickets == list(range(n))
        for i in reversed(range(n)):
            ps.lib.check(ps.lib.dn_pipe_stream_host_wait(ps.handle, tickets[i]))
            assert torch.equal(outs[i], want[i]), f"push {i}"
    assert any(w.any() for w in want)
    assert torch.equal(ps.flush(s16=True).cpu(), ref.flush(s16=True).cpu())


def test_app_parameters_refuse_the_schedules_built_for_n_fft_1024(dev):
    """At the reference app's own STFT parameters (app3.py:29-33: n_fft 1536) the wavefront-per-stream Griffin-Lim and the deep pipe are not built
    (the per-lane state of a stream does not fit one wavefront: measured 1.6x slower): the pipe says so instead of doing something else."""
    from audio_denoising_amd import _lib
    from audio_denoising_amd.pipeline import Denoiser, HopPipeline
    p = _params("R1")
    dn = Denoiser(_model(dev, 4, "dari_tult2"), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    pipe = HopPipeline(dn, 8)
    with pytest.raises(RuntimeError, match="1024"):
        pipe.set_depth(2)
    with pytest.raises(RuntimeError, match="1024"):
        pipe.set_gl_schedule(_lib.DN_GL_WAVE_PER_STREAM)
    pipe.set_gl_schedule(_lib.DN_GL_WAVE_PER_COLUMN)
    pipe.set_depth(1)


@pytest.mark.parametrize("batch,depth,stream", [(1027, 1, False), (300, 2, True), (4100, 1, True)])
def test_split_hop_is_bit_identical_to_the_single_launch(dev, batch, depth, stream):
    """dn_pipe_set_split: a hop as two launches (the chains of the hops in flight, then the new hop's front halves as a launch of their own:
    four workgroups a CU, weights read from L2) against the one-launch hop -- frames / emitted hops, hx and stream state bit for bit; 4,100
    streams is above the automatic threshold (DN_SPLIT_AUTO), so there the default is compared with DN_SPLIT_OFF."""
    from audio_denoising_amd import _lib
    from audio_denoising_amd.pipeline import Denoiser, HopPipeline, PipelinedStream
    p = _params("S")
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    g = torch.Generator().manual_seed(batch)
    n = 5
    res = []
    for mode in (_lib.DN_SPLIT_OFF, _lib.DN_SPLIT_AUTO if batch >= 4096 else _lib.DN_SPLIT_ON):
        if stream:
            g.manual_seed(batch)
            sig = (0.3 * torch.randn(batch, n * p.hop, generator=g)).clamp(-1, 1).to(dev)
            ps = PipelinedStream(dn, batch, seed=5)
            ps.set_depth(depth)
            ps.set_gl_schedule(_lib.DN_GL_WAVE_PER_STREAM if depth == 1 else _lib.DN_GL_AUTO)
            ps.set_split(mode)
            outs = [ps.push(sig[:, i * p.hop:(i + 1) * p.hop].contiguous()) for i in range(n)] + [ps.flush()]
            res.append([torch.cat(outs, 1)] + list(ps.state()[:3]))
        else:
            g.manual_seed(batch)
            frames = [(0.1 * torch.randn(batch, p.n_fft, generator=g)).to(dev) for _ in range(n)]
            pipe = HopPipeline(dn, batch)
            pipe.set_depth(depth)
            pipe.set_gl_schedule(_lib.DN_GL_WAVE_PER_STREAM)
            pipe.set_split(mode)
            hx = dn.init_hx(batch)
            outs = [torch.empty(batch, p.n_fft, device=dev) for _ in range(n)]
            for i in range(n):
                pipe.submit(frames[i], hx, outs[i], seed=9)
            pipe.flush()
            torch.cuda.synchronize()
            res.append(outs + [hx])
    for x, y in zip(*res):
        assert torch.equal(x, y)
    assert res[0][0].abs().max().item() > 1e-3


@pytest.mark.parametrize("batch,queues,depth,pipes", [(1024, 2, 2, None), (37, 3, 1, None), (90, 2, 1, 4)])
def test_queued_pipes_equal_one_pipe(dev, batch, queues, depth, pipes):
    """`QueuedHopPipelines` / `QueuedPipelinedStreams`: B streams as Q pipes on Q HIP streams, split hops (the 1,024-stream configuration: two
    queues at depth 2; four pipes taking turns on two queues).  Frames, hx, emitted hops and captured replays must equal ONE pipe of B streams bit for bit -- the generator is keyed by
    the global stream id, so sharding streams over queues changes nothing."""
    from audio_denoising_amd.pipeline import Denoiser, HopPipeline, PipelinedStream, QueuedHopPipelines, QueuedPipelinedStreams
    p = _params("S")
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    g = torch.Generator().manual_seed(batch)
    n = 4
    frames = [(0.1 * torch.randn(batch, p.n_fft, generator=g)).to(dev) for _ in range(n)]
    sig = (0.3 * torch.randn(batch, n * p.hop, generator=g)).clamp(-1, 1).to(dev)
    torch.cuda.synchronize()
    # frame mode
    one = HopPipeline(dn, batch)
    one.set_depth(depth)
    hx1 = dn.init_hx(batch)
    o1 = [torch.empty(batch, p.n_fft, device=dev) for _ in range(n)]
    for i in range(n):
        one.submit(frames[i], hx1, o1[i], seed=9, stream_id0=5)
    one.flush()
    qp = QueuedHopPipelines(dn, batch, queues=queues, depth=depth, pipes=pipes)
    assert len(qp.pipes) == (pipes or queues) and len(qp.streams) == queues
    hx2 = dn.init_hx(batch)
    o2 = [torch.empty(batch, p.n_fft, device=dev) for _ in range(n)]
    qp.after()
    for i in range(n):
        qp.submit(frames[i], hx2, o2[i], seed=9, stream_id0=5)
    qp.flush()
    qp.synchronize()
    torch.cuda.synchronize()
    for x, y in zip(o1 + [hx1], o2 + [hx2]):
        assert torch.equal(x, y)
    assert o1[0].abs().max().item() > 1e-3
    # streaming mode, eager and captured
    ps = PipelinedStream(dn, batch, seed=3, stream_id0=7)
    ps.set_depth(depth)
    a = torch.cat([ps.push(sig[:, i * p.hop:(i + 1) * p.hop].contiguous()) for i in range(n)] + [ps.flush()], 1)
    for captured in (False, True):
        qs = QueuedPipelinedStreams(dn, batch, queues=queues, depth=depth, seed=3, stream_id0=7, pipes=pipes)
        hop = torch.empty(batch, p.hop, device=dev)
        out = torch.empty(batch, p.hop, device=dev)
        replay = qs.graph_steps(hop, out) if captured else None
        res = []
        for i in range(n):
            hop.copy_(sig[:, i * p.hop:(i + 1) * p.hop])
            qs.after()
            if captured:
                replay()
            else:
                qs.push_(hop, out)
            qs.before()
            res.append(out.clone())
        qs.before()
        torch.cuda.synchronize()
        b = torch.cat(res + [qs.flush()[:, :depth * p.hop]], 1)
        assert torch.equal(a, b), "captured" if captured else "eager"


def test_hop_pipeline_picks_the_planned_composition(dev):
    """`pipeline.hop_pipeline`: the composition `throughput_plan` names is what gets built (1,024 streams: two pipes at depth 2 on two HIP
    streams; 300: one pipe at depth 4), and it runs."""
    from audio_denoising_amd.pipeline import Denoiser, HopPipeline, QueuedHopPipelines, hop_pipeline
    p = _params("S")
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels, n_iter=4)
    a, b, c = hop_pipeline(dn, 1024), hop_pipeline(dn, 300), hop_pipeline(dn, 600, grouped=True)
    assert isinstance(a, QueuedHopPipelines) and len(a.pipes) == 2 and a.depth == 2 and [q.batch for q in a.pipes] == [512, 512]
    assert isinstance(b, HopPipeline) and b.depth == 4 and b.group == 0
    assert isinstance(c, HopPipeline) and c.group == 2 and hop_pipeline(dn, 300, grouped=True).group == 4
    frames = 0.1 * torch.randn(1024, p.n_fft, device=dev)
    hx, out = dn.init_hx(1024), torch.zeros(1024, p.n_fft, device=dev)
    torch.cuda.synchronize()
    for _ in range(3):
        a.submit(frames, hx, out, seed=1)
    a.flush()
    a.synchronize()
    assert torch.isfinite(out).all() and out.abs().max().item() > 1e-3 and out[700:].abs().max().item() > 1e-3


def test_several_pushes_captured_as_one_graph_replay(dev):
    """`PipelinedStream.graph_step` with (K, B, hop) tensors: K consecutive pushes captured as ONE hipGraph (one graph launch per K hops).  Four
    replays of a three-push graph equal twelve eager pushes bit for bit."""
    from audio_denoising_amd.pipeline import Denoiser, PipelinedStream
    p = _params("S")
    B, K = 64, 3
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    g = torch.Generator().manual_seed(6)
    sig = (0.3 * torch.randn(B, 12 * p.hop, generator=g)).clamp(-1, 1).to(dev)
    eager = PipelinedStream(dn, B, seed=3)
    a = torch.cat([eager.push(sig[:, i * p.hop:(i + 1) * p.hop].contiguous()) for i in range(12)] + [eager.flush()], 1)
    ps = PipelinedStream(dn, B, seed=3)
    hops = torch.empty(K, B, p.hop, device=dev)
    outs = torch.empty(K, B, p.hop, device=dev)
    step = ps.graph_step(hops, outs)
    res = []
    for r in range(4):
        for k in range(K):
            hops[k].copy_(sig[:, (r * K + k) * p.hop:(r * K + k + 1) * p.hop])
        step.replay()
        res += [outs[k].clone() for k in range(K)]
    b = torch.cat(res + [ps.flush()], 1)
    assert torch.equal(a, b) and a.abs().max().item() > 1e-3
    with pytest.raises(ValueError):
        ps.graph_step(hops, outs[0])
    # the same K pushes from K separate tensors (lists): what the queued form captures per pipe
    from audio_denoising_amd.pipeline import QueuedPipelinedStreams
    qs = QueuedPipelinedStreams(dn, B, queues=2, depth=1, seed=3)
    hl, ol = [torch.empty(B, p.hop, device=dev) for _ in range(K)], [torch.empty(B, p.hop, device=dev) for _ in range(K)]
    replay = qs.graph_steps(hl, ol)
    res = []
    for r in range(4):
        for k in range(K):
            hl[k].copy_(sig[:, (r * K + k) * p.hop:(r * K + k + 1) * p.hop])
        qs.after()
        replay()
        qs.before()
        res += [ol[k].clone() for k in range(K)]
    torch.cuda.synchronize()
    assert torch.equal(a, torch.cat(res + [qs.flush()], 1))


def test_captured_push_of_a_deep_pipe_replays(dev):
    """BASELINE config 5's captured step on a deep pipe: ONE hipGraph-captured dn_pipe_stream_push at depth 4 (which chain segment of which hop a
    wavefront runs is decided from the device-resident control block, nothing hop-dependent is baked into the launch) replayed twelve times equals
    twelve eager pushes bit for bit, drain included."""
    from audio_denoising_amd.pipeline import Denoiser, PipelinedStream
    p = _params("S")
    B = 256
    dn = Denoiser(_model(dev, 5), p.sample_rate, p.n_fft, p.hop, p.n_mels)
    g = torch.Generator().manual_seed(5)
    sig = (0.3 * torch.randn(B, 12 * p.hop, generator=g)).clamp(-1, 1).to(dev)
    outs = {}
    for mode in ("eager", "graph"):
        ps = PipelinedStream(dn, B, seed=3)
        ps.set_depth(4)
        hop = torch.empty(B, p.hop, device=dev)
        out = torch.empty(B, p.hop, device=dev)
        step = ps.graph_step(hop, out) if mode == "graph" else None
        res = []
        for i in range(12):
            hop.copy_(sig[:, i * p.hop:(i + 1) * p.hop])
            if step is not None:
                step.replay()
            else:
                ps.push_(hop, out)
            res.append(out.clone())
        res.append(ps.flush())
        outs[mode] = torch.cat(res, 1)
    assert torch.equal(outs["eager"], outs["graph"]) and outs["eager"].abs().max().item() > 1e-3
    assert not outs["eager"][:, :4 * p.hop].any()          # priming push + four hops of pipeline before the first samples come out
