"""DSP half of the oracle: PARITY UNPINNED by the reference (torchaudio absent,
no reference tests).  Double-entry check: the torch restatement
(oracle/dsp_ref.py) against the independent float64 numpy implementation
(oracle/dsp_np64.py), plus size-independent properties."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import dsp_np64, dsp_ref, pipeline_ref

PARAMS = [pipeline_ref.PARAMS_S, pipeline_ref.PARAMS_R1, pipeline_ref.PARAMS_R2]


@pytest.mark.parametrize("p", PARAMS, ids=["S", "R1", "R2"])
def test_frame_gives_three_columns_and_matches_np64(p):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, p.n_fft, generator=g)
    s = dsp_ref.spectrogram(x, p.n_fft, p.hop)
    assert s.shape == (3, p.n_stft, 3)
    s64 = dsp_np64.stft(x.numpy(), p.n_fft, p.hop)
    assert np.abs(s.numpy() - s64).max() <= 2e-4 * np.abs(s64).max()


@pytest.mark.parametrize("p", PARAMS, ids=["S", "R1", "R2"])
def test_filterbank_matches_np64_and_is_full_rank(p):
    fb = dsp_ref.melscale_fbanks(p.n_stft, p.n_mels, p.sample_rate)
    fb64 = dsp_np64.mel_fbanks(p.n_stft, p.n_mels, p.sample_rate)
    assert fb.shape == (p.n_stft, p.n_mels)
    assert np.abs(fb.numpy() - fb64).max() <= 5e-5
    assert np.linalg.matrix_rank(fb64) == p.n_mels
    assert (fb.sum(0) > 0).all()


@pytest.mark.parametrize("p", PARAMS, ids=["S", "R1", "R2"])
def test_inverse_mel_lstsq_is_min_norm_pinv(p):
    fb = dsp_ref.melscale_fbanks(p.n_stft, p.n_mels, p.sample_rate)
    g = torch.Generator().manual_seed(2)
    mel = torch.rand(4, p.n_mels, 3, generator=g) * 20
    a = dsp_ref.inverse_mel_scale(mel, fb).numpy()
    b = dsp_np64.inverse_mel_scale(mel.numpy(), fb.numpy())
    assert np.abs(a - b).max() <= 2e-4


@pytest.mark.parametrize("p", PARAMS, ids=["S", "R1", "R2"])
def test_inverse_mel_in_factors_is_the_same_operator(p):
    """What the device's factored inverse mel relies on (dn_plan.hpp: ginv_band / fb2): pinv(fb^T) = fb (fb^T fb)^-1, every bin lies in
    at most two triangles, and (fb^T fb)^-1 has nothing above fp32 resolution beyond +-16 diagonals -- so fb (band16(G^-1) mel) IS the
    min-norm least-squares solution the reference's InverseMelScale computes, to fp32 rounding."""
    fb = dsp_ref.melscale_fbanks(p.n_stft, p.n_mels, p.sample_rate).numpy().astype(np.float64)
    assert ((fb != 0).sum(1) <= 2).all()
    ginv = np.linalg.inv(fb.T @ fb)
    i, j = np.indices(ginv.shape)
    assert np.abs(ginv[np.abs(i - j) > 16]).max() <= 1e-8 * np.abs(ginv).max()
    band = np.where(np.abs(i - j) <= 16, ginv, 0.0)
    g = torch.Generator().manual_seed(12)
    mel = (torch.rand(6, p.n_mels, 3, generator=g) * 20).numpy().astype(np.float64)
    ref = dsp_np64.inverse_mel_scale(mel.astype(np.float32), fb.astype(np.float32))          # relu(lstsq), float64 inside
    fac = np.maximum(np.einsum("km,bmt->bkt", fb, np.einsum("mn,bnt->bmt", band, mel)), 0.0)
    assert np.abs(fac - ref).max() <= 1e-6 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("p", [pipeline_ref.PARAMS_S, pipeline_ref.PARAMS_R1], ids=["S", "R1"])
def test_istft_inverts_stft(p):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, p.n_fft, generator=g)
    y = dsp_ref.inverse_spectrogram(dsp_ref.spectrogram(x, p.n_fft, p.hop), p.n_fft, p.hop)
    assert y.shape == x.shape and (y - x).abs().max() <= 1e-4
    y64 = dsp_np64.istft(dsp_np64.stft(x.numpy(), p.n_fft, p.hop), p.n_fft, p.hop)
    assert np.abs(y64 - x.numpy()).max() <= 1e-10


def test_griffinlim_matches_np64_with_shared_init():
    p = pipeline_ref.PARAMS_S
    g = torch.Generator().manual_seed(4)
    x = 0.1 * torch.randn(2, p.n_fft, generator=g)
    mag = dsp_ref.spectrogram(x, p.n_fft, p.hop).abs()
    init = torch.rand(mag.shape, dtype=torch.complex64, generator=g)
    y = dsp_ref.griffinlim(mag, p.n_fft, p.hop, init_angles=init).numpy()
    y64 = dsp_np64.griffinlim(mag.numpy(), p.n_fft, p.hop, init.numpy())
    assert y.shape == (2, p.n_fft)
    rms = np.sqrt(np.mean((y - y64) ** 2))
    assert rms <= 1e-3            # north_star waveform tolerance; fp32 vs fp64 is ~1e-5


def test_golden_dsp_fixture_is_reproduced():
    """The committed dsp_S.npz freezes the restatement (and documents the
    silent-frame branch app3.py:182-186)."""
    g = load_golden("dsp_S.npz")
    p = pipeline_ref.PARAMS_S
    mi, peak = pipeline_ref.analysis(torch.from_numpy(g["frames"]), p, torch.from_numpy(g["fb"]))
    assert np.abs(mi.numpy() - g["model_input"]).max() <= 1e-4
    assert peak[4] == 1.0 and peak[5] == 1.0


@pytest.mark.parametrize("p", [pipeline_ref.PARAMS_S, pipeline_ref.PARAMS_R1], ids=["S", "R1"])
def test_third_opinion_scipy_signal(p):
    """A third, unrelated implementation: scipy.signal.stft / istft (even-extension boundary == reflect padding, periodic
    Hann, 50 % overlap) agrees with the restatement of torch.stft / torch.istft once its 1/sum(window) scaling is undone."""
    import scipy.signal as ss
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, p.n_fft, generator=g)
    win = ss.get_window("hann", p.n_fft, fftbins=True)
    assert np.abs(win - dsp_np64.hann(p.n_fft)).max() <= 1e-12
    _, _, Z = ss.stft(x.numpy().astype(np.float64), nperseg=p.n_fft, noverlap=p.n_fft - p.hop, window=win, boundary="even",
                      padded=False, return_onesided=True)
    Z = Z * win.sum()
    s = dsp_ref.spectrogram(x, p.n_fft, p.hop).numpy()
    assert Z.shape == s.shape and np.abs(Z - s).max() <= 2e-4 * np.abs(Z).max()
    _, y = ss.istft(Z / win.sum(), nperseg=p.n_fft, noverlap=p.n_fft - p.hop, window=win, boundary=True, input_onesided=True)
    y_ref = dsp_ref.inverse_spectrogram(torch.from_numpy(s), p.n_fft, p.hop).numpy()
    assert np.abs(y[:, :p.n_fft] - y_ref).max() <= 1e-4


def test_real_clip_fixture_is_reproduced_by_the_oracle():
    """tests/golden/clip_S.npz (BASELINE configs[0]: one 3 s / 16 kHz clip, 93 hops, batch 1) is what the oracle's streaming
    loop gives today -- the CPU 'plumbing' run of configs[0]; first 12 hops re-run here to keep the CPU tier quick."""
    from oracle import model_ref
    from oracle.make_clip_golden import clip_init_angles
    import os
    from conftest import GOLDEN
    p = pipeline_ref.PARAMS_S
    g = load_golden("clip_S.npz")
    assert g["signal_s16"].shape == (p.n_fft + p.hop * 92,) and g["out"].shape == (1, 93 * p.hop)
    sig = torch.from_numpy(g["signal_s16"].astype(np.float32) / np.float32(32767))[None]
    sd = model_ref.unflatten_weights(np.fromfile(os.path.join(GOLDEN, "weights_dari_tult.bin"), dtype=np.float32))
    st = pipeline_ref.StreamRef(sd, p, 1)
    n = 12
    with torch.no_grad():
        y = st.push(sig[:, :p.n_fft + p.hop * (n - 1)], init_angles_per_hop=[clip_init_angles(f) for f in range(n)])
    assert np.abs(y.numpy() - g["out"][:, :n * p.hop]).max() <= 1e-5


@pytest.mark.parametrize("p", [pipeline_ref.PARAMS_S, pipeline_ref.PARAMS_R1], ids=["S", "R1"])
def test_third_opinion_griffinlim_inner_step_and_min_norm_inverse_mel(p):
    """scipy as the third opinion for the two stages the float64 double-entry check alone covered so far:
    (1) one Griffin-Lim inner step -- istft of a NON-consistent 3-column spectrogram (random phases on a magnitude), then stft of
        the result -- through scipy.signal.istft / stft against the restatement of torch.istft / torch.stft;
    (2) InverseMelScale's lstsq(driver='gels') on the underdetermined system == the minimum-norm solution scipy.linalg.lstsq
        (gelsd, SVD based) and scipy.linalg.pinv give, at both parameter sets (R1 = the app's own 769 x 64 system)."""
    import scipy.linalg as sl
    import scipy.signal as ss
    g = torch.Generator().manual_seed(21)
    mag = torch.rand(2, p.n_stft, 3, generator=g) * 5.0
    ang = torch.rand(2, p.n_stft, 3, dtype=torch.complex64, generator=g)
    X = (ang * mag)
    inv = dsp_ref.inverse_spectrogram(X, p.n_fft, p.hop)                  # torch.istft ignores Im X[0], Im X[N/2] (C2R convention)
    reb = dsp_ref.spectrogram(inv, p.n_fft, p.hop).numpy()
    win = ss.get_window("hann", p.n_fft, fftbins=True)
    Xn = X.numpy().astype(np.complex128)
    Xn[:, 0, :] = Xn[:, 0, :].real
    Xn[:, -1, :] = Xn[:, -1, :].real
    _, y = ss.istft(Xn / win.sum(), nperseg=p.n_fft, noverlap=p.n_fft - p.hop, window=win, boundary=True, input_onesided=True)
    y = y[:, :p.n_fft]
    assert np.abs(y - inv.numpy()).max() <= 1e-4 * max(1.0, np.abs(y).max())
    _, _, Z = ss.stft(y, nperseg=p.n_fft, noverlap=p.n_fft - p.hop, window=win, boundary="even", padded=False, return_onesided=True)
    Z = Z * win.sum()
    assert Z.shape == reb.shape and np.abs(Z - reb).max() <= 2e-4 * np.abs(Z).max()
    # (2)
    fb = dsp_ref.melscale_fbanks(p.n_stft, p.n_mels, p.sample_rate)
    mel = torch.rand(3, p.n_mels, 3, generator=g) * 20
    ours = dsp_ref.inverse_mel_scale(mel, fb).numpy()                    # relu(lstsq(fb^T, mel))
    A = fb.numpy().astype(np.float64).T                                  # (M, K): underdetermined, full row rank
    for b in range(3):
        sol, _, rank, _ = sl.lstsq(A, mel[b].numpy().astype(np.float64), lapack_driver="gelsd")
        assert rank == p.n_mels
        assert np.abs(np.maximum(sol, 0.0) - ours[b]).max() <= 2e-4
        assert np.abs(sol - sl.pinv(A) @ mel[b].numpy().astype(np.float64)).max() <= 1e-8


def test_filterbank_and_stft_agree_with_a_fourth_independent_implementation():
    """HuggingFace `transformers.audio_utils` (installed here, unrelated to this repo and to torchaudio's code base) documents its `mel_filter_bank(norm=None,
    mel_scale="htk")` and `spectrogram(center=True, pad_mode="reflect")` as equivalents of torchaudio's / librosa's: the oracle's restatement of
    `melscale_fbanks` (app3.py:139-148) agrees with it to fp32 rounding of the fp32 formula at all three parameter sets, the 3-column STFT of a frame
    (app3.py:191) to 2e-7 relative.  Still not a pin to the reference's torchaudio==2.6.0 -- a fourth opinion on what its published algorithm is."""
    au = pytest.importorskip("transformers.audio_utils")
    for p in (pipeline_ref.PARAMS_S, pipeline_ref.PARAMS_R1, pipeline_ref.PARAMS_R2):
        fb = dsp_ref.melscale_fbanks(p.n_stft, p.n_mels, p.sample_rate).numpy()
        other = au.mel_filter_bank(num_frequency_bins=p.n_stft, num_mel_filters=p.n_mels, min_frequency=0.0, max_frequency=float(p.sample_rate // 2),
                                   sampling_rate=p.sample_rate, norm=None, mel_scale="htk")
        assert other.shape == fb.shape and np.abs(fb - other).max() <= 1e-5
        x = (0.1 * torch.randn(p.n_fft, generator=torch.Generator().manual_seed(3))).numpy().astype(np.float64)
        s = au.spectrogram(x, au.window_function(p.n_fft, "hann", periodic=True), frame_length=p.n_fft, hop_length=p.hop, fft_length=p.n_fft, power=None,
                           center=True, pad_mode="reflect", onesided=True)
        ref = dsp_ref.spectrogram(torch.from_numpy(x).float()[None], p.n_fft, p.hop).numpy()[0]
        assert s.shape == ref.shape == (p.n_stft, 3) and np.abs(s - ref).max() <= 5e-7 * np.abs(ref).max()
