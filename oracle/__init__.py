"""CPU oracle for the per-hop denoising path.  TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the arithmetic of the reference's hot path
(belacks/audio-denoising: app3.py:178-226 calling gruunet2.GRUUNet2.forward)
so the HIP kernels can be checked against it.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product package (``audio-denoising_amd``) never does: it fails
loudly when the HIP library is missing.

Pinning status
--------------
* ``model_ref``  -- PINNED.  Checked against golden vectors generated from the
  reference's own ``gruunet2.GRUUNet2`` + ``saves/GRUUNet2-dari_tult*``
  checkpoints (``oracle/make_golden.py``, fixtures in ``tests/golden/``).
* ``dsp_ref`` / ``dsp_np64`` / ``pipeline_ref`` -- PARITY UNPINNED.  The STFT,
  mel, inverse-mel and Griffin-Lim arithmetic of the reference lives in the
  third-party dependency ``torchaudio==2.6.0`` (requirements.txt:4), which is
  neither vendored under /root/reference nor installed here, and the
  reference has no tests or fixtures for these stages.  ``dsp_ref`` restates
  torchaudio's published semantics over first-party ``torch.stft`` /
  ``torch.istft`` / ``torch.linalg.lstsq``; ``dsp_np64`` is an independent
  float64 numpy implementation written separately (double-entry check); scipy.signal / scipy.linalg and
  HuggingFace ``transformers.audio_utils`` (mel filterbank, STFT) are further opinions (tests/test_oracle_dsp.py).
  ``pipeline_np64`` evaluates the whole hop in float64 on the fp32 constants: the yardstick that tells
  rounding from error (tests/golden/metric_f64_B256.npz, its model stage run by the reference's own class in double).
"""
