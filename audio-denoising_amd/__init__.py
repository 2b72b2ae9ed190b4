"""MI355X-native per-hop speech-denoising path (STFT -> mel -> GRUUNet2 -> inverse mel -> Griffin-Lim).

Host side in Python on PyTorch-ROCm tensors; all arithmetic in hand-written HIP kernels for gfx950
behind the C ABI of ``include/dn_denoise.h`` (``lib/libdn_denoise.so``).  Import as
``audio_denoising_amd`` (alias module at the repo root) -- or, for the reference's own import line
``from gruunet2 import GRUUNet2`` (app3.py:38), through the repo-root ``gruunet2.py``.
"""
__version__ = "0.1.0"
