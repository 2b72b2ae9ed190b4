"""`from momo3 import MOMO3` resolves here to the MI355X-native sibling model (reference: momo3.py:247-324)."""
from audio_denoising_amd.momo3 import MOMO3  # noqa: F401

__all__ = ["MOMO3"]
