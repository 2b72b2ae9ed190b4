"""`from gruunet2 import GRUUNet2` -- the import line of the reference's callers (app3.py:38,
app2.py:37, server.py:34, app.py:32) -- resolves to the MI355X-native implementation."""
from audio_denoising_amd.gruunet2 import GRUUNet2  # noqa: F401

__all__ = ["GRUUNet2"]
