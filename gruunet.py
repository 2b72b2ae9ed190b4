"""`from gruunet import GRUUNet` (server.py:33): the reference's GRUUNet is GRUUNet2 under another name
(gruunet.py:246-299 vs gruunet2.py:246-306 differ only in the class name and a dead `prev` argument), with the same
state_dict keys -- so it runs on the same HIP kernels."""
from audio_denoising_amd.gruunet2 import GRUUNet2


class GRUUNet(GRUUNet2):
    pass


__all__ = ["GRUUNet"]
