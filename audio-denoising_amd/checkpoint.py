"""Checkpoint reader / converter for the reference's GRUUNet2 checkpoints.

The reference writes ``{config, model_state_dict, optimizer_state_dict, ...}`` with ``torch.save``
(app.py:75-91) and its loader tolerates a few spellings (app3.py:59-97): the constructor arguments under
``hparams`` or ``config``, the weights under ``model_state_dict`` or ``state_dict`` or as a bare dict of
tensors.  This module accepts the same spellings without needing any reference code, and converts to the
flat fp32 blob + JSON config the C ABI takes (``dn_model_create``).  Host-side plumbing only.
"""
from __future__ import annotations

import json
import os

import torch

from .gruunet2 import GRUUNet2

CTOR_ARGS = ("num_compressed_bins", "in_size", "hidden_sizes", "kernel_sizes", "strides", "paddings", "num_gaussians")
REQUIRED = CTOR_ARGS[:-1]


def split_checkpoint(obj, default_config: dict | None = None):
    """-> (constructor kwargs, state_dict).  Raises ValueError where the reference's loader would give up
    (app3.py:82-83, 107-108)."""
    config, sd = None, None
    if isinstance(obj, dict):
        for key in ("hparams", "config"):                       # app3.py:62-65
            if isinstance(obj.get(key), dict):
                config = obj[key]
                break
        for key in ("model_state_dict", "state_dict"):          # app3.py:67-70
            if key in obj:
                sd = obj[key]
                break
        if sd is None:                                          # app3.py:71-74
            rest = {k: v for k, v in obj.items() if k not in ("hparams", "config", "last_epoch")}
            if rest and all(isinstance(v, torch.Tensor) for v in rest.values()):
                sd = rest
    elif hasattr(obj, "state_dict") and callable(obj.state_dict):   # app3.py:75-78
        sd = obj.state_dict()
        config = getattr(obj, "hparams", None) or getattr(obj, "config", None)
    if sd is None:
        raise ValueError("no state_dict found in the checkpoint")
    if config is None:
        config = default_config                                 # app3.py:85-86
    if config is None:
        raise ValueError("no constructor arguments ('hparams'/'config') found and no default given")
    kwargs = {k: config[k] for k in CTOR_ARGS if k in config}   # app3.py:88-97
    missing = [k for k in REQUIRED if k not in kwargs]
    if missing:
        raise ValueError(f"checkpoint config lacks {missing}")
    return kwargs, sd


def load_model(path: str, device=None, default_config: dict | None = None, num_compressed_bins: int | None = None) -> GRUUNet2:
    """torch.load (weights_only) + GRUUNet2(**config) + load_state_dict + eval [+ to(device)]: app3.py:59-116.
    ``num_compressed_bins`` overrides the stored value (the checkpoints store 4 = 64 mels; 80 mels need 5, and
    every conv is length-agnostic -- SURVEY.md section 0 row 9)."""
    obj = torch.load(path, map_location="cpu", weights_only=True)
    kwargs, sd = split_checkpoint(obj, default_config)
    if num_compressed_bins is not None:
        kwargs["num_compressed_bins"] = num_compressed_bins
    model = GRUUNet2(**kwargs)
    model.load_state_dict(sd)
    model.eval()
    return model.to(device) if device is not None else model


def export_flat(path_or_obj, out_prefix: str, default_config: dict | None = None) -> tuple[str, str]:
    """Write ``<prefix>.bin`` (state_dict flattened in key order, fp32: what dn_model_create takes) and
    ``<prefix>.json`` (constructor arguments + key/shape manifest)."""
    obj = torch.load(path_or_obj, map_location="cpu", weights_only=True) if isinstance(path_or_obj, (str, os.PathLike)) else path_or_obj
    kwargs, sd = split_checkpoint(obj, default_config)
    blob = torch.cat([v.detach().reshape(-1).to(torch.float32) for v in sd.values()]).contiguous()
    blob.numpy().tofile(out_prefix + ".bin")
    meta = dict(config={k: (list(v) if isinstance(v, (tuple, list)) else v) for k, v in kwargs.items()},
                keys=[[k, list(v.shape)] for k, v in sd.items()], n_floats=int(blob.numel()))
    with open(out_prefix + ".json", "w") as f:
        json.dump(meta, f, indent=1)
    return out_prefix + ".bin", out_prefix + ".json"
