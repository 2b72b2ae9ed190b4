#!/usr/bin/env python3
"""Golden vectors for the sibling model MOMO3 from the REFERENCE's own class (TEST INFRASTRUCTURE).

Run ONLY in the build container, where /root/reference is mounted:  python oracle/make_momo_golden.py

The unmodified /root/reference/momo3.py is imported (placeholder modules only for its unused imports av / sounddevice /
torchaudio, absent here) with the reference's checkpoint saves/MOMO3-4d4ea0 and run on seeded inputs; the flat weight
blob (state_dict order) and the input/output tensors are written to tests/golden/.  Data only."""
import os
import sys
import tempfile

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")
REF = "/root/reference"
sys.path.insert(0, REPO)


def main():
    from unittest.mock import MagicMock
    for name in ("av", "sounddevice", "torchaudio", "torchaudio.transforms"):
        sys.modules.setdefault(name, MagicMock(name=name))
    sys.dont_write_bytecode = True
    os.chdir(tempfile.mkdtemp(prefix="dn_golden_"))       # utils.py:60 creates ./cache in CWD
    sys.path.insert(0, REF)
    import momo3  # noqa: the reference's own file
    from oracle import momo_ref
    ck = torch.load(os.path.join(REF, "saves", "MOMO3-4d4ea0", "checkpoint.pth"), map_location="cpu", weights_only=True)
    cfg = dict(ck["config"])
    sd = ck["model_state_dict"]
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == list(momo_ref.STATE_KEYS)
    blob = torch.cat([v.reshape(-1).float() for v in sd.values()]).numpy()
    assert blob.size == momo_ref.N_WEIGHT_FLOATS
    blob.tofile(os.path.join(GOLD, "weights_momo3_4d4ea0.bin"))
    cases = [(1, 3, 22), (4, 3, 22), (256, 3, 22), (3, 7, 24), (2, 1, 23)]
    for (B, T, Fb) in cases:
        g = torch.Generator().manual_seed(2000 + B * 7 + T * 3 + Fb)
        x = torch.rand(B, T, Fb, generator=g) * 6.0
        C = momo_ref.compressed_bins(Fb)
        hx = torch.randn(B, 16, C, generator=g) * 0.5
        prev = torch.rand(B, 1, Fb, generator=g) * 6.0 if B == 4 else None       # one case continues a sequence (prev given)
        cfg_c = dict(cfg, num_compressed_bins=C)
        m = momo3.MOMO3(**cfg_c)
        m.load_state_dict(sd)
        m.eval()
        with torch.no_grad():
            out, hx1 = m(x, hx, prev=None if prev is None else prev.clone())
        save = dict(x=x.numpy(), hx0=hx.numpy(), out=out.numpy(), hx1=hx1.numpy())
        if prev is not None:
            save["prev"] = prev.numpy()
        np.savez(os.path.join(GOLD, f"momo3_B{B}_T{T}_F{Fb}.npz"), **save)
    # conventions: hx=None, 2-D input; a 12-hop chain with hx and prev carried by the caller
    m = momo3.MOMO3(**cfg)
    m.load_state_dict(sd)
    m.eval()
    g = torch.Generator().manual_seed(88)
    x2 = torch.rand(3, 22, generator=g) * 6.0
    xs = torch.rand(12, 5, 3, 22, generator=g) * 6.0
    with torch.no_grad():
        o2, h2 = m(x2)
        hx, prev, outs = None, None, []
        for h in range(12):
            o, hx = m(xs[h], hx, prev=prev)
            prev = xs[h][:, -1:, :].clone()
            outs.append(o)
    np.savez(os.path.join(GOLD, "momo3_conventions.npz"), x2=x2.numpy(), out2=o2.numpy(), hx2=h2.numpy(),
             xs=xs.numpy(), outs=torch.stack(outs).numpy(), hx_final=hx.numpy())
    for fn in sorted(os.listdir(GOLD)):
        if "momo3" in fn:
            print(f"  {fn:40s} {os.path.getsize(os.path.join(GOLD, fn)):>9d} B")


if __name__ == "__main__":
    main()
