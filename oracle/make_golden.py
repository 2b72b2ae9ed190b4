#!/usr/bin/env python3
"""Generate the committed fixtures under tests/golden/ (TEST INFRASTRUCTURE).

Run ONLY in the build container, where /root/reference is mounted:

    python oracle/make_golden.py

* weights_*.bin / weights_manifest.json, cell_*.npz, smear.npz come from the
  REFERENCE itself: the unmodified /root/reference/gruunet2.py is imported
  (its unused top-level imports av / sounddevice / torchaudio are absent here,
  so placeholder modules are registered for exactly those names first) and run
  on seeded inputs with the reference's own checkpoints.
* dsp_*.npz come from oracle/dsp_ref.py (torchaudio is not available): they
  freeze the restatement, they do NOT pin it to the reference ("parity
  unpinned", oracle/__init__.py).

Fixtures are data only (inputs and expected outputs); no reference source or
bytecode is written anywhere.
"""
from __future__ import annotations

import json
import os
import sys
import tempfile

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")
REF = "/root/reference"
sys.path.insert(0, REPO)


def import_reference_model():
    from unittest.mock import MagicMock
    for name in ("av", "sounddevice", "torchaudio", "torchaudio.transforms"):
        sys.modules.setdefault(name, MagicMock(name=name))
    sys.dont_write_bytecode = True
    scratch = tempfile.mkdtemp(prefix="dn_golden_")      # utils.py:60 creates ./cache in CWD
    os.chdir(scratch)
    sys.path.insert(0, REF)
    import gruunet2  # noqa: the reference's own file
    return gruunet2


def load_reference(gruunet2, name: str, num_compressed_bins: int):
    ck = torch.load(os.path.join(REF, "saves", name, "checkpoint.pth"), map_location="cpu", weights_only=True)
    cfg = dict(ck["config"])
    cfg["num_compressed_bins"] = num_compressed_bins
    m = gruunet2.GRUUNet2(**cfg)
    m.load_state_dict(ck["model_state_dict"])
    m.eval()
    return m, ck


def main():
    os.makedirs(GOLD, exist_ok=True)
    gruunet2 = import_reference_model()
    from oracle import dsp_ref, model_ref, pipeline_ref

    # ---- G1: weights -------------------------------------------------------
    manifest = {}
    for name in ("GRUUNet2-dari_tult", "GRUUNet2-dari_tult2", "GRUUNet2-good"):
        m, ck = load_reference(gruunet2, name, 4)
        sd = ck["model_state_dict"]
        assert [(k, tuple(v.shape)) for k, v in sd.items()] == [(k, s) for k, s in model_ref.STATE_KEYS]
        blob = torch.cat([v.reshape(-1).float() for v in sd.values()]).numpy()
        assert blob.size == model_ref.N_WEIGHT_FLOATS
        short = name.replace("GRUUNet2-", "")
        blob.tofile(os.path.join(GOLD, f"weights_{short}.bin"))
        manifest[short] = dict(config={k: (list(v) if isinstance(v, tuple) else v) for k, v in ck["config"].items()},
                               keys=[[k, list(s)] for k, s in model_ref.STATE_KEYS],
                               n_floats=int(blob.size), source=f"saves/{name}/checkpoint.pth")
    with open(os.path.join(GOLD, "weights_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)

    # ---- G2: cell forward from the reference class ---------------------------
    cases = [(1, 3, 64, 4), (4, 3, 64, 4), (256, 3, 64, 4), (4, 3, 80, 5), (256, 3, 80, 5), (2, 1, 64, 4), (3, 7, 80, 5)]
    for ck_name in ("GRUUNet2-dari_tult", "GRUUNet2-dari_tult2"):
        short = ck_name.replace("GRUUNet2-", "")
        for (B, T, F, C) in cases:
            if short != "dari_tult" and B == 256:
                continue
            g = torch.Generator().manual_seed(1000 + B * 7 + T * 3 + F)
            x = torch.rand(B, T, F, generator=g) * 6.0
            hx = torch.randn(B, 17, C, generator=g) * 0.5
            m, _ = load_reference(gruunet2, ck_name, C)
            inter = {}
            hooks = []
            cell = m.cell

            def grab(key):
                def fn(_mod, _inp, out):
                    inter.setdefault(key, []).append(out)
                return fn
            hooks.append(cell.input_gate.register_forward_hook(grab("gate_x")))
            hooks.append(cell.reset_gate.register_forward_hook(grab("gate_h")))
            with torch.no_grad():
                out, hx1 = m(x, hx)
            for h in hooks:
                h.remove()
            save = dict(x=x.numpy(), hx0=hx.numpy(), out=out.numpy(), hx1=hx1.numpy())
            if B <= 4:   # first-step intermediates for kernel-level debugging
                gx = inter["gate_x"][0]
                save.update(d0=gx[1].numpy(), d1=gx[2].numpy(), d2=gx[3].numpy(), d3=gx[4].numpy(),
                            gate_h=inter["gate_h"][0].numpy())
            np.savez(os.path.join(GOLD, f"cell_{short}_B{B}_T{T}_F{F}.npz"), **save)

    # hx=None and 2-D input conventions (gruunet2.py:291-305)
    m, _ = load_reference(gruunet2, "GRUUNet2-dari_tult", 4)
    g = torch.Generator().manual_seed(77)
    x2 = torch.rand(3, 64, generator=g) * 6.0
    with torch.no_grad():
        o2, h2 = m(x2)
        x3 = torch.rand(2, 3, 64, generator=g) * 6.0
        o3, h3 = m(x3)
    np.savez(os.path.join(GOLD, "cell_dari_tult_conventions.npz"), x2=x2.numpy(), out2=o2.numpy(), hx2=h2.numpy(),
             x3=x3.numpy(), out3=o3.numpy(), hx3=h3.numpy())

    # 20-hop chained state carry (hx fed back, T=3 per hop), F=80
    m, _ = load_reference(gruunet2, "GRUUNet2-dari_tult", 5)
    g = torch.Generator().manual_seed(4242)
    xs = torch.rand(20, 8, 3, 80, generator=g) * 6.0
    hx = None
    outs = []
    with torch.no_grad():
        for h in range(20):
            o, hx = m(xs[h], hx)
            outs.append(o)
    np.savez(os.path.join(GOLD, "cell_dari_tult_chain20_F80.npz"), x=xs.numpy(), out=torch.stack(outs).numpy(), hx_final=hx.numpy())

    # ---- G3: smear tables from the reference's GaussianSmearing ---------------
    gs = gruunet2.GaussianSmearing(num_gaussians=6)
    sm = {}
    for L in (80, 64, 40, 32, 20, 16, 10, 8, 5, 4):
        sm[f"L{L}"] = gs(torch.linspace(0, 1, L)).t().contiguous().numpy()     # (6, L)
    np.savez(os.path.join(GOLD, "smear.npz"), coeff=np.float64(gs.coeff), **sm)

    # ---- G4: DSP fixtures from the restatement (UNPINNED) ---------------------
    for tag, p in (("S", pipeline_ref.PARAMS_S), ("R2", pipeline_ref.PARAMS_R2), ("R1", pipeline_ref.PARAMS_R1)):
        g = torch.Generator().manual_seed(1234)
        B = 6
        frames = 0.1 * torch.randn(B, p.n_fft, generator=g)
        frames[4] = 0.0                                  # silent stream: peak <= 1e-6 branch
        frames[5] *= 1e-8
        fb = dsp_ref.melscale_fbanks(p.n_stft, p.n_mels, p.sample_rate)
        sd = model_ref.unflatten_weights(np.fromfile(os.path.join(GOLD, "weights_dari_tult.bin"), dtype=np.float32))
        hx0 = torch.zeros(B, 17, p.num_compressed_bins)
        ga = torch.Generator().manual_seed(4321)
        init = torch.rand(B, p.n_stft, 3, dtype=torch.complex64, generator=ga)
        with torch.no_grad():
            r = pipeline_ref.process_frame(sd, frames, hx0, p, fb, init_angles=init)
            spec = dsp_ref.spectrogram(frames, p.n_fft, p.hop)
        np.savez(os.path.join(GOLD, f"dsp_{tag}.npz"), frames=frames.numpy(), fb=fb.numpy(), spec=spec.numpy(),
                 init_angles=init.numpy(), model_input=r["model_input"].numpy(), predicted_diff=r["predicted_diff"].numpy(),
                 mel_mag=r["mel_mag"].numpy(), lin_mag=r["lin_mag"].numpy(), out=r["out"].numpy(), hx=r["hx"].numpy(),
                 peak=r["peak"].numpy())

    # streaming: 4 streams x 10 hops, S params
    p = pipeline_ref.PARAMS_S
    g = torch.Generator().manual_seed(99)
    n_hops = 10
    sig = 0.1 * torch.randn(4, p.n_fft + p.hop * (n_hops - 1), generator=g)
    ga = torch.Generator().manual_seed(4321)
    inits = [torch.rand(4, p.n_stft, 3, dtype=torch.complex64, generator=ga) for _ in range(n_hops)]
    sd = model_ref.unflatten_weights(np.fromfile(os.path.join(GOLD, "weights_dari_tult.bin"), dtype=np.float32))
    st = pipeline_ref.StreamRef(sd, p, 4)
    with torch.no_grad():
        y = st.push(sig, init_angles_per_hop=inits)
    np.savez(os.path.join(GOLD, "stream_S.npz"), signal=sig.numpy(), init_angles=torch.stack(inits).numpy(),
             out=y.numpy(), ola=st.ola.numpy(), hx=st.hx.numpy())
    # streaming at the app's own parameters (app3.py:29-33): 3 streams x 6 hops
    p = pipeline_ref.PARAMS_R1
    g = torch.Generator().manual_seed(199)
    n_hops = 6
    sig = 0.1 * torch.randn(3, p.n_fft + p.hop * (n_hops - 1), generator=g)
    ga = torch.Generator().manual_seed(4321)
    inits = [torch.rand(3, p.n_stft, 3, dtype=torch.complex64, generator=ga) for _ in range(n_hops)]
    sd = model_ref.unflatten_weights(np.fromfile(os.path.join(GOLD, "weights_dari_tult2.bin"), dtype=np.float32))   # app3.py:13
    st = pipeline_ref.StreamRef(sd, p, 3)
    with torch.no_grad():
        y = st.push(sig, init_angles_per_hop=inits)
    np.savez(os.path.join(GOLD, "stream_R1.npz"), signal=sig.numpy(), init_angles=torch.stack(inits).numpy(),
             out=y.numpy(), ola=st.ola.numpy(), hx=st.hx.numpy())
    # server.py request loop (R2 parameters, checkpoint GRUUNet2-good): 3 consecutive chunks of 9 hops, 2 streams
    from oracle import server_ref
    p = pipeline_ref.PARAMS_R2
    g = torch.Generator().manual_seed(515)
    chunks = 0.1 * torch.randn(3, 2, 9 * p.hop, generator=g)
    sd = model_ref.unflatten_weights(np.fromfile(os.path.join(GOLD, "weights_good.bin"), dtype=np.float32))
    hx, outs = None, []
    with torch.no_grad():
        for c in range(3):
            r = server_ref.process_chunk(sd, chunks[c], hx, p)
            hx = r["hx"]
            outs.append(r["out"])
            if c == 0:
                first = r
    np.savez(os.path.join(GOLD, "server_R2.npz"), chunks=chunks.numpy(), out=torch.stack(outs).numpy(), hx=hx.numpy(),
             log_mel0=first["log_mel"].numpy(), model_out0=first["model_out"].numpy(), lin0=first["lin"].numpy(), spec0=first["spec"].numpy())
    print("fixtures written to", GOLD)
    for fn in sorted(os.listdir(GOLD)):
        print(f"  {fn:45s} {os.path.getsize(os.path.join(GOLD, fn)):>9d} B")


if __name__ == "__main__":
    main()
