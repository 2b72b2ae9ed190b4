"""The model half of the oracle is PINNED: oracle/model_ref.py must reproduce
golden vectors produced by the reference's own gruunet2.GRUUNet2
(oracle/make_golden.py; gruunet2.py:246-306)."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden
from oracle import model_ref

CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "cell_*_B*_T*_F*.npz")))


def _sd(short):
    return model_ref.unflatten_weights(np.fromfile(os.path.join(GOLDEN, f"weights_{short}.bin"), dtype=np.float32))


@pytest.mark.parametrize("name", CASES)
def test_forward_matches_reference_golden(name):
    short = "dari_tult2" if "dari_tult2" in name else "dari_tult"
    g = load_golden(name)
    out, hx = model_ref.forward(_sd(short), torch.from_numpy(g["x"]), torch.from_numpy(g["hx0"]))
    # tolerance: fp32 conv summation order differs between runs/threads (~1e-6)
    assert np.abs(out.numpy() - g["out"]).max() <= 1e-5
    assert np.abs(hx.numpy() - g["hx1"]).max() <= 1e-5


def test_intermediates_match_reference_hooks():
    g = load_golden("cell_dari_tult_B4_T3_F80.npz")
    inter = {}
    model_ref.cell_step(_sd("dari_tult"), torch.from_numpy(g["x"][:, 0]), torch.from_numpy(g["hx0"]), inter)
    for k in ("d0", "d1", "d2", "d3", "gate_h"):
        assert np.abs(inter[k].numpy() - g[k]).max() <= 1e-5, k


def test_conventions_2d_input_and_default_hx():
    g = load_golden("cell_dari_tult_conventions.npz")
    sd = _sd("dari_tult")
    o2, h2 = model_ref.forward(sd, torch.from_numpy(g["x2"]))
    assert o2.shape == (3, 64) and h2.shape == (1, 17, 4)       # gruunet2.py:291-293,304-305
    assert np.abs(o2.numpy() - g["out2"]).max() <= 1e-5
    o3, h3 = model_ref.forward(sd, torch.from_numpy(g["x3"]))
    assert np.abs(o3.numpy() - g["out3"]).max() <= 1e-5 and np.abs(h3.numpy() - g["hx3"]).max() <= 1e-5


def test_chain_of_20_hops_carries_state():
    g = load_golden("cell_dari_tult_chain20_F80.npz")
    sd = _sd("dari_tult")
    hx = None
    for h in range(20):
        o, hx = model_ref.forward(sd, torch.from_numpy(g["x"][h]), hx, num_compressed_bins=5)
        assert np.abs(o.numpy() - g["out"][h]).max() <= 2e-5
    assert np.abs(hx.numpy() - g["hx_final"]).max() <= 2e-5


def test_smear_tables_match_reference():
    g = load_golden("smear.npz")
    off = torch.linspace(0, 1, 6)
    for L in (80, 64, 40, 32, 20, 16, 10, 8, 5, 4):
        assert np.abs(model_ref.smear_table(off, L).numpy() - g[f"L{L}"]).max() <= 1e-7


# ------------------------------------------------------------------ sibling model MOMO3
MOMO_CASES = ["momo3_B1_T3_F22.npz", "momo3_B4_T3_F22.npz", "momo3_B256_T3_F22.npz", "momo3_B3_T7_F24.npz", "momo3_B2_T1_F23.npz"]


def _momo_sd():
    import os
    from conftest import GOLDEN
    from oracle import momo_ref
    return momo_ref.unflatten_weights(np.fromfile(os.path.join(GOLDEN, "weights_momo3_4d4ea0.bin"), dtype=np.float32))


@pytest.mark.parametrize("name", MOMO_CASES)
def test_momo3_restatement_matches_reference_golden(name):
    """oracle/momo_ref.py against vectors produced by the reference's own momo3.MOMO3 (oracle/make_momo_golden.py): PINNED."""
    from oracle import momo_ref
    g = load_golden(name)
    prev = torch.from_numpy(g["prev"]) if "prev" in g.files else None
    with torch.no_grad():
        out, hx = momo_ref.forward(_momo_sd(), torch.from_numpy(g["x"]), torch.from_numpy(g["hx0"]), prev)
    assert np.abs(out.numpy() - g["out"]).max() <= 1e-5 and np.abs(hx.numpy() - g["hx1"]).max() <= 1e-5


def test_momo3_conventions_and_carried_prev():
    from oracle import momo_ref
    g = load_golden("momo3_conventions.npz")
    sd = _momo_sd()
    with torch.no_grad():
        o2, h2 = momo_ref.forward(sd, torch.from_numpy(g["x2"]))             # (T,F) input, hx=None
        assert o2.shape == (3, 22) and h2.shape == (1, 16, 3)
        assert np.abs(o2.numpy() - g["out2"]).max() <= 1e-5
        hx, prev = None, None
        for h in range(12):
            x = torch.from_numpy(g["xs"][h])
            o, hx = momo_ref.forward(sd, x, hx, prev)
            prev = x[:, -1:, :].clone()
            assert np.abs(o.numpy() - g["outs"][h]).max() <= 1e-5
    assert np.abs(hx.numpy() - g["hx_final"]).max() <= 1e-5
