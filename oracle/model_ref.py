"""CPU restatement of the reference GRUUNet2 forward (TEST INFRASTRUCTURE).

Follows /root/reference/gruunet2.py:
  * GaussianSmearing           gruunet2.py:54-68
  * DownBlocks / DownConvBlock gruunet2.py:71-79, 127-156
  * UpBlocks / UpConvBlock     gruunet2.py:81-96, 184-199
  * GRUUNetCell.forward        gruunet2.py:228-244
  * GRUUNet2._gruunet/forward  gruunet2.py:266-306

The op sequence is the one the reference executes (smear channels are built
and concatenated for every conv, no algebraic folding) so that this module can
also serve as the "reference op sequence" CPU baseline in bench.py.

Weights are addressed by the reference's ``state_dict`` key names.
PINNED: tests/test_oracle_model.py checks it against tests/golden/cell_*.npz,
which were produced by the reference's own class (oracle/make_golden.py).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

# state_dict order of the reference model (probed from the checkpoints; see
# SURVEY.md Appendix A.4).  (key, shape) with H=17, G=6.
STATE_KEYS = (
    ("cell.input_gate.downs.0.conv.weight", (17, 7, 3)),
    ("cell.input_gate.downs.0.conv.bias", (17,)),
    ("cell.input_gate.downs.1.conv.weight", (17, 23, 3)),
    ("cell.input_gate.downs.1.conv.bias", (17,)),
    ("cell.input_gate.downs.2.conv.weight", (17, 23, 3)),
    ("cell.input_gate.downs.2.conv.bias", (17,)),
    ("cell.input_gate.downs.3.conv.weight", (51, 23, 3)),
    ("cell.input_gate.downs.3.conv.bias", (51,)),
    ("cell.input_gate.gs.offset", (6,)),
    ("cell.reset_gate.downs.0.conv.weight", (51, 23, 3)),
    ("cell.reset_gate.downs.0.conv.bias", (51,)),
    ("cell.reset_gate.gs.offset", (6,)),
    ("cell.output_gate.ups.0.conv.weight", (23, 17, 3)),
    ("cell.output_gate.ups.0.conv.bias", (17,)),
    ("cell.output_gate.ups.1.conv.weight", (40, 17, 3)),
    ("cell.output_gate.ups.1.conv.bias", (17,)),
    ("cell.output_gate.ups.2.conv.weight", (40, 17, 3)),
    ("cell.output_gate.ups.2.conv.bias", (17,)),
    ("cell.output_gate.ups.3.conv.weight", (40, 1, 3)),
    ("cell.output_gate.ups.3.conv.bias", (1,)),
    ("cell.output_gate.gs.offset", (6,)),
)
N_WEIGHT_FLOATS = 15337


def unflatten_weights(blob: torch.Tensor) -> dict:
    """Flat fp32 blob (state_dict order) -> {key: tensor}."""
    blob = torch.as_tensor(blob).reshape(-1)
    assert blob.numel() == N_WEIGHT_FLOATS, blob.numel()
    sd, off = {}, 0
    for key, shape in STATE_KEYS:
        n = 1
        for s in shape:
            n *= s
        sd[key] = blob[off:off + n].reshape(shape).clone()
        off += n
    return sd


def smear_table(offset: torch.Tensor, length: int) -> torch.Tensor:
    """(G, L) Gaussian position code.  gruunet2.py:54-68 with the call pattern
    of gruunet2.py:139-142: distances are fp32 linspace(0,1,L); coeff is
    computed from the fp32 offset difference through .item()."""
    coeff = -0.5 / (offset[1] - offset[0]).item() ** 2
    pos = torch.linspace(0, 1, length).to(offset.dtype)
    d = pos.view(-1, 1) - offset.view(1, -1)            # (L, G)
    return torch.exp(coeff * torch.pow(d, 2)).t()       # (G, L)


def _with_smear(x: torch.Tensor, offset: torch.Tensor) -> torch.Tensor:
    """cat((x, smear), dim=-2), smear broadcast over batch (gruunet2.py:143)."""
    s = smear_table(offset, x.size(-1)).to(x.dtype)
    return torch.cat((x, s.unsqueeze(0).expand(x.size(0), -1, -1)), dim=-2)


def cell_step(sd: dict, x_t: torch.Tensor, hx: torch.Tensor, intermediates: dict | None = None):
    """One GRUUNetCell.forward (gruunet2.py:228-244).

    x_t (B, F) ; hx (B, H, C) -> out (B, F), h' (B, H, C)
    """
    off_in = sd["cell.input_gate.gs.offset"]
    off_rs = sd["cell.reset_gate.gs.offset"]
    off_out = sd["cell.output_gate.gs.offset"]

    # encoder / input gates: 4x relu(conv1d k3 s2 p1)      gruunet2.py:136-144
    res = [x_t.unsqueeze(1)]
    for lvl in range(4):
        w = sd[f"cell.input_gate.downs.{lvl}.conv.weight"]
        b = sd[f"cell.input_gate.downs.{lvl}.conv.bias"]
        res.append(F.relu(F.conv1d(_with_smear(res[-1], off_in), w, b, stride=2, padding=1)))

    # hidden gates: relu(conv1d k3 s1 p1)                   gruunet2.py:145-155
    gate_h = F.relu(F.conv1d(_with_smear(hx, off_rs),
                             sd["cell.reset_gate.downs.0.conv.weight"],
                             sd["cell.reset_gate.downs.0.conv.bias"], stride=1, padding=1))

    # GRU pointwise, chunk order r, i, n                    gruunet2.py:234-240
    i_r, i_i, i_n = res[-1].chunk(3, 1)
    h_r, h_i, h_n = gate_h.chunk(3, 1)
    inputgate = torch.sigmoid(i_i + h_i)
    resetgate = torch.sigmoid(i_r + h_r)
    newgate = torch.tanh(i_n + resetgate * h_n)
    hi = newgate + inputgate * (hx - newgate)

    # decoder: 4x conv_transpose1d k3 s2 p1 output_padding=1 (L -> 2L)
    #                                                        gruunet2.py:184-199, 89-96
    skips = res[:-1]                       # [x, d0, d1, d2]
    h = hi
    for lvl in range(4):
        w = sd[f"cell.output_gate.ups.{lvl}.conv.weight"]
        b = sd[f"cell.output_gate.ups.{lvl}.conv.bias"]
        s = skips[3 - lvl]
        y = F.conv_transpose1d(_with_smear(h, off_out), w, b, stride=2, padding=1,
                               output_padding=s.size(-1) - (2 * h.size(-1) - 1))
        h = y if lvl == 3 else torch.cat((F.relu(y), s), dim=-2)
    if intermediates is not None:
        intermediates.update(d0=res[1], d1=res[2], d2=res[3], d3=res[4], gate_h=gate_h, hi=hi)
    return h.squeeze(-2), hi


def forward(sd: dict, x: torch.Tensor, hx: torch.Tensor | None = None, num_compressed_bins: int | None = None):
    """GRUUNet2.forward (gruunet2.py:290-306): x (B,T,F) or (T,F)."""
    two_d = x.dim() == 2
    if two_d:
        x = x.unsqueeze(0)
    if hx is None:
        c = num_compressed_bins if num_compressed_bins is not None else x.size(-1) // 16
        hx = torch.zeros(x.size(0), 17, c, dtype=x.dtype, device=x.device)
    outs = []
    for x_t in x.unbind(1):
        o, hx = cell_step(sd, x_t, hx)
        outs.append(o)
    out = torch.stack(outs, dim=1)
    if two_d:
        out = out.squeeze(0)
    return out, hx
