"""CPU restatement of the reference's sibling model MOMO3 (TEST INFRASTRUCTURE).

Follows /root/reference/momo3.py:
  * GaussianSmearing            momo3.py:54-68      (same as gruunet2.py)
  * DownBlocks                  momo3.py:103-157    the position code is concatenated ONCE, at the block's input (140-145);
                                                    the following levels convolve data channels only
  * UpBlocks                    momo3.py:159-189    no position code at all; output_size = length of the skip (185-186)
  * MOMOCell.forward            momo3.py:191-245    conv-GRU gates as in GRUUNetCell
  * MOMO3._momo / forward       momo3.py:266-324    second input channel = frame delta x_t - prev (285-289); prev starts as x_t

Checkpoint saves/MOMO3-4d4ea0: 3 levels, hidden 16, kernel 3, stride 2, paddings (1, 0, 1), 6 gaussians, in_size 1
(22 bins -> 11 -> 5 -> 3 = num_compressed_bins).
PINNED: tests/test_oracle_model.py checks it against tests/golden/momo3_*.npz, produced by the reference's own class
(oracle/make_momo_golden.py).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .model_ref import smear_table

H = 16
PADDINGS = (1, 0, 1)
STATE_KEYS = (
    ("cell.input_gate.downs.0.conv.weight", (16, 8, 3)),
    ("cell.input_gate.downs.0.conv.bias", (16,)),
    ("cell.input_gate.downs.1.conv.weight", (16, 16, 3)),
    ("cell.input_gate.downs.1.conv.bias", (16,)),
    ("cell.input_gate.downs.2.conv.weight", (48, 16, 3)),
    ("cell.input_gate.downs.2.conv.bias", (48,)),
    ("cell.input_gate.gs.offset", (6,)),
    ("cell.reset_gate.downs.0.conv.weight", (48, 22, 3)),
    ("cell.reset_gate.downs.0.conv.bias", (48,)),
    ("cell.reset_gate.gs.offset", (6,)),
    ("cell.output_gate.ups.0.conv.weight", (16, 16, 3)),
    ("cell.output_gate.ups.0.conv.bias", (16,)),
    ("cell.output_gate.ups.1.conv.weight", (32, 16, 3)),
    ("cell.output_gate.ups.1.conv.bias", (16,)),
    ("cell.output_gate.ups.2.conv.weight", (32, 1, 3)),
    ("cell.output_gate.ups.2.conv.bias", (1,)),
)
N_WEIGHT_FLOATS = sum(int(torch.tensor(s).prod()) for _, s in STATE_KEYS)      # 9,165


def unflatten_weights(blob) -> dict:
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


def _informed(x: torch.Tensor, offset: torch.Tensor) -> torch.Tensor:
    s = smear_table(offset, x.size(-1)).to(x.dtype)
    return torch.cat((x, s.unsqueeze(0).expand(x.size(0), -1, -1)), dim=-2)            # momo3.py:140-145


def cell_step(sd: dict, x2: torch.Tensor, hx: torch.Tensor):
    """One MOMOCell.forward (momo3.py:228-245): x2 (B, 2, F) = [x_t ; x_t - prev], hx (B, 16, C) -> out (B, F), h' (B, 16, C)."""
    res = [_informed(x2, sd["cell.input_gate.gs.offset"])]
    for lvl in range(3):
        res.append(F.relu(F.conv1d(res[-1], sd[f"cell.input_gate.downs.{lvl}.conv.weight"], sd[f"cell.input_gate.downs.{lvl}.conv.bias"],
                                   stride=2, padding=PADDINGS[lvl])))
    gate_h = F.relu(F.conv1d(_informed(hx, sd["cell.reset_gate.gs.offset"]), sd["cell.reset_gate.downs.0.conv.weight"],
                             sd["cell.reset_gate.downs.0.conv.bias"], stride=1, padding=1))
    i_r, i_i, i_n = res[-1].chunk(3, 1)
    h_r, h_i, h_n = gate_h.chunk(3, 1)
    inputgate = torch.sigmoid(i_i + h_i)
    resetgate = torch.sigmoid(i_r + h_r)
    newgate = torch.tanh(i_n + resetgate * h_n)
    hi = newgate + inputgate * (hx - newgate)
    skips = res[:-1]                                   # [informed_x, d0, d1]
    h = hi
    for lvl in range(3):
        s = skips[2 - lvl]
        pad = PADDINGS[::-1][lvl]
        base = (h.size(-1) - 1) * 2 - 2 * pad + 3
        y = F.conv_transpose1d(h, sd[f"cell.output_gate.ups.{lvl}.conv.weight"], sd[f"cell.output_gate.ups.{lvl}.conv.bias"],
                               stride=2, padding=pad, output_padding=s.size(-1) - base)          # output_size = len(skip), momo3.py:185-187
        h = y if lvl == 2 else torch.cat((F.relu(y), s), dim=-2)
    return h.squeeze(-2), hi


def compressed_bins(F_bins: int) -> int:
    L = F_bins
    for p in PADDINGS:
        L = (L + 2 * p - 3) // 2 + 1
    return L


def forward(sd: dict, x: torch.Tensor, hx: torch.Tensor | None = None, prev: torch.Tensor | None = None, num_compressed_bins: int = 3):
    """MOMO3.forward (momo3.py:300-324): x (B,T,F) or (T,F); prev (B,1,F) or None -> (out, hx).  Also returns the last frame
    (what a caller must pass as `prev` to continue the sequence in another call)."""
    two_d = x.dim() == 2
    if two_d:
        x = x.unsqueeze(0)
    if hx is None:
        hx = torch.zeros(x.size(0), H, num_compressed_bins, dtype=x.dtype)
    outs = []
    for x_t in x.unbind(1):
        x_t = x_t.unsqueeze(1)
        if prev is None:
            prev = x_t.clone()
        o, hx = cell_step(sd, torch.cat([x_t, x_t - prev], -2), hx)
        prev = x_t.clone()
        outs.append(o)
    out = torch.stack(outs, dim=1)
    if two_d:
        out = out.squeeze(0)
    return out, hx
