"""TEST INFRASTRUCTURE: Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) restated from
the paper in numpy, independent of the device code, and pinned by the three known-answer vectors Random123 publishes for philox4x32_10
(kat_vectors).  The hop's Griffin-Lim initial phases (torchaudio GriffinLim(rand_init=True) = torch.rand(complex64), app3.py:149-153) are drawn
on the device as: counter = (bin, column, stream id lo, stream id hi), key = (seed lo, seed hi); real = word0 >> 8, imag = word1 >> 8, * 2^-24.
Only tests/ import this."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)

# (counter, key, output) of philox4x32_10 from Random123's kat_vectors
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Arrays (or scalars) of 32-bit words -> four arrays of uint64 holding the 32-bit output words."""
    c = [np.asarray(x, dtype=np.uint64) & MASK for x in (c0, c1, c2, c3)]
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0)) & MASK, p1 & MASK, ((p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1)) & MASK, p0 & MASK]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c


def draw_phases(seed, stream_id0, batch, n_stft):
    """complex64 (batch, n_stft, 3): what dn_griffinlim_draw_phases must return.  One block per bin PAIR (m, NC - m), NC = n_stft - 1,
    m = 0..NC/2, counter (m, column, stream id lo, hi): words 0, 1 -> bin m, words 2, 3 -> bin NC - m (csrc/dn_gl_body.hpp rand_angle_pair)."""
    out = np.empty((batch, n_stft, 3), np.complex64)
    nc = n_stft - 1
    m = np.arange(nc // 2 + 1, dtype=np.uint64)
    n = len(m)
    unit = lambda w: (w >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)  # noqa: E731
    for b in range(batch):
        sid = stream_id0 + b
        for col in range(3):
            w = philox4x32_10(m, np.full(n, col, np.uint64), np.full(n, sid & 0xFFFFFFFF, np.uint64),
                              np.full(n, (sid >> 32) & 0xFFFFFFFF, np.uint64), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
            out[b, nc - m.astype(np.int64), col] = unit(w[2]) + 1j * unit(w[3])
            out[b, m.astype(np.int64), col] = unit(w[0]) + 1j * unit(w[1])          # (bin NC/2 pairs with itself: words 0, 1)
    return out
