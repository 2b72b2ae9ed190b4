// dn_wavefft.hpp -- one-wavefront FFT machinery for gfx950 (wave64).
//
// A length-1024 real transform is done as a 512-point complex FFT plus a Hermitian
// split.  ONE 64-lane wavefront owns one transform: 512 = 8*8*8, so each of the three
// Stockham radix-8 passes is exactly one butterfly per lane (8 complex values = 16
// VGPRs per lane).  Between passes the values are exchanged through a wave-private
// LDS tile with ds_write_b64/ds_read_b64; no workgroup barrier is involved, only a
// wavefront-scope fence (LDS operations of one wave execute in issue order).
// Complex arithmetic is packed fp32 with operand modifiers (dn_cpx.hpp): a radix-8
// butterfly is 24 v_pk_add + 2 v_pk_mul, a twiddle multiply 2 instructions.
//
// Data convention everywhere: lane j holds element  j + 64*t  in v[t], t = 0..7,
// natural order on input AND on output.
#pragma once
#include <hip/hip_runtime.h>

#include <dn_cpx.hpp>

namespace dn {

constexpr int kWave = 64;
constexpr int kNC = 512;            // complex FFT length
constexpr int kNR = 1024;           // real FFT length (n_fft)
constexpr int kBins = 513;          // n_fft/2+1
constexpr int kFftTile = 576;       // complex entries of one wave's exchange tile (512 + padding)

// Wavefront-scope synchronisation point for wave-private LDS exchanges.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 8-point DFT in registers, natural order in and out.  INV selects e^{+...}.
// "rot" below is multiplication by -i (forward) / +i (inverse); it rides on the add.
template <bool INV>
__device__ __forceinline__ void dft8(v2f (&v)[8]) {
    constexpr float kH = 0.70710678118654752440f;
    const v2f a0 = cadd(v[0], v[4]), a4 = csub(v[0], v[4]);
    const v2f a1 = cadd(v[1], v[5]), b5 = csub(v[1], v[5]);
    const v2f a2 = cadd(v[2], v[6]), b6 = csub(v[2], v[6]);
    const v2f a3 = cadd(v[3], v[7]), b7 = csub(v[3], v[7]);
    // odd-branch twiddles: a5 = b5 W8, a6 = b6 W8^2 = rot(b6), a7 = b7 W8^3 with W8 = (1 + rot)/sqrt2, W8^3 = (rot - 1)/sqrt2
    const v2f a5 = cscale(cadd_rot<INV>(b5, b5), kH);
    const v2f a7 = cscale(csub_rot<INV>(b7, b7), -kH);      // -(b7 - rot b7) = rot b7 - b7
    // even outputs: DFT4(a0, a1, a2, a3)
    const v2f c0 = cadd(a0, a2), c1 = csub(a0, a2), c2 = cadd(a1, a3), c3 = csub(a1, a3);
    v[0] = cadd(c0, c2); v[4] = csub(c0, c2); v[2] = cadd_rot<INV>(c1, c3); v[6] = csub_rot<INV>(c1, c3);
    // odd outputs: DFT4(a4, a5, rot b6, a7)
    const v2f e0 = cadd_rot<INV>(a4, b6), e1 = csub_rot<INV>(a4, b6), e2 = cadd(a5, a7), e3 = csub(a5, a7);
    v[1] = cadd(e0, e2); v[5] = csub(e0, e2); v[3] = cadd_rot<INV>(e1, e3); v[7] = csub_rot<INV>(e1, e3);
}

// Per-lane twiddles of passes 1 and 2 (forward sign); the inverse uses conjugates.
struct FftTwiddles {
    v2f p1[7];   // exp(-2 pi i (j&7) t / 64),  t = 1..7
    v2f p2[7];   // exp(-2 pi i  j    t / 512), t = 1..7
};

// tw512[k] = exp(-2 pi i k / 512), k = 0..511 (device global table, built on the host in double).
__device__ __forceinline__ void load_twiddles(FftTwiddles& tw, const float2* __restrict__ tw512, int lane) {
    const v2f* t512 = reinterpret_cast<const v2f*>(tw512);
#pragma unroll
    for (int t = 1; t < 8; ++t) {
        tw.p1[t - 1] = t512[((lane & 7) * t * 8) & 511];
        tw.p2[t - 1] = t512[(lane * t) & 511];
    }
}

// Padded tile index maps: chosen so that the scattered ds_write_b64 of each exchange and the
// strided ds_read_b64 that follows are (nearly) bank-conflict free ("LDS exchange").
__device__ __forceinline__ int pad0(int c) { return c + (c >> 4); }        // exchange after pass 0
__device__ __forceinline__ int pad1(int c) { return c + ((c >> 6) << 3); }  // exchange after pass 1

// 512-point complex FFT of one wavefront.  tile: this wave's kFftTile complex LDS entries.
// Unnormalised in both directions.
template <bool INV>
__device__ __forceinline__ void fft512(v2f (&v)[8], const FftTwiddles& tw, v2f* tile, int lane) {
    // pass 0 (Ns = 1): no twiddles
    dft8<INV>(v);
    wave_sync();                       // previous readers of the tile are done
#pragma unroll
    for (int t = 0; t < 8; ++t) tile[pad0(8 * lane + t)] = v[t];
    wave_sync();
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = tile[pad0(lane + 64 * t)];
    // pass 1 (Ns = 8)
#pragma unroll
    for (int t = 1; t < 8; ++t) v[t] = INV ? cmul_conj(v[t], tw.p1[t - 1]) : cmul(v[t], tw.p1[t - 1]);
    dft8<INV>(v);
    wave_sync();
    {
        const int base = ((lane >> 3) << 6) + (lane & 7);
#pragma unroll
        for (int t = 0; t < 8; ++t) tile[pad1(base + 8 * t)] = v[t];
    }
    wave_sync();
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = tile[pad1(lane + 64 * t)];
    // pass 2 (Ns = 64): output index j + 64 t stays in registers
#pragma unroll
    for (int t = 1; t < 8; ++t) v[t] = INV ? cmul_conj(v[t], tw.p2[t - 1]) : cmul(v[t], tw.p2[t - 1]);
    dft8<INV>(v);
}

// ---- Hermitian split / merge for the real transform of length 1024 -----------------------------
// z[m] = x[2m] + i x[2m+1], Z = FFT512(z), W = exp(-2 pi i / 1024).  For k = 0..512 (Z[512] := Z[0])
//     X[k]     = 1/2 (Z[k] + conj Z[512-k])  -  i W^k 1/2 (Z[k] - conj Z[512-k])
//     X[512-k] = conj( 1/2 (Z[k] + conj Z[512-k])  +  i W^k 1/2 (Z[k] - conj Z[512-k]) )
// and in the other direction (Im X[0], Im X[512] ignored, as C2R transforms do)
//     Z[k]     = 1/2 (X[k] + conj X[512-k])  +  i conj(W^k) 1/2 (X[k] - conj X[512-k])
//     Z[512-k] = conj(1/2 (X[k] + conj X[512-k]))  +  i conj( conj(W^k) 1/2 (X[k] - conj X[512-k]) )
// IFFT512(Z) then yields x[2m] + i x[2m+1] (times 512; the 1/512 is folded into the synthesis window).
//
// Bins k and 512-k are produced together, so the lane that owns k = lane + 64 t (t < 4) also owns
// 512-k ("pair order"): the split, any per-bin work and the merge are then lane-local and only the
// FFT-order <-> pair-order hand-off crosses lanes.  That hand-off is the fixed involution
// (lane j, reg t) <-> (lane 64-j, reg 7-t), done with ds_bpermute (no LDS memory, no barrier).
// Lane 0 pairs with itself: (0,512), (64,448), (128,384), (192,320) and the self-paired bin 256.
// `wkh[t]` = 1/2 W^k for k = lane + 64 t, t = 0..3.
__device__ __forceinline__ v2f shfl2(v2f v, int src) { return mk2(__shfl(v[0], src), __shfl(v[1], src)); }

// Forward: v[t] = Z[lane + 64 t] -> lo[t] = X[k], hi[t] = X[512-k]; mid = X[256] (meaningful in lane 0).
__device__ __forceinline__ void rfft_split_pairs(const v2f (&v)[8], const v2f (&wkh)[4], int lane,
                                                 v2f (&lo)[4], v2f (&hi)[4], v2f& mid) {
    const int partner = (64 - lane) & 63;
    v2f zp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) zp[t] = shfl2(v[7 - t], partner);
    if (lane == 0) { zp[0] = v[0]; zp[1] = v[7]; zp[2] = v[6]; zp[3] = v[5]; }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const v2f s = cadd_conj(v[t], zp[t]);                  // Z + conj Zp
        const v2f wd = cmul(wkh[t], csub_conj(v[t], zp[t]));   // 1/2 W^k (Z - conj Zp)
        lo[t] = chalf_add_mi(s, wd);                           // s/2 - i wd
        hi[t] = cconj_half_add_pi(s, wd);                      // conj(s/2 + i wd)
    }
    mid = mk2(v[4][0], -v[4][1]);                              // X[256] = conj Z[256]
}

// Inverse: lo[t] = X[k], hi[t] = X[512-k], mid = X[256] -> v[t] = Z[lane + 64 t] ready for IFFT512.
__device__ __forceinline__ void irfft_merge_pairs(v2f (&lo)[4], v2f (&hi)[4], v2f mid, const v2f (&wkh)[4],
                                                  int lane, v2f (&v)[8]) {
    if (lane == 0) { lo[0][1] = 0.0f; hi[0][1] = 0.0f; }
    v2f zh[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const v2f s = cadd_conj(lo[t], hi[t]);                       // X + conj Xp
        const v2f dd = cmul_conj(csub_conj(lo[t], hi[t]), wkh[t]);   // 1/2 conj(W^k) (X - conj Xp)
        v[t] = chalf_add_pi(s, dd);                                  // s/2 + i dd
        zh[t] = chalf_conj_add_iconj(s, dd);                         // conj(s/2) + i conj(dd)
    }
    const int partner = (64 - lane) & 63;
#pragma unroll
    for (int t = 0; t < 4; ++t) v[7 - t] = shfl2(zh[t], partner);
    if (lane == 0) { v[7] = zh[1]; v[6] = zh[2]; v[5] = zh[3]; v[4] = mk2(mid[0], -mid[1]); }
}

}  // namespace dn
