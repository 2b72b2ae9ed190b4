// dn_wavefft.hpp -- one-wavefront FFT machinery for gfx950 (wave64).
//
// A length-1024 real transform is done as a 512-point complex FFT plus a Hermitian
// split.  ONE 64-lane wavefront owns one transform: 512 = 8*8*8, so each of the three
// Stockham radix-8 passes is exactly one butterfly per lane (8 complex values = 16
// VGPRs per lane).  Between passes the values are exchanged through a wave-private
// LDS tile with ds_write_b64/ds_read_b64; no workgroup barrier is involved, only a
// wavefront-scope fence (LDS operations of one wave execute in issue order).
//
// Data convention everywhere: lane j holds element  j + 64*t  in v[t], t = 0..7,
// natural order on input AND on output.
#pragma once
#include <hip/hip_runtime.h>

namespace dn {

constexpr int kWave = 64;
constexpr int kNC = 512;            // complex FFT length
constexpr int kNR = 1024;           // real FFT length (n_fft)
constexpr int kBins = 513;          // n_fft/2+1
constexpr int kFftTile = 576;       // float2 entries of one wave's exchange tile (512 + padding)

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cmul_conj(float2 a, float2 b) {   // a * conj(b)
    return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -a.x * b.y));
}
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ float2 cscale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }

// multiply by -i (forward) or +i (inverse)
template <bool INV>
__device__ __forceinline__ float2 rot90(float2 a) {
    return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

// Wavefront-scope synchronisation point for wave-private LDS exchanges.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 8-point DFT in registers, natural order in and out.  INV selects e^{+...}.
template <bool INV>
__device__ __forceinline__ void dft8(float2 (&v)[8]) {
    constexpr float kH = 0.70710678118654752440f;
    float2 a0 = cadd(v[0], v[4]), a4 = csub(v[0], v[4]);
    float2 a1 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
    float2 a2 = cadd(v[2], v[6]), a6 = csub(v[2], v[6]);
    float2 a3 = cadd(v[3], v[7]), a7 = csub(v[3], v[7]);
    // odd branch twiddles W8^1, W8^2, W8^3 (conjugated for the inverse)
    float2 r5 = rot90<INV>(a5);                       // a5 * (-/+ i)
    a5 = make_float2((a5.x + r5.x) * kH, (a5.y + r5.y) * kH);     // a5 * (1 -/+ i)/sqrt2
    a6 = rot90<INV>(a6);
    float2 r7 = rot90<INV>(a7);
    a7 = make_float2((r7.x - a7.x) * kH, (r7.y - a7.y) * kH);     // a7 * (-1 -/+ i)/sqrt2
    // even outputs: DFT4(a0,a1,a2,a3)
    float2 c0 = cadd(a0, a2), c1 = csub(a0, a2), c2 = cadd(a1, a3), c3 = rot90<INV>(csub(a1, a3));
    v[0] = cadd(c0, c2); v[4] = csub(c0, c2); v[2] = cadd(c1, c3); v[6] = csub(c1, c3);
    // odd outputs: DFT4(a4,a5,a6,a7)
    float2 e0 = cadd(a4, a6), e1 = csub(a4, a6), e2 = cadd(a5, a7), e3 = rot90<INV>(csub(a5, a7));
    v[1] = cadd(e0, e2); v[5] = csub(e0, e2); v[3] = cadd(e1, e3); v[7] = csub(e1, e3);
}

// Per-lane twiddles of passes 1 and 2 (forward sign); the inverse uses conjugates.
struct FftTwiddles {
    float2 p1[7];   // exp(-2 pi i (j&7) t / 64),  t = 1..7
    float2 p2[7];   // exp(-2 pi i  j    t / 512), t = 1..7
};

// tw512[k] = exp(-2 pi i k / 512), k = 0..511 (device global table, built on the host in double).
__device__ __forceinline__ void load_twiddles(FftTwiddles& tw, const float2* __restrict__ tw512, int lane) {
#pragma unroll
    for (int t = 1; t < 8; ++t) {
        tw.p1[t - 1] = tw512[((lane & 7) * t * 8) & 511];
        tw.p2[t - 1] = tw512[(lane * t) & 511];
    }
}

// Padded tile index maps: chosen so that the scattered ds_write_b64 of each exchange and the
// strided ds_read_b64 that follows are (nearly) bank-conflict free ("LDS exchange").
__device__ __forceinline__ int pad0(int c) { return c + (c >> 4); }        // exchange after pass 0
__device__ __forceinline__ int pad1(int c) { return c + ((c >> 6) << 3); }  // exchange after pass 1

// 512-point complex FFT of one wavefront.  tile: this wave's kFftTile float2 LDS entries.
// Unnormalised in both directions.
template <bool INV>
__device__ __forceinline__ void fft512(float2 (&v)[8], const FftTwiddles& tw, float2* tile, int lane) {
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

// ---- Hermitian split / merge for the real transform of length 1024 -----------------
// wk[t] = exp(-2 pi i k / 1024) for k = lane + 64 t.
//
// Forward: z[m] = x[2m] + i x[2m+1], Z = FFT512(z).  For k = 0..511
//     X[k] = 1/2 [ (Z[k] + conj Z[512-k]) - i wk (Z[k] - conj Z[512-k]) ],  Z[512] := Z[0]
// and X[512] = Re Z[0] - Im Z[0].  `zp` is Z[(512-k) & 511] fetched through LDS.
__device__ __forceinline__ float2 rfft_post(float2 z, float2 zp, float2 wk) {
    float2 e = make_float2(0.5f * (z.x + zp.x), 0.5f * (z.y - zp.y));   // (Z + conj Zp)/2
    float2 d = make_float2(0.5f * (z.x - zp.x), 0.5f * (z.y + zp.y));   // (Z - conj Zp)/2
    float2 wd = cmul(wk, d);
    return make_float2(e.x + wd.y, e.y - wd.x);                         // e - i*wd
}

// Inverse: given the one-sided spectrum X[0..512] (Im of DC and Nyquist ignored, as C2R
// transforms do), build Z[k] = E + i D with E = (X[k] + conj X[512-k])/2,
// D = (X[k] - conj X[512-k])/2 * conj(wk).  IFFT512(Z) then yields x[2m] + i x[2m+1]
// (times 512; the 1/512 is folded into the synthesis window).
__device__ __forceinline__ float2 irfft_pre(float2 x, float2 xp, float2 wk) {
    float2 e = make_float2(0.5f * (x.x + xp.x), 0.5f * (x.y - xp.y));
    float2 d = make_float2(0.5f * (x.x - xp.x), 0.5f * (x.y + xp.y));
    float2 dd = cmul_conj(d, wk);
    return make_float2(e.x - dd.y, e.y + dd.x);                         // e + i*dd
}

// Forward real FFT of one wave: v[t] = (x[2m], x[2m+1]) for m = lane + 64 t on entry,
// X[k] for k = lane + 64 t on exit; returns X[512] (valid in every lane).
// hbuf: this wave's 513-entry float2 LDS line (may alias the fft tile).
__device__ __forceinline__ float rfft1024(float2 (&v)[8], const FftTwiddles& tw, const float2 (&wk)[8],
                                          float2* tile, float2* hbuf, int lane) {
    fft512<false>(v, tw, tile, lane);
    wave_sync();
#pragma unroll
    for (int t = 0; t < 8; ++t) hbuf[lane + 64 * t] = v[t];
    wave_sync();
    float2 z0 = hbuf[0];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        float2 zp = hbuf[(512 - (lane + 64 * t)) & 511];
        v[t] = rfft_post(v[t], zp, wk[t]);
    }
    return z0.x - z0.y;
}

// Inverse real FFT of one wave: v[t] = X[k], k = lane + 64 t, xnyq = Re X[512] on entry;
// on exit v[t] = 512 * (x[2m], x[2m+1]), m = lane + 64 t.
__device__ __forceinline__ void irfft1024(float2 (&v)[8], float xnyq, const FftTwiddles& tw,
                                          const float2 (&wk)[8], float2* tile, float2* hbuf, int lane) {
    if (lane == 0) v[0].y = 0.0f;                    // Im X[0] ignored
    wave_sync();
#pragma unroll
    for (int t = 0; t < 8; ++t) hbuf[lane + 64 * t] = v[t];
    if (lane == 0) hbuf[512] = make_float2(xnyq, 0.0f);
    wave_sync();
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        float2 xp = hbuf[512 - (lane + 64 * t)];
        v[t] = irfft_pre(v[t], xp, wk[t]);
    }
    fft512<true>(v, tw, tile, lane);
}

// ---- Pair-owned Hermitian split/merge (Griffin-Lim inner loop) ---------------------------------
// Bins k and 512-k are produced together by the split, so the lane that owns k = lane + 64 t (t < 4)
// also owns 512-k: the split, the per-bin phase update and the merge for the next inverse FFT are
// then lane-local and only the FFT-order <-> pair-order hand-off crosses lanes.  That hand-off is
// the fixed involution (lane j, reg t) <-> (lane 64-j, reg 7-t), done with ds_bpermute (no LDS
// memory, no barrier).  Lane 0 pairs with itself: (0,512), (64,448), (128,384), (192,320) and the
// self-paired bin 256.
__device__ __forceinline__ float2 shfl2(float2 v, int src) { return make_float2(__shfl(v.x, src), __shfl(v.y, src)); }

// Forward: v[t] = Z[lane + 64 t] (FFT512 of the packed real frame) ->
//   lo[t] = X[k], hi[t] = X[512-k] for k = lane + 64 t, t = 0..3; mid = X[256] (meaningful in lane 0).
__device__ __forceinline__ void rfft_split_pairs(const float2 (&v)[8], const float2 (&wk)[4], int lane,
                                                 float2 (&lo)[4], float2 (&hi)[4], float2& mid) {
    const int partner = (64 - lane) & 63;
    float2 zp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) zp[t] = shfl2(v[7 - t], partner);
    if (lane == 0) { zp[0] = v[0]; zp[1] = v[7]; zp[2] = v[6]; zp[3] = v[5]; }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float2 e = make_float2(0.5f * (v[t].x + zp[t].x), 0.5f * (v[t].y - zp[t].y));   // (Z + conj Zp)/2
        const float2 d = make_float2(0.5f * (v[t].x - zp[t].x), 0.5f * (v[t].y + zp[t].y));   // (Z - conj Zp)/2
        const float2 wd = cmul(wk[t], d);
        lo[t] = make_float2(e.x + wd.y, e.y - wd.x);          // E - i wd
        hi[t] = make_float2(e.x - wd.y, -(e.y + wd.x));       // conj(E + i wd)
    }
    mid = make_float2(v[4].x, -v[4].y);                        // X[256] = conj Z[256]
}

// Inverse: lo[t] = X[k], hi[t] = X[512-k], mid = X[256] -> v[t] = Z[lane + 64 t] ready for IFFT512
// (Im X[0] and Im X[512] are ignored, as C2R transforms do).
__device__ __forceinline__ void irfft_merge_pairs(float2 (&lo)[4], float2 (&hi)[4], float2 mid, const float2 (&wk)[4],
                                                  int lane, float2 (&v)[8]) {
    if (lane == 0) { lo[0].y = 0.0f; hi[0].y = 0.0f; }
    float2 zh[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float2 e = make_float2(0.5f * (lo[t].x + hi[t].x), 0.5f * (lo[t].y - hi[t].y));  // (X + conj Xp)/2
        const float2 d = make_float2(0.5f * (lo[t].x - hi[t].x), 0.5f * (lo[t].y + hi[t].y));  // (X - conj Xp)/2
        const float2 dd = cmul_conj(d, wk[t]);
        v[t] = make_float2(e.x - dd.y, e.y + dd.x);           // E + i D
        zh[t] = make_float2(e.x + dd.y, dd.x - e.y);          // conj(E) + i conj(D)
    }
    const int partner = (64 - lane) & 63;
#pragma unroll
    for (int t = 0; t < 4; ++t) v[7 - t] = shfl2(zh[t], partner);
    if (lane == 0) { v[7] = zh[1]; v[6] = zh[2]; v[5] = zh[3]; v[4] = make_float2(mid.x, -mid.y); }
}

}  // namespace dn
