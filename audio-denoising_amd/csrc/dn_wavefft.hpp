// dn_wavefft.hpp -- one-wavefront FFT machinery for gfx950 (wave64).
//
// A real transform of length n_fft is done as a complex FFT of length NC = n_fft/2 plus a
// Hermitian split.  ONE 64-lane wavefront owns one transform and holds NV = NC/64 complex values
// per lane; every Stockham pass is a whole number of radix-R butterflies per lane:
//     n_fft 1024: NC = 512 = 8 * 8 * 8      (NV =  8; three radix-8 passes, one butterfly per lane)
//     n_fft 1536: NC = 768 = 4 * 4 * 4 * 12 (NV = 12; three radix-4 passes of three butterflies per
//                                             lane, then one radix-12 pass)           [app3.py:29-33]
// Between passes the values are exchanged through a wave-private LDS tile with
// ds_write_b64/ds_read_b64; no workgroup barrier is involved, only a wavefront-scope fence (LDS
// operations of one wave execute in issue order).  Complex arithmetic is packed fp32 with operand
// modifiers (dn_cpx.hpp): a radix-8 butterfly is 26 v_pk_add + 2 v_pk_mul, a twiddle multiply 2.
//
// Data convention everywhere: lane j holds element  j + 64*t  in v[t], t = 0..NV-1, natural order
// on input AND on output.  A pass of radix R runs G = NV/R butterflies per lane; butterfly g works
// on registers v[g + G*r], r = 0..R-1 (elements j + 64 g + r*NC/R).
#pragma once
#include <hip/hip_runtime.h>

#include <dn_cpx.hpp>

namespace dn {

// Wavefront-scope synchronisation point for wave-private LDS exchanges.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- small DFTs in registers, natural order in and out.  INV selects e^{+...}.
// "rot" is multiplication by -i (forward) / +i (inverse); it rides on the add (dn_cpx.hpp).
template <bool INV>
__device__ __forceinline__ void dft4(v2f& x0, v2f& x1, v2f& x2, v2f& x3) {
    const v2f c0 = cadd(x0, x2), c1 = csub(x0, x2), c2 = cadd(x1, x3), c3 = csub(x1, x3);
    x0 = cadd(c0, c2); x2 = csub(c0, c2); x1 = cadd_rot<INV>(c1, c3); x3 = csub_rot<INV>(c1, c3);
}

template <bool INV>
__device__ __forceinline__ void dft3(v2f& x0, v2f& x1, v2f& x2) {
    constexpr float kS = 0.86602540378443864676f;            // sin(pi/3)
    const v2f t = cadd(x1, x2), s = cscale(csub(x1, x2), kS);
    const v2f m = x0 - t * 0.5f;
    x0 = cadd(x0, t); x1 = cadd_rot<INV>(m, s); x2 = csub_rot<INV>(m, s);
}

template <bool INV>
__device__ __forceinline__ void dft8(v2f (&v)[8]) {
    constexpr float kH = 0.70710678118654752440f;
    const v2f a0 = cadd(v[0], v[4]), a4 = csub(v[0], v[4]);
    const v2f a1 = cadd(v[1], v[5]), b5 = csub(v[1], v[5]);
    const v2f a2 = cadd(v[2], v[6]), b6 = csub(v[2], v[6]);
    const v2f a3 = cadd(v[3], v[7]), b7 = csub(v[3], v[7]);
    // odd-branch twiddles: a5 = b5 W8, a6 = rot(b6), a7 = b7 W8^3 with W8 = (1 + rot)/sqrt2, W8^3 = (rot - 1)/sqrt2
    const v2f a5 = cscale(cadd_rot<INV>(b5, b5), kH);
    const v2f a7 = cscale(csub_rot<INV>(b7, b7), -kH);
    const v2f c0 = cadd(a0, a2), c1 = csub(a0, a2), c2 = cadd(a1, a3), c3 = csub(a1, a3);
    v[0] = cadd(c0, c2); v[4] = csub(c0, c2); v[2] = cadd_rot<INV>(c1, c3); v[6] = csub_rot<INV>(c1, c3);
    const v2f e0 = cadd_rot<INV>(a4, b6), e1 = csub_rot<INV>(a4, b6), e2 = cadd(a5, a7), e3 = csub(a5, a7);
    v[1] = cadd(e0, e2); v[5] = csub(e0, e2); v[3] = cadd_rot<INV>(e1, e3); v[7] = csub_rot<INV>(e1, e3);
}

// 12-point DFT = 3 x DFT4 over n2 (n = 3 n2 + n1), twiddles W12^(n1 k2), 4 x DFT3 over n1; X[k2 + 4 k1].
template <bool INV>
__device__ __forceinline__ void dft12(v2f (&v)[12]) {
    constexpr float kC = 0.86602540378443864676f;            // cos(pi/6)
    // rows n1 = 0,1,2: elements v[n1], v[n1+3], v[n1+6], v[n1+9]
    dft4<INV>(v[0], v[3], v[6], v[9]);
    dft4<INV>(v[1], v[4], v[7], v[10]);
    dft4<INV>(v[2], v[5], v[8], v[11]);
    // after dft4 the k2-th output of row n1 sits in v[n1 + 3 k2].  Twiddle W12^(n1 k2), W12 = exp(-/+ i pi/6):
    const v2f w1 = mk2(kC, INV ? 0.5f : -0.5f), w2 = mk2(0.5f, INV ? kC : -kC), w4 = mk2(-0.5f, INV ? kC : -kC);
    v[4] = cmul(v[4], w1);                                   // n1=1,k2=1: W^1
    v[7] = cmul(v[7], w2);                                   // n1=1,k2=2: W^2
    v[10] = cadd_rot<INV>(mk2(0.f, 0.f), v[10]);             // n1=1,k2=3: W^3 = rot
    v[5] = cmul(v[5], w2);                                   // n1=2,k2=1: W^2
    v[8] = cmul(v[8], w4);                                   // n1=2,k2=2: W^4
    v[11] = mk2(-v[11][0], -v[11][1]);                       // n1=2,k2=3: W^6 = -1
    // columns k2: DFT3 over n1 of v[0+3k2], v[1+3k2], v[2+3k2] -> X[k2], X[k2+4], X[k2+8]
    dft3<INV>(v[0], v[1], v[2]);
    dft3<INV>(v[3], v[4], v[5]);
    dft3<INV>(v[6], v[7], v[8]);
    dft3<INV>(v[9], v[10], v[11]);
    // gather to natural order: X[k2 + 4 k1] is in v[3 k2 + k1]
    const v2f x0 = v[0], x4 = v[1], x8 = v[2], x1 = v[3], x5 = v[4], x9 = v[5];
    const v2f x2 = v[6], x6 = v[7], x10 = v[8], x3 = v[9], x7 = v[10], x11 = v[11];
    v[0] = x0; v[1] = x1; v[2] = x2; v[3] = x3; v[4] = x4; v[5] = x5;
    v[6] = x6; v[7] = x7; v[8] = x8; v[9] = x9; v[10] = x10; v[11] = x11;
}

template <bool INV> __device__ __forceinline__ v2f twmul(v2f a, v2f w) { return INV ? cmul_conj(a, w) : cmul(a, w); }

// ---- per complex length: geometry, per-lane twiddles, the FFT itself -------------------------------
template <int NC> struct WaveFft;

template <> struct WaveFft<512> {
    static constexpr int kNV = 8;
    static constexpr int kTile = 576;        // complex entries of the exchange tile (512 + padding)
    struct Tw { v2f p1[7], p2[7]; };         // exp(-2 pi i (j&7) t/64), exp(-2 pi i j t/512), t = 1..7
    // twc[k] = exp(-2 pi i k / NC), k = 0..NC-1 (device global table, built on the host in double)
    static constexpr int kPdTable = 0;
    static __device__ __forceinline__ void fill_pd(v2f*, const v2f* __restrict__, int) {}
    template <bool WITH_PD = true>
    static __device__ __forceinline__ void load(Tw& tw, const v2f* __restrict__ twc, int lane) {
#pragma unroll
        for (int t = 1; t < 8; ++t) {
            tw.p1[t - 1] = twc[((lane & 7) * t * 8) & 511];
            tw.p2[t - 1] = twc[(lane * t) & 511];
        }
    }
    // Padded tile index maps keep the scattered ds_write_b64 and the strided ds_read_b64 (nearly) conflict free.
    static __device__ __forceinline__ int pad0(int c) { return c + (c >> 4); }
    static __device__ __forceinline__ int pad1(int c) { return c + ((c >> 6) << 3); }

    template <bool INV>
    static __device__ __forceinline__ void run(v2f (&v)[8], const Tw& tw, v2f* tile, int lane, const v2f* = nullptr) {
        dft8<INV>(v);                                         // pass 0 (Ns = 1)
        wave_sync();                                          // previous readers of the tile are done
#pragma unroll
        for (int t = 0; t < 8; ++t) tile[pad0(8 * lane + t)] = v[t];
        wave_sync();
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = tile[pad0(lane + 64 * t)];
#pragma unroll
        for (int t = 1; t < 8; ++t) v[t] = twmul<INV>(v[t], tw.p1[t - 1]);    // pass 1 (Ns = 8)
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
#pragma unroll
        for (int t = 1; t < 8; ++t) v[t] = twmul<INV>(v[t], tw.p2[t - 1]);    // pass 2 (Ns = 64): output stays in registers
        dft8<INV>(v);
    }

    // NB independent transforms of one wavefront, SKEWED by half a pass (one tile each): while transform c waits for an exchange -- its LDS writes
    // and the strided reads behind them, ~200 ticks -- transform c+1 runs the butterflies of its previous pass, so that only the first exchange of
    // the group is exposed (dn_glw_body.hpp: one wavefront per stream).  LDS operations of one wave execute in order and the tiles are disjoint, so the
    // wave_sync()s are compiler barriers only.  Same arithmetic as run().
    template <bool INV, int NB>
    static __device__ __forceinline__ void run_n(v2f (&v)[NB][8], const Tw& tw, v2f* const (&tile)[NB], int lane) {
        const int base = ((lane >> 3) << 6) + (lane & 7);
        auto put0 = [&](int c) {
            wave_sync();
#pragma unroll
            for (int t = 0; t < 8; ++t) tile[c][pad0(8 * lane + t)] = v[c][t];
            wave_sync();
        };
        auto get0 = [&](int c) {
#pragma unroll
            for (int t = 0; t < 8; ++t) v[c][t] = tile[c][pad0(lane + 64 * t)];
        };
        auto put1 = [&](int c) {
            wave_sync();
#pragma unroll
            for (int t = 0; t < 8; ++t) tile[c][pad1(base + 8 * t)] = v[c][t];
            wave_sync();
        };
        auto get1 = [&](int c) {
#pragma unroll
            for (int t = 0; t < 8; ++t) v[c][t] = tile[c][pad1(lane + 64 * t)];
        };
        auto pass1 = [&](int c) {
#pragma unroll
            for (int t = 1; t < 8; ++t) v[c][t] = twmul<INV>(v[c][t], tw.p1[t - 1]);
            dft8<INV>(v[c]);
        };
        auto pass2 = [&](int c) {
#pragma unroll
            for (int t = 1; t < 8; ++t) v[c][t] = twmul<INV>(v[c][t], tw.p2[t - 1]);
            dft8<INV>(v[c]);
        };
        // pass 0 of transform 0, its first exchange on the way; then every transform's butterflies sit between another one's write and use
        dft8<INV>(v[0]);
        put0(0); get0(0);
#pragma unroll
        for (int c = 1; c < NB; ++c) {
            dft8<INV>(v[c]);                 // (covers the exchange of transform c - 1)
            put0(c); get0(c);
            pass1(c - 1);
            put1(c - 1); get1(c - 1);
        }
        pass1(NB - 1);
        put1(NB - 1); get1(NB - 1);
#pragma unroll
        for (int c = 0; c < NB; ++c) pass2(c);
    }
};

template <> struct WaveFft<768> {
    static constexpr int kNV = 12;
    static constexpr int kTile = 960;        // 768 + padding of the widest map
    struct Tw { v2f pb[3], pc[3], pd[11]; }; // passes B (Ns=4), C (Ns=16): r = 1..3;  pass D (Ns=64, R=12): r = 1..11
    // WITH_PD = false leaves the eleven radix-12 twiddles out of the registers: the caller keeps them in an LDS table instead
    // (`pd_lds[(r-1)*64 + lane]`, fill_pd) and passes it to run() -- the Griffin-Lim body at n_fft 1536, whose 330 live values
    // do not fit the 256 registers that let two workgroups share a CU.
    template <bool WITH_PD = true>
    static __device__ __forceinline__ void load(Tw& tw, const v2f* __restrict__ twc, int lane) {
#pragma unroll
        for (int r = 1; r < 4; ++r) {
            tw.pb[r - 1] = twc[(lane & 3) * r * 48];          // exp(-2 pi i (j%4) r / 16)
            tw.pc[r - 1] = twc[(lane & 15) * r * 12];         // exp(-2 pi i (j%16) r / 64)
        }
        if (WITH_PD) {
#pragma unroll
            for (int r = 1; r < 12; ++r) tw.pd[r - 1] = twc[lane * r];         // exp(-2 pi i j r / 768)
        }
    }
    static constexpr int kPdTable = 11 * 64;                  // complex entries of the LDS twiddle table
    static __device__ __forceinline__ void fill_pd(v2f* pd_lds, const v2f* __restrict__ twc, int lane) {
#pragma unroll
        for (int r = 1; r < 12; ++r) pd_lds[(r - 1) * 64 + lane] = twc[lane * r];
    }
    static __device__ __forceinline__ int padA(int c) { return c + (c >> 4); }
    static __device__ __forceinline__ int padB(int c) { return c + ((c >> 4) << 2); }

    // one radix-4 pass over the three butterflies of a lane (registers v[g], v[g+3], v[g+6], v[g+9])
    template <bool INV, bool TWIDDLE>
    static __device__ __forceinline__ void pass4(v2f (&v)[12], const v2f (&w)[3]) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            if (TWIDDLE) {
                v[g + 3] = twmul<INV>(v[g + 3], w[0]);
                v[g + 6] = twmul<INV>(v[g + 6], w[1]);
                v[g + 9] = twmul<INV>(v[g + 9], w[2]);
            }
            dft4<INV>(v[g], v[g + 3], v[g + 6], v[g + 9]);
        }
    }

    template <bool INV>
    static __device__ __forceinline__ void run(v2f (&v)[12], const Tw& tw, v2f* tile, int lane, const v2f* pd_lds = nullptr) {
        // pass A (R=4, Ns=1): out[4 j + r], j = lane + 64 g
        pass4<INV, false>(v, tw.pb);
        wave_sync();
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) tile[padA(4 * (lane + 64 * g) + r)] = v[g + 3 * r];
        wave_sync();
#pragma unroll
        for (int t = 0; t < 12; ++t) v[t] = tile[padA(lane + 64 * t)];
        // pass B (R=4, Ns=4): out[16 (j>>2) + (j&3) + 4 r]
        pass4<INV, true>(v, tw.pb);
        wave_sync();
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) tile[padB(16 * ((lane + 64 * g) >> 2) + (lane & 3) + 4 * r)] = v[g + 3 * r];
        wave_sync();
#pragma unroll
        for (int t = 0; t < 12; ++t) v[t] = tile[padB(lane + 64 * t)];
        // pass C (R=4, Ns=16): out[64 (j>>4) + (j&15) + 16 r]  (conflict free without padding)
        pass4<INV, true>(v, tw.pc);
        wave_sync();
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) tile[64 * ((lane + 64 * g) >> 4) + (lane & 15) + 16 * r] = v[g + 3 * r];
        wave_sync();
#pragma unroll
        for (int t = 0; t < 12; ++t) v[t] = tile[lane + 64 * t];
        // pass D (R=12, Ns=64): one butterfly per lane, output index lane + 64 r stays in registers
        if (pd_lds != nullptr) {
#pragma unroll
            for (int r = 1; r < 12; ++r) v[r] = twmul<INV>(v[r], pd_lds[(r - 1) * 64 + lane]);
        } else {
#pragma unroll
            for (int r = 1; r < 12; ++r) v[r] = twmul<INV>(v[r], tw.pd[r - 1]);
        }
        dft12<INV>(v);
    }
};

// ---- geometry of the real transform ----------------------------------------------------------------
template <int NFFT> struct Geo {
    static constexpr int kNR = NFFT;             // real length (n_fft)
    static constexpr int kNC = NFFT / 2;         // complex FFT length
    static constexpr int kHop = NFFT / 2;
    static constexpr int kBins = NFFT / 2 + 1;
    using Fft = WaveFft<NFFT / 2>;
    static constexpr int kNV = Fft::kNV;         // complex values per lane
    static constexpr int kNP = kNV / 2;          // bin pairs (k, NC-k) per lane
    static constexpr int kTile = Fft::kTile;
};

// ---- Hermitian split / merge for the real transform of length 2 NC ---------------------------------
// z[m] = x[2m] + i x[2m+1], Z = FFT_NC(z), W = exp(-2 pi i / (2 NC)).  For k = 0..NC (Z[NC] := Z[0])
//     X[k]      = 1/2 (Z[k] + conj Z[NC-k])  -  i W^k 1/2 (Z[k] - conj Z[NC-k])
//     X[NC-k]   = conj( 1/2 (Z[k] + conj Z[NC-k])  +  i W^k 1/2 (Z[k] - conj Z[NC-k]) )
// and in the other direction (Im X[0], Im X[NC] ignored, as C2R transforms do)
//     Z[k]      = 1/2 (X[k] + conj X[NC-k])  +  i conj(W^k) 1/2 (X[k] - conj X[NC-k])
//     Z[NC-k]   = conj(1/2 (X[k] + conj X[NC-k]))  +  i conj( conj(W^k) 1/2 (X[k] - conj X[NC-k]) )
// IFFT_NC(Z) then yields x[2m] + i x[2m+1] (times NC; the 1/NC is folded into the synthesis window).
//
// Bins k and NC-k are produced together, so the lane that owns k = lane + 64 t (t < NP) also owns
// NC-k ("pair order"): the split, any per-bin work and the merge are then lane-local and only the
// FFT-order <-> pair-order hand-off crosses lanes.  That hand-off is the fixed involution
// (lane j, reg t) <-> (lane 64-j, reg NV-1-t), done with ds_bpermute (no LDS memory, no barrier).
// Lane 0 pairs with itself: (0,NC), (64,NC-64), .. and the self-paired bin NC/2.
// `wkh[t]` = 1/2 W^k for k = lane + 64 t, t = 0..NP-1.
__device__ __forceinline__ v2f shfl2(v2f v, int src) { return mk2(__shfl(v[0], src), __shfl(v[1], src)); }

// Lane 0 is its own partner and its pairs sit one register further (Z[NC - 64 t] = own v[NV - t]); both directions handle that with
// per-value selects on the sending / receiving side (v_cndmask), not with a divergent block: a block forces `s_waitcnt lgkmcnt(0)` at its
// head, i.e. every permute has to be back before the first pair can be combined.
__device__ __forceinline__ v2f sel2(bool c, v2f a, v2f b) { return mk2(c ? a[0] : b[0], c ? a[1] : b[1]); }

// Forward: v[t] = Z[lane + 64 t] -> lo[t] = X[k], hi[t] = X[NC-k]; mid = X[NC/2] (meaningful in lane 0).
template <int NV>
__device__ __forceinline__ void rfft_split_pairs(const v2f (&v)[NV], const v2f (&wkh)[NV / 2], int lane,
                                                 v2f (&lo)[NV / 2], v2f (&hi)[NV / 2], v2f& mid) {
    constexpr int NP = NV / 2;
    const int partner = (64 - lane) & 63;
    const bool l0 = lane == 0;
    v2f zp[NP];
#pragma unroll
    for (int t = 0; t < NP; ++t) zp[t] = shfl2(sel2(l0, t == 0 ? v[0] : v[NV - t], v[NV - 1 - t]), partner);
#pragma unroll
    for (int t = 0; t < NP; ++t) {
        const v2f s = cadd_conj(v[t], zp[t]);                  // Z + conj Zp
        const v2f wd = cmul(wkh[t], csub_conj(v[t], zp[t]));   // 1/2 W^k (Z - conj Zp)
        lo[t] = chalf_add_mi(s, wd);                           // s/2 - i wd
        hi[t] = cconj_half_add_pi(s, wd);                      // conj(s/2 + i wd)
    }
    mid = mk2(v[NP][0], -v[NP][1]);                            // X[NC/2] = conj Z[NC/2]
}

// Inverse: lo[t] = X[k], hi[t] = X[NC-k], mid = X[NC/2] -> v[t] = Z[lane + 64 t] ready for IFFT_NC.
template <int NV>
__device__ __forceinline__ void irfft_merge_pairs(v2f (&lo)[NV / 2], v2f (&hi)[NV / 2], v2f mid,
                                                  const v2f (&wkh)[NV / 2], int lane, v2f (&v)[NV]) {
    constexpr int NP = NV / 2;
    const bool l0 = lane == 0;
    lo[0][1] = l0 ? 0.0f : lo[0][1];                           // Im X[0], Im X[NC] are ignored (C2R convention)
    hi[0][1] = l0 ? 0.0f : hi[0][1];
    v2f zh[NP];
#pragma unroll
    for (int t = 0; t < NP; ++t) {
        const v2f s = cadd_conj(lo[t], hi[t]);                       // X + conj Xp
        const v2f dd = cmul_conj(csub_conj(lo[t], hi[t]), wkh[t]);   // 1/2 conj(W^k) (X - conj Xp)
        v[t] = chalf_add_pi(s, dd);                                  // s/2 + i dd
        zh[t] = chalf_conj_add_iconj(s, dd);                         // conj(s/2) + i conj(dd)
    }
    const int partner = (64 - lane) & 63;
    v2f r[NP];
#pragma unroll
    for (int t = 0; t < NP; ++t) r[t] = shfl2(zh[t], partner);
    // lanes 1..63: v[NV-1-t] = r[t];  lane 0 (received its own zh): v[NV-t] = r[t] for t >= 1 and v[NP] = conj(mid)
    const v2f cm = mk2(mid[0], -mid[1]);
#pragma unroll
    for (int t = 0; t < NP; ++t) v[NV - 1 - t] = sel2(l0, t + 1 < NP ? r[t + 1] : cm, r[t]);
}

}  // namespace dn
