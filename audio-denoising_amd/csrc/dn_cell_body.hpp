// dn_cell_body.hpp -- device body of the GRUUNet2 forward (see dn_cell.hip for the description).
#pragma once
#include <dn_cpx.hpp>

#include "dn_internal.hpp"

namespace dn {

// Diagnostic build only (make probe): s_memtime stamps of the phases of one forward of workgroup 0 (tools/cell_probe.py).
#ifdef DN_PROBE
static __device__ unsigned long long g_cell_probe[32];
#define DN_CSTAMP(id)                                                                             \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        unsigned long long t_;                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
        if (b == 0 && tid == 0) g_cell_probe[id] = t_;                                            \
        __builtin_amdgcn_sched_barrier(0);                                                        \
    } while (0)
// the same from inside a conv tile of the level under the magnifying glass (DN_PROBE_LEVEL: CIN * 100 + COUT of an encoder level)
#define DN_CSTAMP_IN(id)                                                                          \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        unsigned long long t_;                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_cell_probe[id] = t_;                           \
        __builtin_amdgcn_sched_barrier(0);                                                        \
    } while (0)
#else
#define DN_CSTAMP(id) do { } while (0)
#define DN_CSTAMP_IN(id) do { } while (0)
#endif

// Rows 16 mt + 4 q + r (r = 0..3) of one accumulator tile -> relu -> dst[row * stride]; see the note on exec-mask regions below.
template <int COUT>
__device__ __forceinline__ void store_rows(const f32x4& acc, float* dst, int stride, int mt, int q, float* trash) {
    constexpr int kRows = COUT % 16;                 // rows of the last row tile when it is a partial one
    if (kRows == 0 || (mt + 1) * 16 <= COUT) {       // wave-uniform
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(mt * 16 + q * 4 + r) * stride] = fmaxf(acc[r], 0.0f);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r >= kRows) continue;                // 4 q + r >= r
            float* d = 4 * q + r < kRows ? dst + (mt * 16 + q * 4 + r) * stride : trash;
            *d = fmaxf(acc[r], 0.0f);
        }
    }
}
template <int COUT>
__device__ __forceinline__ void store_rows2(const f32x4& ev, const f32x4& od, float* dst, int stride, int mt, int q, float* trash) {
    constexpr int kRows = COUT % 16;
    if (kRows == 0 || (mt + 1) * 16 <= COUT) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<float2*>(dst + (mt * 16 + q * 4 + r) * stride) = make_float2(fmaxf(ev[r], 0.0f), fmaxf(od[r], 0.0f));
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r >= kRows) continue;
            float* d = 4 * q + r < kRows ? dst + (mt * 16 + q * 4 + r) * stride : trash;
            *reinterpret_cast<float2*>(d) = make_float2(fmaxf(ev[r], 0.0f), fmaxf(od[r], 0.0f));
        }
    }
}

// ---- Conv1d k3 s2 p1 (+ folded position bias, relu) as an MFMA contraction.
//   in [TT][CIN][2*lout] (LDS) -> out [TT][COUT][lout] (LDS)
//   afrag: [MTILES][KS][64] weight fragments, K order = tap-major, channels in groups of 4
//          (CIN = 1: one k-step whose 4 K slots are the 3 taps + a zero)
//   work split: tile = (n-tile of 16 items) x (group of MT m-tiles); tiles are dealt to waves round robin.
//
// The levels are issue bound, not MFMA bound (a phase is a few hundred instructions of one wavefront around 15-60 MFMAs), so the
// tiles avoid exec-mask regions (four scalar instructions and a branch each):
//   * lanes past the last item work on item 0 again and hold exactly item 0's results -- their stores rewrite the same values;
//   * accumulator rows that do not exist (o >= COUT) start from the last real row's bias and are never stored;
//   * the rows of the last, partial row tile that do not exist are stored to `trash` (an LDS word nobody reads): a select on the
//     address instead of a predicated store;
//   * operands next to an edge are loaded unconditionally (one word outside the buffer at most, still inside the LDS plan) and
//     zeroed by a select.
template <int NW, int CIN, int COUT, int MT>
__device__ __forceinline__ void mconv_down(const float* afrag, const float* bt, const float* in,
                                           float* out, float* trash, int lout, int tt, int wv, int lane) {
    constexpr int CS = (CIN + 3) / 4;
    constexpr int KS = CIN == 1 ? 1 : 3 * CS;
    constexpr int MTILES = (COUT + 15) / 16;
    constexpr int MG = MTILES / MT;
    const int lin = 2 * lout, items = tt * lout, ntiles = (items + 15) >> 4;
    const int q = lane >> 4, jl = lane & 15;
    for (int tile = wv; tile < ntiles * MG; tile += NW) {
        const int nt = tile % ntiles, mt0 = (tile / ntiles) * MT;
        const int item = nt * 16 + jl;
        const bool valid = item < items;
        const int itc = valid ? item : 0;
        const int t = itc / lout, p = itc - t * lout;
        f32x4 acc[MT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = (mt0 + mi) * 16 + q * 4 + r;
                acc[mi][r] = bt[min(o, COUT - 1) * lout + p];
            }
        const float* af = afrag + (size_t)mt0 * KS * 64 + lane;
        if (CIN == 1) {
            const int idx = t * lin + 2 * p - 1 + q;
            const bool ok = q < 3 && !(q == 0 && p == 0);
            float b = in[idx];
            b = ok ? b : 0.0f;
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) acc[mi] = mfma16(af[mi * KS * 64], b, acc[mi]);
        } else {
            // every operand of the tile is requested before the first MFMA: fetched a k-step at a time, each step paid an LDS round
            // trip (~150 cycles) in front of its 32-cycle MFMA
            const float* base = in + (size_t)t * CIN * lin + 2 * p - 1;
            float bv[KS], av[MT][KS];
#pragma unroll
            for (int tap = 0; tap < 3; ++tap)
#pragma unroll
                for (int cs = 0; cs < CS; ++cs) {
                    const int c = min(4 * cs + q, CIN - 1);          // padded K slots carry zero weights
                    float b = base[c * lin + tap];
                    if (tap == 0 && p == 0) b = 0.0f;                // left zero padding of the conv
                    bv[tap * CS + cs] = b;
                }
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) av[mi][ks] = af[(mi * KS + ks) * 64];
            __builtin_amdgcn_sched_barrier(0);
            if (CIN == kHidden && COUT == kHidden && MT == 2) DN_CSTAMP_IN(18);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) acc[mi] = mfma16(av[mi][ks], bv[ks], acc[mi]);
            if (CIN == kHidden && COUT == kHidden && MT == 2) DN_CSTAMP_IN(19);
        }
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) store_rows<COUT>(acc[mi], out + (size_t)t * COUT * lout + p, lout, mt0 + mi, q, trash);
    }
}

// ---- bf16 variants (BASELINE config 3): conv inputs and weights rounded to bf16, fp32 accumulate, on
// v_mfma_f32_16x16x32_bf16.  One k-step = 32 K slots = channels 8 q + j (q = lane >> 4, j = 0..7) of one tap, so a
// 17-channel level needs 3 k-steps instead of 15; activations stay fp32 in LDS and are rounded when the B fragment is built.
__device__ __forceinline__ f32x4 mfma16_bf16(const bf16x8& a, const bf16x8& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <int NW, int CIN, int COUT, int MT>
__device__ __forceinline__ void mconv_down_bf16(const void* afrag, const float* bt, const float* in,
                                                float* out, float* trash, int lout, int tt, int wv, int lane) {
    constexpr int KS = CIN == 1 ? 1 : 3;
    constexpr int MTILES = (COUT + 15) / 16;
    constexpr int MG = MTILES / MT;
    const bf16x8* af8 = static_cast<const bf16x8*>(afrag);
    const int lin = 2 * lout, items = tt * lout, ntiles = (items + 15) >> 4;
    const int q = lane >> 4, jl = lane & 15;
    for (int tile = wv; tile < ntiles * MG; tile += NW) {
        const int nt = tile % ntiles, mt0 = (tile / ntiles) * MT;
        const int item = nt * 16 + jl;
        const bool valid = item < items;
        const int itc = valid ? item : 0;
        const int t = itc / lout, p = itc - t * lout;
        f32x4 acc[MT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = (mt0 + mi) * 16 + q * 4 + r;
                acc[mi][r] = bt[min(o, COUT - 1) * lout + p];
            }
        if (CIN == 1) {
            bf16x8 b;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool ok = q == 0 && j < 3 && !(j == 0 && p == 0);
                const float x = j < 3 ? in[t * lin + 2 * p - 1 + j] : 0.0f;
                b[j] = f2bf(ok ? x : 0.0f);
            }
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) acc[mi] = mfma16_bf16(af8[((mt0 + mi) * KS) * 64 + lane], b, acc[mi]);
        } else {
            const float* base = in + (size_t)t * CIN * lin + 2 * p - 1;
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                bf16x8 b;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = 8 * q + j;
                    float x = base[min(c, CIN - 1) * lin + tap];
                    if (c >= CIN || (tap == 0 && p == 0)) x = 0.0f;
                    b[j] = f2bf(x);
                }
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) acc[mi] = mfma16_bf16(af8[((mt0 + mi) * KS + tap) * 64 + lane], b, acc[mi]);
            }
        }
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) store_rows<COUT>(acc[mi], out + (size_t)t * COUT * lout + p, lout, mt0 + mi, q, trash);
    }
}

template <int NW, bool SKIP, int MT>
__device__ __forceinline__ void mconv_up_bf16(const void* afrag, const float* bt, const float* a,
                                              const float* skip, float* out, float* trash, int l, int tt, int wv, int lane) {
    constexpr int PARTS = SKIP ? 2 : 1;
    constexpr int MTILES = 2, MG = MTILES / MT;
    const bf16x8* af8 = static_cast<const bf16x8*>(afrag);
    const int lo = 2 * l, items = tt * l, ntiles = (items + 15) >> 4;
    const int q = lane >> 4, jl = lane & 15;
    for (int tile = wv; tile < ntiles * MG; tile += NW) {
        const int nt = tile % ntiles, mt0 = (tile / ntiles) * MT;
        const int item = nt * 16 + jl;
        const bool valid = item < items;
        const int itc = valid ? item : 0;
        const int t = itc / l, i = itc - t * l;
        const bool has_next = i + 1 < l;
        f32x4 ev[MT], od[MT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = min((mt0 + mi) * 16 + q * 4 + r, kHidden - 1);
                const float2 bb = *reinterpret_cast<const float2*>(bt + o * lo + 2 * i);
                ev[mi][r] = bb.x;
                od[mi][r] = bb.y;
            }
#pragma unroll
        for (int part = 0; part < PARTS; ++part) {
            const float* src = (part == 0 ? a : skip) + (size_t)t * kHidden * l + i;
            bf16x8 x0, x1;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = 8 * q + j;
                const bool ok = c < kHidden;
                const float* sc = src + min(c, kHidden - 1) * l;
                const float s0 = sc[0], s1 = sc[1];
                x0[j] = f2bf(ok ? s0 : 0.0f);
                x1[j] = f2bf(ok && has_next ? s1 : 0.0f);
            }
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                const bf16x8* am = af8 + ((size_t)(mt0 + mi) * 3 * PARTS + part) * 64 + lane;    // [mt][set][part][64]
                ev[mi] = mfma16_bf16(am[0 * PARTS * 64], x0, ev[mi]);
                od[mi] = mfma16_bf16(am[1 * PARTS * 64], x0, od[mi]);
                od[mi] = mfma16_bf16(am[2 * PARTS * 64], x1, od[mi]);
            }
        }
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
            store_rows2<kHidden>(ev[mi], od[mi], out + (size_t)t * kHidden * lo + 2 * i, lo, mt0 + mi, q, trash);
    }
}

// ---- ConvTranspose1d k3 s2 p1 output_padding 1 (L -> 2L) on cat(a, skip) data channels (+ folded position
// bias, relu).  a [TT][17][l], skip [TT][17][l] (absent at the first level) -> out [TT][17][2l].
// Item = (t, input position i):  out[2i]   = sum_c w[c][o][1] x[c][i]
//                                out[2i+1] = sum_c w[c][o][2] x[c][i] + w[c][o][0] x[c][i+1]
// afrag: [MTILES=2][3 tap sets: k=1, k=2, k=0][KSU][64]; K order = part-major (a, then skip), channels in fours.
template <int NW, bool SKIP, int MT, int COUT = kHidden>
__device__ __forceinline__ void mconv_up(const float* afrag, const float* bt, const float* a,
                                         const float* skip, float* out, float* trash, int l, int tt, int wv, int lane) {
    constexpr int CS = 5;                      // ceil(17 / 4)
    constexpr int KSU = SKIP ? 2 * CS : CS;
    constexpr int MTILES = (COUT + 15) / 16, MG = MTILES / MT;
    const int lo = 2 * l, items = tt * l, ntiles = (items + 15) >> 4;
    const int q = lane >> 4, jl = lane & 15;
    for (int tile = wv; tile < ntiles * MG; tile += NW) {
        const int nt = tile % ntiles, mt0 = (tile / ntiles) * MT;
        const int item = nt * 16 + jl;
        const bool valid = item < items;
        const int itc = valid ? item : 0;
        const int t = itc / l, i = itc - t * l;
        const bool has_next = i + 1 < l;
        f32x4 ev[MT], od[MT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = min((mt0 + mi) * 16 + q * 4 + r, COUT - 1);
                const float2 bb = *reinterpret_cast<const float2*>(bt + o * lo + 2 * i);
                ev[mi][r] = bb.x;
                od[mi][r] = bb.y;
            }
        const float* af = afrag + (size_t)mt0 * 3 * KSU * 64 + lane;
#pragma unroll
        for (int part = 0; part < (SKIP ? 2 : 1); ++part) {
            // all operands of one part (a, then skip) are requested before its first MFMA (see mconv_down)
            const float* src = (part == 0 ? a : skip) + (size_t)t * kHidden * l + i;
            float x0[CS], x1[CS], av[MT][3][CS];
#pragma unroll
            for (int cs = 0; cs < CS; ++cs) {
                const int c = min(4 * cs + q, kHidden - 1);
                x0[cs] = src[c * l];
                const float xn = src[c * l + 1];
                x1[cs] = has_next ? xn : 0.0f;
            }
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int set = 0; set < 3; ++set)
#pragma unroll
                    for (int cs = 0; cs < CS; ++cs) av[mi][set][cs] = af[((size_t)(mi * 3 + set) * KSU + part * CS + cs) * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cs = 0; cs < CS; ++cs)
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    ev[mi] = mfma16(av[mi][0][cs], x0[cs], ev[mi]);
                    od[mi] = mfma16(av[mi][1][cs], x0[cs], od[mi]);
                    od[mi] = mfma16(av[mi][2][cs], x1[cs], od[mi]);
                }
        }
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
            store_rows2<COUT>(ev[mi], od[mi], out + (size_t)t * COUT * lo + 2 * i, lo, mt0 + mi, q, trash);
    }
}

// ---- the last decoder level (ConvTranspose1d 34 -> 1, linear: gruunet2.py:94-96, 242) on the VALU.  One output channel is one
// useful row of a 16-row MFMA tile: as a contraction it cost 60 MFMAs a wavefront (1,920 cycles) for 120 x 2 outputs.  Here a thread
// owns one item (t, input position i) and one half of the input channels (a: wavefronts 0-1, skip: wavefronts 2-3): 17 channels x
// 3 taps of FMAs on operands read straight from LDS (weights: one broadcast 16-byte read per channel), the skip half handed over
// through `part`.  wt: [2 parts][17][4] = w[k=1], w[k=2], w[k=0], 0.
template <int NW>
__device__ __forceinline__ void last_up(const float* wt, const float* bt, const float* a, const float* skip, float* out,
                                        float* part, int l, int tt, int tid, size_t out_t_stride) {
    static_assert(NW == 4, "two wavefronts per half of the channels");
    const int h = tid >> 7, item = tid & 127, items = tt * l;      // items <= 3 * 8 * kMaxC = 120
    const int itc = item < items ? item : 0;
    const int t = itc / l, i = itc - t * l;
    const bool has_next = i + 1 < l;
    const float* src = (h == 0 ? a : skip) + (size_t)t * kHidden * l + i;
    const float4* w4 = reinterpret_cast<const float4*>(wt) + h * kHidden;
    float ev = 0.0f, od = 0.0f;
#pragma unroll
    for (int c = 0; c < kHidden; ++c) {
        const float4 w = w4[c];
        const float x0 = src[c * l];
        float x1 = src[c * l + 1];
        x1 = has_next ? x1 : 0.0f;
        ev = fmaf(w.x, x0, ev);
        od = fmaf(w.y, x0, od);
        od = fmaf(w.z, x1, od);
    }
    if (h == 1) *reinterpret_cast<float2*>(part + 2 * item) = make_float2(ev, od);
    DN_LDS_BARRIER();
    if (h == 0 && item < items) {
        const float2 ps = *reinterpret_cast<const float2*>(part + 2 * item);
        const float2 bb = *reinterpret_cast<const float2*>(bt + 2 * i);
        *reinterpret_cast<float2*>(out + (size_t)t * out_t_stride + 2 * i) = make_float2(bb.x + (ev + ps.x), bb.y + (od + ps.y));
    }
}

// GRU gate nonlinearities on the hardware transcendental units (v_exp_f32, v_rcp_f32: 1 ulp each) instead of libm's expf / tanhf and
// an IEEE divide: ~6 instructions a gate instead of ~40, on the serial path of every time step.  Absolute error <= ~2e-7 (the gates
// enter the state linearly, so absolute error is what matters); saturates cleanly (exp2 -> inf / 0 gives exactly 0 / 1 / -1).
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x)); }
__device__ __forceinline__ float tanhf_(float x) {
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * x)), 1.0f);
}

// LDS plan (floats), C = compressed bins, per chunk of TT <= 3 steps.
struct CellLds {
    int x, d0, d1, d2, d3, h, gh, hi, u0, u1, u2, total;
    __host__ __device__ explicit CellLds(int C) {
        const int T = kCellChunk, F = 16 * C;
        int o = 8;                            // words 0..3: trash (stores of rows that do not exist); 4..7: zero guard in front of the first buffer (index -1 reads)
        x = o;  o += T * F;
        d0 = o; o += T * kHidden * 8 * C;
        d1 = o; o += T * kHidden * 4 * C;
        d2 = o; o += T * kHidden * 2 * C;
        d3 = o; o += T * kGates * C;
        h = o;  o += kHidden * C;
        gh = o; o += kGates * C;
        hi = o; o += T * kHidden * C;
        u0 = o; o += T * kHidden * 2 * C;
        u1 = o; o += T * kHidden * 4 * C;
        u2 = o; o += T * kHidden * 8 * C;
        total = o;
    }
};

constexpr int kCellActFloats = 8 + 3 * 16 * kMaxC + 3 * 17 * 14 * kMaxC * 2 + 3 * 51 * kMaxC + 17 * kMaxC + 51 * kMaxC + 3 * 17 * kMaxC;
// Weight fragments and bias tables are staged through LDS one level ahead (double buffered): while level L multiplies, every
// thread has level L+1's share of fragments in flight from L2 and drops it into the other buffer before the barrier.  The
// levels are a strictly serial chain of short phases (~1 us each): with the fragments fetched at the head of every phase the
// kernel spent 71 % of its time in s_waitcnt (profiles/r01_v4_pmc_sq.txt); staged, a phase starts on LDS-resident operands.
constexpr int kCellWFloats = 4096;               // largest level: 51 -> 4 row tiles x 15 k-steps x 64 = 3840 (= 2 x 3 x 10 x 64 of a skip decoder level), in whole rounds
constexpr int kCellBFloats = 768;                // largest bias table: 17 channels x 8C positions = 680 at C = 5, in whole rounds
constexpr int kCellLdsFloats = kCellActFloats + 2 * kCellWFloats + 2 * kCellBFloats;
constexpr int kCellSmem = 4 * kCellLdsFloats;
static_assert(kCellActFloats % 4 == 0, "fragment buffers are written 16 bytes at a time");

// One level's operands in HBM/L2: weight fragments (fp32 or bf16, counted in floats) and the bias table.
struct CellLevel { const float* w; int wn; const float* b; int bn; };

// A round = one load per thread (16 bytes of fragments / 4 bytes of bias).  With the sizes known at compile time (CT != 0; PAD) the
// stager moves WHOLE rounds: every load and every LDS store is unconditional, the tail of the last round reads what follows the
// array in its arena (the arenas end in a slack of kArenaSlack bytes) and lands in buffer space nobody reads.
template <int THREADS, bool PAD>
struct CellStager {
    static constexpr int kW4 = (kCellWFloats / 4 + THREADS - 1) / THREADS;     // float4 per thread
    static constexpr int kB1 = (kCellBFloats + THREADS - 1) / THREADS;         // floats per thread
    static_assert(!PAD || (kW4 * THREADS * 4 <= kCellWFloats && kB1 * THREADS <= kCellBFloats), "whole rounds must fit the buffers");
    float4 w[kW4];
    float b[kB1];
    __device__ __forceinline__ void issue(const CellLevel& lv, int tid) {
        const float4* w4 = reinterpret_cast<const float4*>(lv.w);
#pragma unroll
        for (int i = 0; i < kW4; ++i) {
            const int j = tid + THREADS * i;
            w[i] = (PAD ? 4 * THREADS * i < lv.wn : 4 * j < lv.wn) ? w4[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < kB1; ++i) {
            const int j = tid + THREADS * i;
            b[i] = (PAD ? THREADS * i < lv.bn : j < lv.bn) ? lv.b[j] : 0.0f;
        }
    }
    __device__ __forceinline__ void commit(const CellLevel& lv, float* wbuf, float* bbuf, int tid) const {
#pragma unroll
        for (int i = 0; i < kW4; ++i) {
            const int j = tid + THREADS * i;
            if (PAD ? 4 * THREADS * i < lv.wn : 4 * j < lv.wn) reinterpret_cast<float4*>(wbuf)[j] = w[i];
        }
#pragma unroll
        for (int i = 0; i < kB1; ++i) {
            const int j = tid + THREADS * i;
            if (PAD ? THREADS * i < lv.bn : j < lv.bn) bbuf[j] = b[i];
        }
    }
};

// One workgroup of NW wavefronts runs the T-step forward of stream `b`.  `smem`: kCellSmem bytes of LDS.
// CT = the number of compressed bins when it is known at compile time (5: 80 mels, 4: 64 mels), 0 = use the run-time value.
// With CT every length, stride and item/length division of the conv tiles is a constant: the per-k-step address arithmetic
// (which, not the MFMAs, was what a phase spent its issue slots on) folds into immediate LDS offsets.
// STAGE = false: every level reads its fragments and bias table straight from L2 -- no staging loads, no LDS copies, 35 KB of LDS instead of 74
// (kCellSmemUnstaged): what lets FOUR front workgroups share a CU when they run as a launch of their own (dn_hop.hip, FRONT).
#ifdef DN_CELL_NOSTAGE
constexpr bool kCellStageDefault = false;
#else
constexpr bool kCellStageDefault = true;
#endif
constexpr int kCellSmemUnstaged = 4 * kCellActFloats;
template <int NW, bool BF16 = false, int CT = 0, bool STAGE = kCellStageDefault>
__device__ __forceinline__ void cell_body(char* smem, const CellDev& cd, const float* __restrict__ x,
                                          const float* __restrict__ hx_in, float* __restrict__ out,
                                          float* __restrict__ hx_out, int T, int C_rt, size_t b, int tid,
                                          float hx_scale = 1.0f) {
    constexpr int kCellThreads = NW * 64;
    const int C = CT ? CT : C_rt;
    float* lds = reinterpret_cast<float*>(smem);
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int F = 16 * C;
    const CellLds L(C);
    float* sx = lds + L.x;   float* sd0 = lds + L.d0; float* sd1 = lds + L.d1; float* sd2 = lds + L.d2;
    float* sd3 = lds + L.d3; float* sh = lds + L.h;   float* sgh = lds + L.gh; float* shi = lds + L.hi;
    float* su0 = lds + L.u0; float* su1 = lds + L.u1; float* su2 = lds + L.u2;
    float* wbuf[2] = {lds + kCellActFloats, lds + kCellActFloats + kCellWFloats};
    float* bbuf[2] = {lds + kCellActFloats + 2 * kCellWFloats, lds + kCellActFloats + 2 * kCellWFloats + kCellBFloats};

    // the eight conv levels in execution order: encoder 0..3, decoder 0..3 (fragment sizes in floats; bf16 fragments are 16 B per lane)
    auto level = [&](int i) -> CellLevel {
        CellLevel lv;
        if (i < 4) {
            const int mt = i == 3 ? 4 : 2, ks32 = i == 0 ? 1 : 15, ks16 = i == 0 ? 1 : 3;
            lv.w = BF16 ? static_cast<const float*>(cd.wb_down[i]) : cd.w_down[i];
            lv.wn = BF16 ? mt * ks16 * 64 * 4 : mt * ks32 * 64;
            lv.b = cd.bt_down[i];
            lv.bn = (i == 3 ? kGates : kHidden) * (8 * C >> i);
        } else {
            const int l = i - 4, parts = l == 0 ? 1 : 2;
            const bool bf = BF16 && l < 3;                        // the single-channel last level stays fp32
            lv.w = bf ? static_cast<const float*>(cd.wb_up[l]) : cd.w_up[l];
            lv.wn = bf ? 2 * 3 * parts * 64 * 4 : l == 3 ? 2 * kHidden * 4 : 2 * 3 * parts * 5 * 64;
            lv.b = cd.bt_up[l];
            lv.bn = (l == 3 ? 1 : kHidden) * (2 * C << l);
        }
        return lv;
    };
    // two register sets: level L+2 is requested while level L multiplies and level L+1 (requested a phase earlier) is dropped
    // into its buffer -- every request has more than a whole phase to come back from L2
    CellStager<kCellThreads, CT != 0 && kCellThreads == 256> stgA, stgB;
    // (build knob DN_CELL_NOSTAGE flips the default of STAGE)
#define DN_STG_ISSUE(s, i) do { if constexpr (STAGE) s.issue(level(i), tid); } while (0)
#define DN_STG_COMMIT(s, i) do { if constexpr (STAGE) s.commit(level(i), wbuf[(i) & 1], bbuf[(i) & 1], tid); } while (0)
#define DN_LW(i) (STAGE ? static_cast<const float*>(wbuf[(i) & 1]) : level(i).w)
#define DN_LB(i) (STAGE ? static_cast<const float*>(bbuf[(i) & 1]) : level(i).b)
    DN_STG_ISSUE(stgA, 0);
    DN_STG_ISSUE(stgB, 1);

    DN_CSTAMP(0);
    if (tid < 8) lds[tid] = 0.0f;
    float* trash = lds;
    // hidden state -> LDS (gruunet2.py:294-301: zeros when the caller passes none)
    for (int i = tid; i < kHidden * C; i += kCellThreads) sh[i] = hx_in != nullptr ? hx_in[b * kHidden * C + i] : 0.0f;

    // hidden-gate conv gh = relu(conv k3 s1 p1 (hx) + position bias): rows = 51 gate channels in four row tiles dealt to the
    // waves, columns = the C positions, K = (channel, tap).  Its fragments and bias are requested here, last in the prologue
    // (they are not needed before the encoder is through and the LDS-only barriers below let them arrive meanwhile), and
    // stay in VGPRs for every time step.
    constexpr int kGhKS = 13, kGhTiles = (4 + NW - 1) / NW;
    float agh[kGhTiles][kGhKS];
    f32x4 bgh[kGhTiles];
    float xin[(kCellChunk * 16 * kMaxC + kCellThreads - 1) / kCellThreads];
    constexpr int kXR = (kCellChunk * 16 * kMaxC + kCellThreads - 1) / kCellThreads;
    {
        const int tt0 = min(kCellChunk, T);
#pragma unroll
        for (int r = 0; r < kXR; ++r) {
            const int i = tid + kCellThreads * r;
            xin[r] = i < tt0 * F ? x[(b * T) * F + i] : 0.0f;
        }
    }
    {
        const int q = lane >> 4, p = lane & 15;
#pragma unroll
        for (int g = 0; g < kGhTiles; ++g) {
            const int mt = wv + NW * g;                                  // wave-uniform
#pragma unroll
            for (int ks = 0; ks < kGhKS; ++ks) agh[g][ks] = mt < 4 ? cd.w_gh[((size_t)mt * kGhKS + ks) * 64 + lane] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = mt * 16 + 4 * q + r;
                bgh[g][r] = (mt < 4 && o < kGates && p < C) ? cd.bt_gh[o * C + p] : 0.0f;
            }
        }
    }

    for (int t0 = 0; t0 < T; t0 += kCellChunk) {
        const int tt = min(kCellChunk, T - t0);
        DN_LDS_BARRIER();
        if (t0 > 0) {
#pragma unroll
            for (int r = 0; r < kXR; ++r) {
                const int i = tid + kCellThreads * r;
                xin[r] = i < tt * F ? x[(b * T + t0) * F + i] : 0.0f;
            }
        }
#pragma unroll
        for (int r = 0; r < kXR; ++r) {
            const int i = tid + kCellThreads * r;
            if (i < tt * F) sx[i] = xin[r];
        }
        DN_STG_COMMIT(stgA, 0);
        DN_LDS_BARRIER();
        DN_CSTAMP(1);
        // ---- encoder, batched over the chunk (gruunet2.py:136-144); level i multiplies out of buffer i & 1
        DN_STG_ISSUE(stgA, 2);
        if (BF16) mconv_down_bf16<NW, 1, kHidden, 2>(DN_LW(0), DN_LB(0), sx, sd0, trash, 8 * C, tt, wv, lane);
        else mconv_down<NW, 1, kHidden, 2>(DN_LW(0), DN_LB(0), sx, sd0, trash, 8 * C, tt, wv, lane);
        DN_STG_COMMIT(stgB, 1);
        DN_LDS_BARRIER();
        DN_CSTAMP(2);
        DN_STG_ISSUE(stgB, 3);
        DN_CSTAMP(17);
        if (BF16) mconv_down_bf16<NW, kHidden, kHidden, 2>(DN_LW(1), DN_LB(1), sd0, sd1, trash, 4 * C, tt, wv, lane);
        else mconv_down<NW, kHidden, kHidden, 2>(DN_LW(1), DN_LB(1), sd0, sd1, trash, 4 * C, tt, wv, lane);
        DN_CSTAMP(20);
        DN_STG_COMMIT(stgA, 2);
        DN_CSTAMP(21);
        DN_LDS_BARRIER();
        DN_CSTAMP(3);
        DN_STG_ISSUE(stgA, 4);
        if (BF16) mconv_down_bf16<NW, kHidden, kHidden, 1>(DN_LW(2), DN_LB(2), sd1, sd2, trash, 2 * C, tt, wv, lane);
        else mconv_down<NW, kHidden, kHidden, 1>(DN_LW(2), DN_LB(2), sd1, sd2, trash, 2 * C, tt, wv, lane);
        DN_STG_COMMIT(stgB, 3);
        DN_LDS_BARRIER();
        DN_CSTAMP(4);
        if (BF16) mconv_down_bf16<NW, kHidden, kGates, 1>(DN_LW(3), DN_LB(3), sd2, sd3, trash, C, tt, wv, lane);
        else mconv_down<NW, kHidden, kGates, 1>(DN_LW(3), DN_LB(3), sd2, sd3, trash, C, tt, wv, lane);
        DN_STG_COMMIT(stgA, 4);
        DN_LDS_BARRIER();
        DN_CSTAMP(5);
        // ---- recurrent part, sequential in t (gruunet2.py:232-240)
        for (int t = 0; t < tt; ++t) {
            {
                const int q = lane >> 4, p = lane & 15;
                f32x4 acc[kGhTiles];
#pragma unroll
                for (int g = 0; g < kGhTiles; ++g) acc[g] = bgh[g];
                float bv[kGhKS];
#pragma unroll
                for (int ks = 0; ks < kGhKS; ++ks) {
                    const int kk = 4 * ks + q, c = kk / 3, k = kk - 3 * c, src = p - 1 + k;
                    const float v = sh[c * C + src];   // at most one word in front of / a few behind `sh`: inside the plan
                    bv[ks] = (c < kHidden && p < C && src >= 0 && src < C) ? v : 0.0f;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < kGhKS; ++ks)
#pragma unroll
                    for (int g = 0; g < kGhTiles; ++g)
                        if (wv + NW * g < 4) acc[g] = mfma16(agh[g][ks], bv[ks], acc[g]);
#pragma unroll
                for (int g = 0; g < kGhTiles; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int o = (wv + NW * g) * 16 + 4 * q + r;
                        if (wv + NW * g < 4 && o < kGates && p < C) sgh[o * C + p] = fmaxf(acc[g][r], 0.0f);
                    }
            }
            // (requested here and not a phase earlier: the first use of the pinned gate fragments above waits for every load in
            // flight, and this one would be a phase old; it is not needed before the second decoder level)
            if (t == 0) DN_STG_ISSUE(stgB, 5);
            DN_LDS_BARRIER();
            if (tid < kHidden * C) {   // chunk order r, i, n (gruunet2.py:234-240)   (17 C <= 85 < threads)
                const float* gx = sd3 + (size_t)t * kGates * C;
                const float r = sigmoidf_(gx[tid] + sgh[tid]);
                const float z = sigmoidf_(gx[kHidden * C + tid] + sgh[kHidden * C + tid]);
                const float n = tanhf_(gx[2 * kHidden * C + tid] + r * sgh[2 * kHidden * C + tid]);
                const float hn = n + z * (sh[tid] - n);
                sh[tid] = hn;
                shi[t * kHidden * C + tid] = hn;
            }
            DN_LDS_BARRIER();
            DN_CSTAMP(6 + t);
        }
        // ---- decoder, batched over the chunk (gruunet2.py:184-199); skips are d2, d1, d0 (the last level has no cat)
        DN_STG_ISSUE(stgA, 6);
        if (BF16) mconv_up_bf16<NW, false, 1>(DN_LW(4), DN_LB(4), shi, nullptr, su0, trash, C, tt, wv, lane);
        else mconv_up<NW, false, 1>(DN_LW(4), DN_LB(4), shi, nullptr, su0, trash, C, tt, wv, lane);
        DN_STG_COMMIT(stgB, 5);      // requested four phases ago (buffer 1 has been free since the last encoder level)
        DN_STG_ISSUE(stgB, 7);
        DN_LDS_BARRIER();
        DN_CSTAMP(9);
        if (BF16) mconv_up_bf16<NW, true, 1>(DN_LW(5), DN_LB(5), su0, sd2, su1, trash, 2 * C, tt, wv, lane);
        else mconv_up<NW, true, 1>(DN_LW(5), DN_LB(5), su0, sd2, su1, trash, 2 * C, tt, wv, lane);
        DN_STG_COMMIT(stgA, 6);
        DN_LDS_BARRIER();
        DN_CSTAMP(10);
        const bool more = t0 + kCellChunk < T;
        if (more) DN_STG_ISSUE(stgA, 0);
        if (BF16) mconv_up_bf16<NW, true, 2>(DN_LW(6), DN_LB(6), su1, sd1, su2, trash, 4 * C, tt, wv, lane);
        else mconv_up<NW, true, 2>(DN_LW(6), DN_LB(6), su1, sd1, su2, trash, 4 * C, tt, wv, lane);
        DN_STG_COMMIT(stgB, 7);
        DN_LDS_BARRIER();
        DN_CSTAMP(11);
        // last level: one output channel, VALU, fp32 in both precisions; its rows are the model output.  The next chunk's first levels
        // are on their way meanwhile.  (`su0` has been dead since the second decoder level: it carries the skip half's partial sums.)
        if (more) DN_STG_ISSUE(stgB, 1);
        last_up<NW>(DN_LW(7), DN_LB(7), su2, sd0, out + (b * T + t0) * F, su0, 8 * C, tt, tid, (size_t)F);
    }
    __syncthreads();
    DN_CSTAMP(12);
    // (hx_scale != 1: the server variant's `hx = hx * 0.9`, server.py:214)
    for (int i = tid; i < kHidden * C; i += kCellThreads) hx_out[b * kHidden * C + i] = sh[i] * hx_scale;
}

}  // namespace dn
