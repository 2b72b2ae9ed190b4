// dn_cell_body.hpp -- device body of the GRUUNet2 forward (see dn_cell.hip for the description).
#pragma once
#include <dn_cpx.hpp>

#include "dn_internal.hpp"

namespace dn {


__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- Conv1d k3 s2 p1 (+ folded position bias, relu) as an MFMA contraction.
//   in [TT][CIN][2*lout] (LDS) -> out [TT][COUT][lout] (LDS)
//   afrag: [MTILES][KS][64] weight fragments, K order = tap-major, channels in groups of 4
//          (CIN = 1: one k-step whose 4 K slots are the 3 taps + a zero)
//   work split: tile = (n-tile of 16 items) x (group of MT m-tiles); tiles are dealt to waves round robin.
template <int NW, int CIN, int COUT, int MT>
__device__ __forceinline__ void mconv_down(const float* __restrict__ afrag, const float* __restrict__ bt, const float* in,
                                           float* out, int lout, int tt, int wv, int lane) {
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
                acc[mi][r] = o < COUT ? bt[o * lout + p] : 0.0f;
            }
        const float* af = afrag + (size_t)mt0 * KS * 64 + lane;
        if (CIN == 1) {
            const int idx = t * lin + 2 * p - 1 + q;
            const bool ok = q < 3 && !(q == 0 && p == 0);
            const float b = ok ? in[idx] : 0.0f;
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) acc[mi] = mfma16(af[mi * KS * 64], b, acc[mi]);
        } else {
            const float* base = in + (size_t)t * CIN * lin + 2 * p - 1;
#pragma unroll
            for (int tap = 0; tap < 3; ++tap)
#pragma unroll
                for (int cs = 0; cs < CS; ++cs) {
                    const int c = min(4 * cs + q, CIN - 1);          // padded K slots carry zero weights
                    float b = base[c * lin + tap];
                    if (tap == 0 && p == 0) b = 0.0f;                // left zero padding of the conv
                    const int ks = tap * CS + cs;
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi) acc[mi] = mfma16(af[(mi * KS + ks) * 64], b, acc[mi]);
                }
        }
        if (valid) {
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = (mt0 + mi) * 16 + q * 4 + r;
                    if (o < COUT) out[((size_t)t * COUT + o) * lout + p] = fmaxf(acc[mi][r], 0.0f);
                }
        }
    }
}

// ---- bf16 variants (BASELINE config 3): conv inputs and weights rounded to bf16, fp32 accumulate, on
// v_mfma_f32_16x16x32_bf16.  One k-step = 32 K slots = channels 8 q + j (q = lane >> 4, j = 0..7) of one tap, so a
// 17-channel level needs 3 k-steps instead of 15; activations stay fp32 in LDS and are rounded when the B fragment is built.
__device__ __forceinline__ f32x4 mfma16_bf16(const bf16x8& a, const bf16x8& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <int NW, int CIN, int COUT, int MT>
__device__ __forceinline__ void mconv_down_bf16(const void* __restrict__ afrag, const float* __restrict__ bt, const float* in,
                                                float* out, int lout, int tt, int wv, int lane) {
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
                acc[mi][r] = o < COUT ? bt[o * lout + p] : 0.0f;
            }
        if (CIN == 1) {
            bf16x8 b;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool ok = q == 0 && j < 3 && !(j == 0 && p == 0);
                b[j] = f2bf(ok ? in[t * lin + 2 * p - 1 + j] : 0.0f);
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
        if (valid) {
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = (mt0 + mi) * 16 + q * 4 + r;
                    if (o < COUT) out[((size_t)t * COUT + o) * lout + p] = fmaxf(acc[mi][r], 0.0f);
                }
        }
    }
}

template <int NW, bool SKIP, int MT>
__device__ __forceinline__ void mconv_up_bf16(const void* __restrict__ afrag, const float* __restrict__ bt, const float* a,
                                              const float* skip, float* out, int l, int tt, int wv, int lane) {
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
                const int o = (mt0 + mi) * 16 + q * 4 + r;
                ev[mi][r] = o < kHidden ? bt[o * lo + 2 * i] : 0.0f;
                od[mi][r] = o < kHidden ? bt[o * lo + 2 * i + 1] : 0.0f;
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
                x0[j] = f2bf(ok ? sc[0] : 0.0f);
                x1[j] = f2bf(ok && has_next ? sc[1] : 0.0f);
            }
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                const bf16x8* am = af8 + ((size_t)(mt0 + mi) * 3 * PARTS + part) * 64 + lane;    // [mt][set][part][64]
                ev[mi] = mfma16_bf16(am[0 * PARTS * 64], x0, ev[mi]);
                od[mi] = mfma16_bf16(am[1 * PARTS * 64], x0, od[mi]);
                od[mi] = mfma16_bf16(am[2 * PARTS * 64], x1, od[mi]);
            }
        }
        if (valid) {
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = (mt0 + mi) * 16 + q * 4 + r;
                    if (o < kHidden)
                        *reinterpret_cast<float2*>(out + ((size_t)t * kHidden + o) * lo + 2 * i) =
                            make_float2(fmaxf(ev[mi][r], 0.0f), fmaxf(od[mi][r], 0.0f));
                }
        }
    }
}

// ---- ConvTranspose1d k3 s2 p1 output_padding 1 (L -> 2L) on cat(a, skip) data channels (+ folded position
// bias, relu).  a [TT][17][l], skip [TT][17][l] (absent at the first level) -> out [TT][17][2l].
// Item = (t, input position i):  out[2i]   = sum_c w[c][o][1] x[c][i]
//                                out[2i+1] = sum_c w[c][o][2] x[c][i] + w[c][o][0] x[c][i+1]
// afrag: [MTILES=2][3 tap sets: k=1, k=2, k=0][KSU][64]; K order = part-major (a, then skip), channels in fours.
template <int NW, bool SKIP, int MT>
__device__ __forceinline__ void mconv_up(const float* __restrict__ afrag, const float* __restrict__ bt, const float* a,
                                         const float* skip, float* out, int l, int tt, int wv, int lane) {
    constexpr int CS = 5;                      // ceil(17 / 4)
    constexpr int KSU = SKIP ? 2 * CS : CS;
    constexpr int MTILES = 2, MG = MTILES / MT;
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
                const int o = (mt0 + mi) * 16 + q * 4 + r;
                ev[mi][r] = o < kHidden ? bt[o * lo + 2 * i] : 0.0f;
                od[mi][r] = o < kHidden ? bt[o * lo + 2 * i + 1] : 0.0f;
            }
        const float* af = afrag + (size_t)mt0 * 3 * KSU * 64 + lane;
#pragma unroll
        for (int part = 0; part < (SKIP ? 2 : 1); ++part) {
            const float* src = (part == 0 ? a : skip) + (size_t)t * kHidden * l + i;
#pragma unroll
            for (int cs = 0; cs < CS; ++cs) {
                const int c = min(4 * cs + q, kHidden - 1);
                const float x0 = src[c * l];
                const float x1 = has_next ? src[c * l + 1] : 0.0f;
                const int ks = part * CS + cs;
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    const float* am = af + (size_t)mi * 3 * KSU * 64;
                    ev[mi] = mfma16(am[(0 * KSU + ks) * 64], x0, ev[mi]);
                    od[mi] = mfma16(am[(1 * KSU + ks) * 64], x0, od[mi]);
                    od[mi] = mfma16(am[(2 * KSU + ks) * 64], x1, od[mi]);
                }
            }
        }
        if (valid) {
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = (mt0 + mi) * 16 + q * 4 + r;
                    if (o < kHidden)
                        *reinterpret_cast<float2*>(out + ((size_t)t * kHidden + o) * lo + 2 * i) =
                            make_float2(fmaxf(ev[mi][r], 0.0f), fmaxf(od[mi][r], 0.0f));
                }
        }
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// LDS plan (floats), C = compressed bins, per chunk of TT <= 3 steps.
struct CellLds {
    int x, d0, d1, d2, d3, h, gh, hi, u0, u1, u2, total;
    __host__ __device__ explicit CellLds(int C) {
        const int T = kCellChunk, F = 16 * C;
        int o = 4;                            // words 0..3: guard in front of the first buffer (index -1 reads)
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

constexpr int kCellLdsFloats = 4 + 3 * 16 * kMaxC + 3 * 17 * 14 * kMaxC * 2 + 3 * 51 * kMaxC + 17 * kMaxC + 51 * kMaxC + 3 * 17 * kMaxC;
constexpr int kCellSmem = 4 * kCellLdsFloats;      // 34,976 B

// One workgroup of NW wavefronts runs the T-step forward of stream `b`.  `smem`: kCellSmem bytes of LDS.
template <int NW, bool BF16 = false>
__device__ __forceinline__ void cell_body(char* smem, const CellDev& cd, const float* __restrict__ x,
                                          const float* __restrict__ hx_in, float* __restrict__ out,
                                          float* __restrict__ hx_out, int T, int C, size_t b, int tid,
                                          float hx_scale = 1.0f) {
    constexpr int kCellThreads = NW * 64;
    constexpr int kGateSlots = (kGates * kMaxC + kCellThreads - 1) / kCellThreads;   // gate items per thread
    float* lds = reinterpret_cast<float*>(smem);
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int F = 16 * C;
    const CellLds L(C);
    float* sx = lds + L.x;   float* sd0 = lds + L.d0; float* sd1 = lds + L.d1; float* sd2 = lds + L.d2;
    float* sd3 = lds + L.d3; float* sh = lds + L.h;   float* sgh = lds + L.gh; float* shi = lds + L.hi;
    float* su0 = lds + L.u0; float* su1 = lds + L.u1; float* su2 = lds + L.u2;

    if (tid < 4) lds[tid] = 0.0f;
    // hidden state -> LDS (gruunet2.py:294-301: zeros when the caller passes none)
    for (int i = tid; i < kHidden * C; i += kCellThreads) sh[i] = hx_in != nullptr ? hx_in[b * kHidden * C + i] : 0.0f;

    // bottleneck mapping: thread <-> (gate channel o, position p) items tid, tid + threads, ..; the 51 recurrent
    // weights of each owned item stay in VGPRs across time steps
    const int gate_items = kGates * C;
    int g_p[kGateSlots];
    bool g_on[kGateSlots];
    float wgh[kGateSlots][kHidden * 3], bgh[kGateSlots];
#pragma unroll
    for (int sl = 0; sl < kGateSlots; ++sl) {
        const int gi = tid + sl * kCellThreads;
        const int g_o = gi / C;
        g_p[sl] = gi - g_o * C;
        g_on[sl] = gi < gate_items;
#pragma unroll
        for (int i = 0; i < kHidden * 3; ++i) wgh[sl][i] = g_on[sl] ? cd.w_gh[i * kGates + g_o] : 0.0f;
        bgh[sl] = g_on[sl] ? cd.bt_gh[g_o * C + g_p[sl]] : 0.0f;
    }

    for (int t0 = 0; t0 < T; t0 += kCellChunk) {
        const int tt = min(kCellChunk, T - t0);
        __syncthreads();
        for (int i = tid; i < tt * F; i += kCellThreads) sx[i] = x[(b * T + t0) * F + i];
        __syncthreads();
        // ---- encoder, batched over the chunk (gruunet2.py:136-144)
        if (BF16) {
            mconv_down_bf16<NW, 1, kHidden, 2>(cd.wb_down[0], cd.bt_down[0], sx, sd0, 8 * C, tt, wv, lane);
            __syncthreads();
            mconv_down_bf16<NW, kHidden, kHidden, 2>(cd.wb_down[1], cd.bt_down[1], sd0, sd1, 4 * C, tt, wv, lane);
            __syncthreads();
            mconv_down_bf16<NW, kHidden, kHidden, 1>(cd.wb_down[2], cd.bt_down[2], sd1, sd2, 2 * C, tt, wv, lane);
            __syncthreads();
            mconv_down_bf16<NW, kHidden, kGates, 1>(cd.wb_down[3], cd.bt_down[3], sd2, sd3, C, tt, wv, lane);
        } else {
            mconv_down<NW, 1, kHidden, 2>(cd.w_down[0], cd.bt_down[0], sx, sd0, 8 * C, tt, wv, lane);
            __syncthreads();
            mconv_down<NW, kHidden, kHidden, 2>(cd.w_down[1], cd.bt_down[1], sd0, sd1, 4 * C, tt, wv, lane);
            __syncthreads();
            mconv_down<NW, kHidden, kHidden, 1>(cd.w_down[2], cd.bt_down[2], sd1, sd2, 2 * C, tt, wv, lane);
            __syncthreads();
            mconv_down<NW, kHidden, kGates, 1>(cd.w_down[3], cd.bt_down[3], sd2, sd3, C, tt, wv, lane);
        }
        __syncthreads();
        // ---- recurrent part, sequential in t (gruunet2.py:232-240)
        for (int t = 0; t < tt; ++t) {
#pragma unroll
            for (int sl = 0; sl < kGateSlots; ++sl) {
                if (g_on[sl]) {            // gh = relu(conv k3 s1 p1 (hx) + position bias)
                    float acc = bgh[sl];
#pragma unroll
                    for (int c = 0; c < kHidden; ++c) {
                        const float* hc = sh + c * C + g_p[sl];
                        const float x0 = g_p[sl] > 0 ? hc[-1] : 0.0f;
                        const float x1 = hc[0];
                        const float x2 = g_p[sl] + 1 < C ? hc[1] : 0.0f;
                        acc = fmaf(wgh[sl][c * 3 + 0], x0, acc);
                        acc = fmaf(wgh[sl][c * 3 + 1], x1, acc);
                        acc = fmaf(wgh[sl][c * 3 + 2], x2, acc);
                    }
                    sgh[tid + sl * kCellThreads] = fmaxf(acc, 0.0f);
                }
            }
            __syncthreads();
            if (tid < kHidden * C) {   // chunk order r, i, n (gruunet2.py:234-240)   (17 C <= 85 < threads)
                const float* gx = sd3 + (size_t)t * kGates * C;
                const float r = sigmoidf_(gx[tid] + sgh[tid]);
                const float z = sigmoidf_(gx[kHidden * C + tid] + sgh[kHidden * C + tid]);
                const float n = tanhf(gx[2 * kHidden * C + tid] + r * sgh[2 * kHidden * C + tid]);
                const float hn = n + z * (sh[tid] - n);
                sh[tid] = hn;
                shi[t * kHidden * C + tid] = hn;
            }
            __syncthreads();
        }
        // ---- decoder, batched over the chunk (gruunet2.py:184-199); skips are d2, d1, d0 (the last level has no cat)
        if (BF16) {
            mconv_up_bf16<NW, false, 1>(cd.wb_up[0], cd.bt_up[0], shi, nullptr, su0, C, tt, wv, lane);
            __syncthreads();
            mconv_up_bf16<NW, true, 1>(cd.wb_up[1], cd.bt_up[1], su0, sd2, su1, 2 * C, tt, wv, lane);
            __syncthreads();
            mconv_up_bf16<NW, true, 2>(cd.wb_up[2], cd.bt_up[2], su1, sd1, su2, 4 * C, tt, wv, lane);
        } else {
            mconv_up<NW, false, 1>(cd.w_up[0], cd.bt_up[0], shi, nullptr, su0, C, tt, wv, lane);
            __syncthreads();
            mconv_up<NW, true, 1>(cd.w_up[1], cd.bt_up[1], su0, sd2, su1, 2 * C, tt, wv, lane);
            __syncthreads();
            mconv_up<NW, true, 2>(cd.w_up[2], cd.bt_up[2], su1, sd1, su2, 4 * C, tt, wv, lane);
        }
        __syncthreads();
        // last level: one output channel; lane = (t, input position), weights through the scalar cache
        {
            const int l = 8 * C, items = tt * l;
            cfloat_ptr wgt = (cfloat_ptr)cd.w_up[3];
            for (int it = tid; it < items; it += kCellThreads) {
                const int t = it / l, i = it - t * l;
                float ev = cd.bt_up[3][2 * i], od = cd.bt_up[3][2 * i + 1];
                const bool has_next = i + 1 < l;
#pragma unroll 2
                for (int c = 0; c < 2 * kHidden; ++c) {
                    const float* src = c < kHidden ? su2 + ((size_t)t * kHidden + c) * l
                                                   : sd0 + ((size_t)t * kHidden + (c - kHidden)) * l;
                    const float x0 = src[i];
                    const float x1 = has_next ? src[i + 1] : 0.0f;
                    ev = fmaf(wgt[c * 3 + 1], x0, ev);
                    od = fmaf(wgt[c * 3 + 2], x0, od);
                    od = fmaf(wgt[c * 3 + 0], x1, od);
                }
                float* dst = out + (b * T + t0 + t) * F + 2 * i;
                *reinterpret_cast<float2*>(dst) = make_float2(ev, od);
            }
        }
    }
    __syncthreads();
    // (hx_scale != 1: the server variant's `hx = hx * 0.9`, server.py:214)
    for (int i = tid; i < kHidden * C; i += kCellThreads) hx_out[b * kHidden * C + i] = sh[i] * hx_scale;
}

}  // namespace dn
