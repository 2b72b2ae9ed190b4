// dn_invmel_body.hpp -- device body of the inverse-mel stage (see dn_invmel.hip for the description).
#pragma once
#include "dn_internal.hpp"
#include "dn_wavefft.hpp"

namespace dn {

constexpr int kInvThreads = 192;
constexpr int kInvRows = 3;
constexpr int kMaxMels = 128;

constexpr int kInvSmem = 4 * kInvRows * kMaxMels;

#ifndef DN_INVMEL_UNROLL
#define DN_INVMEL_UNROLL 8
#endif
#define DN_PRAGMA_(x) _Pragma(#x)
#define DN_PRAGMA(x) DN_PRAGMA_(x)

// One workgroup (192 threads) handles rows r0 .. r0+2 (the three columns of one stream).  `smem`: kInvSmem bytes.
template <int NFFT, bool RESIDUAL, int THREADS = kInvThreads>
__device__ __forceinline__ void invmel_body(char* smem, const DspDev& d, const float* __restrict__ x,
                                            const float* __restrict__ diff, float* __restrict__ lin, int rows,
                                            size_t r0, int tid) {
    constexpr int kBins = Geo<NFFT>::kBins;
    float (*mm)[kMaxMels] = reinterpret_cast<float (*)[kMaxMels]>(smem);
    const int M = d.n_mels;
    for (int i = tid; i < kInvRows * M; i += THREADS) {
        const int r = i / M, m = i - r * M;
        float v = 0.0f;
        if (r0 + r < (size_t)rows) {
            v = x[(r0 + r) * M + m];
            if (RESIDUAL) {
                v = v - diff[(r0 + r) * M + m];
                v = v >= 0.0f ? v : 0.2f * v;          // leaky_relu, app3.py:204
                v = fmaxf(expm1f(v), 0.0f);            // app3.py:207-208
            }
        }
        mm[r][m] = v;
    }
    __syncthreads();
    // thread <-> bins tid, tid+THREADS, ..: independent load streams over the transposed pseudo-inverse (rows zero padded, so every load
    // of a full round is in bounds).  n_fft/2 + 1 bins = whole rounds + a few left-over bins (ONE for 256 threads): those get a loop of
    // their own that only the wavefront owning them runs, instead of a padded round in which every thread loads zeros.
    constexpr int kFull = kBins / THREADS, kRem = kBins - kFull * THREADS;
    float acc[kFull][3];
#pragma unroll
    for (int r = 0; r < kFull; ++r) acc[r][0] = acc[r][1] = acc[r][2] = 0.0f;
    const float* p = d.pinv_t + tid;
DN_PRAGMA(unroll DN_INVMEL_UNROLL)
    for (int m = 0; m < M; ++m) {
        const float* pm = p + (size_t)m * d.pinv_stride;
        const float m0 = mm[0][m], m1 = mm[1][m], m2 = mm[2][m];
#pragma unroll
        for (int r = 0; r < kFull; ++r) {
            const float pv = pm[THREADS * r];
            acc[r][0] = fmaf(pv, m0, acc[r][0]); acc[r][1] = fmaf(pv, m1, acc[r][1]); acc[r][2] = fmaf(pv, m2, acc[r][2]);
        }
    }
#pragma unroll
    for (int r = 0; r < kFull; ++r) {
        const int k = tid + THREADS * r;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            if (r0 + c < (size_t)rows) lin[(r0 + c) * kBins + k] = fmaxf(acc[r][c], 0.0f);
    }
    if (kRem > 0 && tid < kRem) {
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
        const float* pr = p + THREADS * kFull;
#pragma unroll 8
        for (int m = 0; m < M; ++m) {
            const float pv = pr[(size_t)m * d.pinv_stride];
            a0 = fmaf(pv, mm[0][m], a0); a1 = fmaf(pv, mm[1][m], a1); a2 = fmaf(pv, mm[2][m], a2);
        }
        const int k = tid + THREADS * kFull;
        if (r0 + 0 < (size_t)rows) lin[(r0 + 0) * kBins + k] = fmaxf(a0, 0.0f);
        if (r0 + 1 < (size_t)rows) lin[(r0 + 1) * kBins + k] = fmaxf(a1, 0.0f);
        if (r0 + 2 < (size_t)rows) lin[(r0 + 2) * kBins + k] = fmaxf(a2, 0.0f);
    }
}

}  // namespace dn
