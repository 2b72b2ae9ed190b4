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
static_assert(16 % (2 * DN_INVMEL_UNROLL) == 0, "n_mels is a multiple of 16");

#ifdef DN_PROBE
static __device__ unsigned long long g_inv_probe[8];   // diagnostic build: phases of workgroup r0 == 0 (tools/hop_wg_probe.py)
#define DN_ISTAMP(id) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); if (r0 == 0 && tid == 0) g_inv_probe[id] = t_; \
    __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DN_ISTAMP(id) do { } while (0)
#endif

// One workgroup (192 threads) handles rows r0 .. r0+2 (the three columns of one stream).  `smem`: kInvSmem bytes.
template <int NFFT, bool RESIDUAL, int THREADS = kInvThreads>
__device__ __forceinline__ void invmel_body(char* smem, const DspDev& d, const float* __restrict__ x,
                                            const float* __restrict__ diff, float* __restrict__ lin, int rows,
                                            size_t r0, int tid) {
    constexpr int kBins = Geo<NFFT>::kBins;
    float (*mm)[kMaxMels] = reinterpret_cast<float (*)[kMaxMels]>(smem);
    const int M = d.n_mels;
    DN_ISTAMP(0);
    for (int i = tid; i < kInvRows * M; i += THREADS) {
        const int r = i / M, m = i - r * M;
        float v = 0.0f;
        if (r0 + r < (size_t)rows) {
            v = x[(r0 + r) * M + m];
            if (RESIDUAL) {
                v = v - diff[(r0 + r) * M + m];
                v = v >= 0.0f ? v : 0.2f * v;          // leaky_relu, app3.py:204
                v = fmaxf(fast_expm1(v), 0.0f);        // app3.py:207-208
            }
        }
        mm[r][m] = v;
    }
    __syncthreads();
    DN_ISTAMP(1);
    // thread <-> bins tid, tid+THREADS, ..: independent load streams over the transposed pseudo-inverse (rows zero padded, so every load
    // of a full round is in bounds).  n_fft/2 + 1 bins = whole rounds + a few left-over bins (ONE for 256 threads): those get a loop of
    // their own that only the wavefront owning them runs, instead of a padded round in which every thread loads zeros.
    constexpr int kFull = kBins / THREADS, kRem = kBins - kFull * THREADS;
    float acc[kFull][3];
#pragma unroll
    for (int r = 0; r < kFull; ++r) acc[r][0] = acc[r][1] = acc[r][2] = 0.0f;
    const float* p = d.pinv_t + tid;
    // The matrix comes from L2 (every workgroup reads all of it) at ~1.5 k cycles a round trip: batches of kUB mel rows, the next
    // batch's loads issued BEFORE the current batch's FMAs (two register sets), so one round trip is exposed instead of M / kUB.
    constexpr int kUB = DN_INVMEL_UNROLL;
    float pv[2][kUB][kFull];
    auto fetch = [&](float (&buf)[kUB][kFull], int m0) {
#pragma unroll
        for (int u = 0; u < kUB; ++u)
#pragma unroll
            for (int r = 0; r < kFull; ++r) buf[u][r] = p[(size_t)(m0 + u) * d.pinv_stride + THREADS * r];
    };
    auto accumulate = [&](const float (&buf)[kUB][kFull], int m0) {
#pragma unroll
        for (int u = 0; u < kUB; ++u) {
            const float m0v = mm[0][m0 + u], m1v = mm[1][m0 + u], m2v = mm[2][m0 + u];
#pragma unroll
            for (int r = 0; r < kFull; ++r) {
                acc[r][0] = fmaf(buf[u][r], m0v, acc[r][0]); acc[r][1] = fmaf(buf[u][r], m1v, acc[r][1]); acc[r][2] = fmaf(buf[u][r], m2v, acc[r][2]);
            }
        }
    };
    // (M is a multiple of 16 = 2 kUB: dn_dsp_create)
    fetch(pv[0], 0);
    DN_ISTAMP(2);
    for (int m0 = 0; m0 < M; m0 += 2 * kUB) {
        fetch(pv[1], m0 + kUB);
        accumulate(pv[0], m0);
        if (m0 + 2 * kUB < M) fetch(pv[0], m0 + 2 * kUB);
        accumulate(pv[1], m0 + kUB);
    }
    DN_ISTAMP(3);
#pragma unroll
    for (int r = 0; r < kFull; ++r) {
        const int k = tid + THREADS * r;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            if (r0 + c < (size_t)rows) lin[(r0 + c) * kBins + k] = fmaxf(acc[r][c], 0.0f);
    }
    DN_ISTAMP(4);
    if (kRem > 0 && kRem <= 4) {
        // A few left-over bins (one at 256 threads): a thread of its own per bin walked the M rows alone -- M / 8 serial round trips to
        // L2 while the rest of the workgroup waited.  Here a wavefront takes a bin: its lanes split the M rows (one round trip) and the
        // three partial sums are folded across the wave.
        const int wv = tid >> 6, lane = tid & 63;
        for (int j = wv; j < kRem; j += THREADS / 64) {
            const int k = THREADS * kFull + j;
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
            for (int m = lane; m < M; m += 64) {
                const float pm = d.pinv_t[(size_t)m * d.pinv_stride + k];
                a0 = fmaf(pm, mm[0][m], a0); a1 = fmaf(pm, mm[1][m], a1); a2 = fmaf(pm, mm[2][m], a2);
            }
            a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
            if (lane == 0) {
                if (r0 + 0 < (size_t)rows) lin[(r0 + 0) * kBins + k] = fmaxf(a0, 0.0f);
                if (r0 + 1 < (size_t)rows) lin[(r0 + 1) * kBins + k] = fmaxf(a1, 0.0f);
                if (r0 + 2 < (size_t)rows) lin[(r0 + 2) * kBins + k] = fmaxf(a2, 0.0f);
            }
        }
    } else if (kRem > 0 && tid < kRem) {
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
        const float* pr = p + THREADS * kFull;
#pragma unroll 8
        for (int m = 0; m < M; ++m) {
            const float pm = pr[(size_t)m * d.pinv_stride];
            a0 = fmaf(pm, mm[0][m], a0); a1 = fmaf(pm, mm[1][m], a1); a2 = fmaf(pm, mm[2][m], a2);
        }
        const int k = tid + THREADS * kFull;
        if (r0 + 0 < (size_t)rows) lin[(r0 + 0) * kBins + k] = fmaxf(a0, 0.0f);
        if (r0 + 1 < (size_t)rows) lin[(r0 + 1) * kBins + k] = fmaxf(a1, 0.0f);
        if (r0 + 2 < (size_t)rows) lin[(r0 + 2) * kBins + k] = fmaxf(a2, 0.0f);
    }
    DN_ISTAMP(5);
}

}  // namespace dn
