// dn_invmel_body.hpp -- device body of the inverse-mel stage (see dn_invmel.hip for the description).
#pragma once
#include "dn_internal.hpp"
#include "dn_wavefft.hpp"

namespace dn {

constexpr int kInvThreads = 192;
constexpr int kInvRows = 3;
constexpr int kMaxMels = 128;

constexpr int kInvTaps = 2 * kInvBand + 1, kInvThirds = 3, kInvTapsThird = kInvTaps / kInvThirds;     // 33 diagonals, 11 a third
static_assert(kInvTapsThird * kInvThirds == kInvTaps, "the band splits into three equal parts");
// LDS of the factored form, in float4 (x, y, z = the hop's three columns): mel with kInvBand zero rows on either side, the three
// partial sums of y = G^-1 mel, y
constexpr int kInvMel4 = kMaxMels + 2 * kInvBand, kInvLds4 = kInvMel4 + (kInvThirds + 1) * kMaxMels;
constexpr int kInvSmem = 16 * kInvLds4;

#ifndef DN_INVMEL_UNROLL
#define DN_INVMEL_UNROLL 8
#endif

// ---- the factored inverse mel (DspDev::ginv_band / fb2): lin = relu(fb (G^-1 mel)), G = fb^T fb, G^-1 banded.
// y = G^-1 mel is summed in ONE canonical order whatever the number of threads, so that the pipelined hop (256 threads) and the
// unpipelined one (its Griffin-Lim prologue, 192 threads) agree to the bit: item (third t, filter a) folds diagonals 11 t .. 11 t + 10
// of row a in ascending order, and y[a] = (p0 + p1) + p2.  An item's eleven weights are requested in one go (fetch) and can be in
// flight while the caller does something else before fold.
template <int THREADS>
struct InvBandFold {
    static constexpr int kItems = (kInvThirds * kMaxMels + THREADS - 1) / THREADS;     // items a thread may own (2)
    float g[kItems][kInvTapsThird];
    __device__ __forceinline__ void fetch(const DspDev& d, int tid) {
        const int M = d.n_mels;
#pragma unroll
        for (int j = 0; j < kItems; ++j) {
            const int it = tid + THREADS * j;
            const bool on = it < kInvThirds * M;
            const int t = on ? (it >= M) + (it >= 2 * M) : 0, a = on ? it - t * M : 0;
#pragma unroll
            for (int i = 0; i < kInvTapsThird; ++i) g[j][i] = on ? d.ginv_band[(size_t)(t * kInvTapsThird + i) * M + a] : 0.0f;
        }
    }
    // mel4: [kInvMel4] with the frame's mel magnitudes at rows kInvBand .. kInvBand + M and zeros around; yp4: [3][kMaxMels];
    // y4: [kMaxMels] (LDS).  Two workgroup barriers inside; y4 is complete on return.
    __device__ __forceinline__ void fold(const DspDev& d, const float4* mel4, float4* yp4, float4* y4, int tid) const {
        const int M = d.n_mels;
#pragma unroll
        for (int j = 0; j < kItems; ++j) {
            const int it = tid + THREADS * j;
            if (it < kInvThirds * M) {
                const int t = (it >= M) + (it >= 2 * M), a = it - t * M;
                const float4* mrow = mel4 + a + t * kInvTapsThird;          // row (a - kInvBand + 11 t + i) + kInvBand
                float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
#pragma unroll
                for (int i = 0; i < kInvTapsThird; ++i) {
                    const float4 mv = mrow[i];
                    a0 = fmaf(g[j][i], mv.x, a0); a1 = fmaf(g[j][i], mv.y, a1); a2 = fmaf(g[j][i], mv.z, a2);
                }
                yp4[t * kMaxMels + a] = make_float4(a0, a1, a2, 0.0f);
            }
        }
        __syncthreads();
        for (int a = tid; a < M; a += THREADS) {
            const float4 p0 = yp4[a], p1 = yp4[kMaxMels + a], p2 = yp4[2 * kMaxMels + a];
            y4[a] = make_float4((p0.x + p1.x) + p2.x, (p0.y + p1.y) + p2.y, (p0.z + p1.z) + p2.z, 0.0f);
        }
        __syncthreads();
    }
};
// one bin, the three columns: the (at most) two filters the bin belongs to
__device__ __forceinline__ float4 invmel_bin(const float4& f, const float4* y4) {
    const float4 u = y4[__builtin_bit_cast(int, f.z)], v = y4[__builtin_bit_cast(int, f.w)];
    return make_float4(fmaxf(fmaf(f.y, v.x, f.x * u.x), 0.0f), fmaxf(fmaf(f.y, v.y, f.x * u.y), 0.0f), fmaxf(fmaf(f.y, v.z, f.x * u.z), 0.0f), 0.0f);
}

static_assert(16 % (2 * DN_INVMEL_UNROLL) == 0, "the dense loop's round divides the 16 filters the plan pads to");

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
    // ---- factored form (DspDev::ginv_band): everything that does not depend on the frame is requested first, in front of the residual arithmetic
    const bool factored = d.ginv_band != nullptr;                       // (uniform)
    constexpr int kRounds = (kBins + THREADS - 1) / THREADS;
    InvBandFold<THREADS> inv;
    float4 f2[kRounds];
    float4* mel4 = reinterpret_cast<float4*>(smem);
    if (factored) {
        inv.fetch(d, tid);
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            const int k = tid + THREADS * r;
            f2[r] = k < kBins ? d.fb2[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int i = tid; i < kInvMel4; i += THREADS) mel4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
    }
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
        if (factored) reinterpret_cast<float*>(mel4 + kInvBand + m)[r] = v;
        else mm[r][m] = v;
    }
    if (!factored) {          // the dense loop walks 16 filters a round: zero magnitudes for the rows the plan padded (n_mels % 16 != 0)
        const int Mp = (M + 15) & ~15;
        for (int i = tid; i < kInvRows * (Mp - M); i += THREADS) mm[i / (Mp - M)][M + i % (Mp - M)] = 0.0f;
    }
    __syncthreads();
    DN_ISTAMP(1);
    if (factored) {
        float4* yp4 = mel4 + kInvMel4;
        float4* y4 = yp4 + kInvThirds * kMaxMels;
        inv.fold(d, mel4, yp4, y4, tid);
        DN_ISTAMP(2);
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            const int k = tid + THREADS * r;
            if (k < kBins) {
                const float4 o = invmel_bin(f2[r], y4);
                if (r0 + 0 < (size_t)rows) lin[(r0 + 0) * kBins + k] = o.x;
                if (r0 + 1 < (size_t)rows) lin[(r0 + 1) * kBins + k] = o.y;
                if (r0 + 2 < (size_t)rows) lin[(r0 + 2) * kBins + k] = o.z;
            }
        }
        DN_ISTAMP(3);
        DN_ISTAMP(4);
        DN_ISTAMP(5);
        return;
    }
    // ---- dense form.  thread <-> bins tid, tid+THREADS, ..: independent load streams over the transposed pseudo-inverse (rows zero padded, so every load
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
    // (the plan pads pinv_t with zero rows to a multiple of 16 = 2 kUB filters; mm is zero there)
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
