// dn_stft_body.hpp -- device body of the analysis stage (see dn_stft.hip for the description).
#pragma once
#include "dn_internal.hpp"
#include "dn_wavefft.hpp"

namespace dn {

constexpr int kStftThreads = 192;


template <int NFFT> constexpr int stft_smem() { return 4 * NFFT + 8 * 3 * Geo<NFFT>::kTile + 4 * 3 * (Geo<NFFT>::kBins + 7) + 16 + 4 * 3 * 128; }

#ifdef DN_PROBE
static __device__ unsigned long long g_stft_probe[8];   // diagnostic build: phases of wave 0 of the workgroup of stream 0 (tools/hop_wg_probe.py)
#define DN_SSTAMP(id) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; \
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); if (b == 0 && tid == 0) g_stft_probe[id] = t_; \
    __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DN_SSTAMP(id) do { } while (0)
#endif

// One workgroup (192 threads = 3 wavefronts = 3 STFT columns) processes frame `b`.  `smem`: stft_smem<NFFT>() bytes of LDS.
// THREADS > 192: the extra wavefront helps load / normalise the frame and then waits at the end (the three columns are three waves).
template <int NFFT, bool WRITE_SPEC, bool WRITE_MEL, int THREADS = kStftThreads>
__device__ __forceinline__ void stft_body(char* smem, const DspDev& d, const float* __restrict__ frames,
                                          float2* __restrict__ spec, float* __restrict__ mel,
                                          float* __restrict__ peak_out, uint32_t flags, size_t b, int tid) {
    using G = Geo<NFFT>;
    constexpr int kNR = G::kNR, kHop = G::kHop, kBins = G::kBins, kNV = G::kNV, kNP = G::kNP, kFftTile = G::kTile;
    float* xs = reinterpret_cast<float*>(smem);
    v2f (*tile)[kFftTile] = reinterpret_cast<v2f (*)[kFftTile]>(smem + 4 * kNR);
    float (*magbuf)[kBins + 7] = reinterpret_cast<float (*)[kBins + 7]>(smem + 4 * kNR + 8 * 3 * kFftTile);
    float* red = reinterpret_cast<float*>(smem + 4 * kNR + 8 * 3 * kFftTile + 4 * 3 * (kBins + 7));

    float (*melsum)[128] = reinterpret_cast<float (*)[128]>(red + 4);

    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

    DN_SSTAMP(0);
    // ---- P1: load the frame once, find max|x|
    const float4* f4 = reinterpret_cast<const float4*>(frames + b * kNR);
    static_assert(kNR / 4 <= 2 * THREADS && THREADS <= 256, "two float4 per thread cover the frame");
    const bool one = tid < kNR / 4;
    float4 q0 = one ? f4[tid] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 q1 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool two = tid < (kNR / 4 - THREADS);
    if (two) q1 = f4[tid + THREADS];
#ifndef DN_STFT_PREFETCH
#define DN_STFT_PREFETCH 0
#endif
    // (experiment knob DN_STFT_PREFETCH: request window and twiddles here, in front of the first barrier, and let both barriers order LDS only)
    const float4* w4 = reinterpret_cast<const float4*>(d.window);
    const bool prewin = (flags & DN_PRE_WINDOW) != 0;
    float4 wq0 = make_float4(1.f, 1.f, 1.f, 1.f), wq1 = wq0;
    typename G::Fft::Tw tw;
    v2f wkh[kNP], wwin[kNV];
    if (DN_STFT_PREFETCH) {
        wq0 = prewin && one ? w4[tid] : wq0;
        wq1 = prewin && two ? w4[tid + THREADS] : wq1;
        if (w < 3) {
            G::Fft::load(tw, reinterpret_cast<const v2f*>(d.twc), lane);
#pragma unroll
            for (int t = 0; t < kNP; ++t) wkh[t] = cscale(reinterpret_cast<const v2f*>(d.twr)[lane + 64 * t], 0.5f);
#pragma unroll
            for (int t = 0; t < kNV; ++t) wwin[t] = reinterpret_cast<const v2f*>(d.window)[lane + 64 * t];
        }
    }
    float mx = fmaxf(fmaxf(fabsf(q0.x), fabsf(q0.y)), fmaxf(fabsf(q0.z), fabsf(q0.w)));
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(q1.x), fabsf(q1.y)), fmaxf(fabsf(q1.z), fabsf(q1.w))));
    mx = wave_max(mx);
    if (lane == 0) red[w] = mx;
    if (DN_STFT_PREFETCH) DN_LDS_BARRIER(); else __syncthreads();
    DN_SSTAMP(1);
    float pk = fmaxf(red[0], fmaxf(red[1], red[2]));
    if (THREADS > 192) pk = fmaxf(pk, red[3]);
    const bool norm = (flags & DN_PEAK_NORMALIZE) && pk > 1e-6f;     // app3.py:182
    if (!norm) pk = 1.0f;                                             // app3.py:186
    if (peak_out != nullptr && tid == 0) peak_out[b] = pk;

    // ---- P1/P2: x / peak, optional first Hann multiply (app3.py:183,188)
    const float ipk = __builtin_amdgcn_rcpf(pk);
    auto prep = [&](float4 q, float4 ww, int i4) {
        if (norm) { q.x *= ipk; q.y *= ipk; q.z *= ipk; q.w *= ipk; }
        if (prewin) {
            if (!DN_STFT_PREFETCH) ww = w4[i4];
            q.x *= ww.x; q.y *= ww.y; q.z *= ww.z; q.w *= ww.w;
        }
        reinterpret_cast<float4*>(xs)[i4] = q;
    };
    if (one) prep(q0, wq0, tid);
    if (two) prep(q1, wq1, tid + THREADS);
    if (DN_STFT_PREFETCH) DN_LDS_BARRIER(); else __syncthreads();
    if (THREADS > 192 && w >= 3) return;          // (no workgroup barrier below this point)
    DN_SSTAMP(2);

    // ---- P4: column w of the centred STFT: padded position p = hop w + n, source i = p - hop reflected
    v2f v[kNV];
    if (!DN_STFT_PREFETCH) {
        G::Fft::load(tw, reinterpret_cast<const v2f*>(d.twc), lane);
#pragma unroll
        for (int t = 0; t < kNP; ++t) wkh[t] = cscale(reinterpret_cast<const v2f*>(d.twr)[lane + 64 * t], 0.5f);
    }
#pragma unroll
    for (int t = 0; t < kNV; ++t) {
        const int m = lane + 64 * t;
        const int n0 = 2 * m;
        int i0 = w * kHop + n0 - kHop, i1 = i0 + 1;
        i0 = i0 < 0 ? -i0 : (i0 >= kNR ? 2 * kNR - 2 - i0 : i0);
        i1 = i1 < 0 ? -i1 : (i1 >= kNR ? 2 * kNR - 2 - i1 : i1);
        const v2f ww = DN_STFT_PREFETCH ? wwin[t] : reinterpret_cast<const v2f*>(d.window)[m];
        v[t] = mk2(xs[i0] * ww[0], xs[i1] * ww[1]);
    }
    DN_SSTAMP(3);
    G::Fft::template run<false>(v, tw, tile[w], lane);
    // Hermitian split in pair order: this lane gets bins k = lane + 64 t and NC - k (t < NP); lane 0 also bin NC/2
    v2f lo[kNP], hi[kNP], mid;
    rfft_split_pairs<kNV>(v, wkh, lane, lo, hi, mid);
    DN_SSTAMP(4);

    if (WRITE_SPEC) {
        v2f* srow = reinterpret_cast<v2f*>(spec) + (b * 3 + w) * kBins;
#pragma unroll
        for (int t = 0; t < kNP; ++t) {
            srow[lane + 64 * t] = lo[t];
            srow[G::kNC - (lane + 64 * t)] = hi[t];
        }
        if (lane == 0) srow[G::kNC / 2] = mid;
    }
    if (WRITE_MEL) {
        // ---- P5: magnitude -> banded mel filterbank -> log1p; P6: rows are already (B,3,M)
#pragma unroll
        for (int t = 0; t < kNP; ++t) {
            magbuf[w][lane + 64 * t] = fast_abs2(lo[t][0], lo[t][1]);
            magbuf[w][G::kNC - (lane + 64 * t)] = fast_abs2(hi[t][0], hi[t][1]);
        }
        if (lane == 0) magbuf[w][G::kNC / 2] = fast_abs2(mid[0], mid[1]);
        wave_sync();
        DN_SSTAMP(5);
        float* mrow = mel + (b * 3 + w) * d.n_mels;
        const int qsteps = d.mel_qsteps;
        if (qsteps > 0) {
            // packed schedule (DspDev::mel_q): four lanes share a filter and split its taps, 16 filters a group; a step is one coalesced
            // 8-byte load (weight, bin), one LDS read and one FMA -- 19 steps against 56 taps of ~6 instructions with a lane per filter
            // (80 HTK mels at 16 kHz).  This stage is issue bound: it shares its SIMDs with a Griffin-Lim chain that wins arbitration.
            // The schedule is padded with (weight 0, bin 0) steps to its full length by the plan, so every load below is unconditional; the magnitudes of
            // sixteen steps are requested from LDS in one go, ahead of their FMAs (fetched step by step between the uniform branches of the group ends,
            // every step paid its own LDS round trip: 6.8 k of the analysis stage's 19 k ticks).  Same products, same order of accumulation.
            constexpr int kMelQSteps = mel_q_steps(NFFT), kChunk = 16;
            static_assert(kMelQSteps % kChunk == 0, "whole chunks");
            float2 q[kMelQSteps];
#pragma unroll
            for (int t = 0; t < kMelQSteps; ++t) q[t] = d.mel_q[t * 64 + lane];
            float acc = 0.0f;
            int mbase = lane >> 2;
#pragma unroll
            for (int c0 = 0; c0 < kMelQSteps; c0 += kChunk) {
                if (c0 < qsteps) {                                                 // wave-uniform: whole chunks past the schedule do nothing
                    float mg[kChunk];
#pragma unroll
                    for (int i = 0; i < kChunk; ++i) mg[i] = magbuf[w][__builtin_bit_cast(int, q[c0 + i].y)];
#pragma unroll
                    for (int i = 0; i < kChunk; ++i) {
                        const int t = c0 + i;
                        acc = fmaf(q[t].x, mg[i], acc);                            // (a padded step adds 0 * |X[0]|)
                        if ((d.mel_qlast >> t) & 1ull) {                             // the group's last step: fold the four lanes of a filter
                            acc = quad_sum(acc);
                            if ((lane & 3) == 0) melsum[w][mbase] = acc;
                            mbase += 16;
                            acc = 0.0f;
                        }
                    }
                }
            }
            wave_sync();
            for (int m = lane; m < d.n_mels; m += 64) mrow[m] = fast_log1p(melsum[w][m]);
        } else {
            for (int m = lane; m < d.n_mels; m += 64) {
                const int s = d.mel_start[m], len = d.mel_len[m];
                float acc = 0.0f;
                for (int i = 0; i < len; ++i) acc = fmaf(d.mel_w[i * d.n_mels + m], magbuf[w][s + i], acc);
                mrow[m] = fast_log1p(acc);
            }
        }
        DN_SSTAMP(6);
    }
}

}  // namespace dn
