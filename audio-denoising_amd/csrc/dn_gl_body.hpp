// dn_gl_body.hpp -- device body of the Griffin-Lim stage (see dn_griffinlim.hip for the description).
#pragma once
#include <type_traits>

#include "dn_internal.hpp"
#include "dn_wavefft.hpp"
#include "dn_invmel_body.hpp"

namespace dn {

constexpr int kGlThreads = 192;

// Diagnostic build only (make probe: -DDN_PROBE): s_memtime stamps of one Griffin-Lim iteration of workgroup 0, stored to a
// buffer of their own that nothing else reads (profiles/ holds the shares they gave; the product build has no stamp).
#ifdef DN_PROBE
static __device__ unsigned long long g_gl_probe[3][16];   // one copy per translation unit
#define DN_STAMP(id)                                                                              \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        unsigned long long t_;                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
        if (b == 0 && it == 10 && lane == 0) g_gl_probe[W][id] = t_;                              \
        __builtin_amdgcn_sched_barrier(0);                                                        \
    } while (0)
#else
#define DN_STAMP(id) do { } while (0)
#endif

// Philox4x32-10 counter-based generator (Salmon et al. 2011).
__device__ __forceinline__ void philox4x32(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// real, imag ~ U[0,1) independently (torch.rand(dtype=complex64) semantics), keyed by
// (seed, global stream id, column, bin) so sharding streams over GPUs does not change results.
// One Philox block serves the bin PAIR (m, NC - m), m = 0..NC/2: words 0, 1 -> bin m, words 2, 3 -> bin NC - m (counter word 0 = m; the
// self-paired bin NC/2 takes words 0, 1).  Every user below already owns its bins in such pairs, so a frame costs 3 (NC/2 + 1) blocks, not
// 3 (NC + 1): the ten rounds of 32x32->64 multiplies are the whole cost of a draw (oracle/philox_ref.py restates the same numbering).
__device__ __forceinline__ float rand_unit(uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }
__device__ __forceinline__ void rand_angle_pair(uint64_t seed, uint64_t sid, int col, int m, v2f& lo, v2f& hi) {
    uint32_t c[4] = {(uint32_t)m, (uint32_t)col, (uint32_t)sid, (uint32_t)(sid >> 32)};
    philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    lo = mk2(rand_unit(c[0]), rand_unit(c[1]));
    hi = mk2(rand_unit(c[2]), rand_unit(c[3]));
}
__device__ __forceinline__ v2f rand_angle_mid(uint64_t seed, uint64_t sid, int col, int nc) {
    v2f lo, hi;
    rand_angle_pair(seed, sid, col, nc / 2, lo, hi);
    return lo;
}

// FROM_MEL = false: `mag` is the linear magnitude [B][3][513] (GriffinLim / istft entry points).
// FROM_MEL = true : `mag` is the model input x [B][3][M] and `diff` the model output; the prologue
//                   computes P8-P10 in place -- leaky_relu(x - diff, 0.2), expm1, clamp, the pinv(fb^T)
//                   contraction and relu (app3.py:203-211) -- into LDS, so the linear magnitudes never
//                   touch HBM and the separate inverse-mel launch disappears from the fused hop.
// n_fft 1536 ("lean"): the per-lane window constants (synthesis window / NC: one table; analysis window x 1/envelope: one per
// column) and the radix-12 twiddles live in LDS tables instead of 70 registers per lane, read back where they are used.  With
// them in registers the body needs ~330 live values; capped at the 256 that let a Griffin-Lim and a front workgroup share a CU
// it spilled ~75 of them to scratch in every iteration (188 us per batch-256 hop, round 1).
template <int NFFT> constexpr bool gl_lean() { return NFFT == 1536; }
template <int NFFT> constexpr int gl_tables() { return gl_lean<NFFT>() ? 8 * (4 * Geo<NFFT>::kNC + Geo<NFFT>::Fft::kPdTable) : 0; }
template <int NFFT> constexpr int gl_smem() { return 8 * 3 * Geo<NFFT>::kTile + 4 * 2 * 2 * NFFT + gl_tables<NFFT>(); }    // 30,208 B at 1024, 77,824 B at 1536

// One workgroup (192 threads = 3 wavefronts = 3 columns) runs all iterations for stream `b`.  `smem`: gl_smem<NFFT>() bytes.
// STREAM = true: instead of storing the frame, fold it into the stream's overlap-add line (P12, app3.py:219-224):
//   hop_out <- ola[:hop] (float, or clipped int16 as app3.py:244-245); ola <- concat(ola[hop:], 0) + frame.
// EXIT_GE: how the loop tests for its last iteration.  Semantically the same (n_last >= it_begin); the compiler lays the loop out differently, and
// measurably so at n_fft 1536 (256 registers, a few values in scratch): `==` suits the chain that runs to the end (115 -> 103 us per batch-256 hop),
// `>=` the head start that stops early (its iterations 17.2 k -> 13.5 k ticks).  Each call site of dn_hop.hip uses what measured best.
template <int NFFT, bool FROM_MEL, bool STREAM = false, bool EXIT_GE = false>
__device__ __forceinline__ void gl_body(char* smem, const DspDev& d, const float* __restrict__ mag,
                                        const float* __restrict__ diff, const v2f* __restrict__ init, uint64_t seed,
                                        uint64_t sid0, const float* __restrict__ scale, float* __restrict__ wave,
                                        int n_iter, float mom, size_t b, int tid, float* ola = nullptr,
                                        void* hop_out = nullptr, int out_s16 = 0, int it_begin = 0, int it_stop = -1,
                                        v2f* state = nullptr) {
    // it_begin / it_stop / state: the chain can be cut at the top of an iteration.  A call with it_stop = s >= 0 runs iterations
    // [it_begin, s) and leaves X = angles * magnitude and the previous rebuilt spectrum of every lane in `state`
    // ([stream][column][2 NV + 2][64] complex, one coalesced row per register); a call with it_begin = s > 0 picks them up and runs on.
    // Everything else an iteration needs is rebuilt from those two, so the continuation is bit-identical to the uncut chain.
    using G = Geo<NFFT>;
    constexpr int kNR = G::kNR, kNC = G::kNC, kHop = G::kHop, kBins = G::kBins, kNV = G::kNV, kNP = G::kNP, kFftTile = G::kTile;
    constexpr int kBinPad = (kBins + 7) & ~7;                  // row stride of the LDS magnitude scratch
    v2f (*tile)[kFftTile] = reinterpret_cast<v2f (*)[kFftTile]>(smem);
    // [ping-pong][0: centre column, 1: halves of columns 0 and 2][n]
    float (*ybuf)[2][kNR] = reinterpret_cast<float (*)[2][kNR]>(smem + 8 * 3 * kFftTile);
    constexpr bool kLean = gl_lean<NFFT>();
    v2f* wsyn_t = reinterpret_cast<v2f*>(smem + 8 * 3 * kFftTile + 4 * 2 * 2 * kNR);        // [NC]       lean only
    v2f* cw_t = wsyn_t + kNC;                                                                // [3][NC]
    v2f* pd_t = kLean ? cw_t + 3 * kNC : nullptr;                                            // [11][64]

    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    v2f* mytile = tile[w];

    // prologue scratch aliases the overlap-add lines (used only before the first iteration)
    float* mm = &ybuf[0][0][0];           // [3][128]      mel magnitudes
    float* lmag = &ybuf[1][0][0];         // [3][kBinPad]  linear magnitudes
    static_assert(3 * kBinPad <= 2 * kNR && 3 * 128 <= 2 * kNR, "prologue scratch fits one ping-pong half");
    if (FROM_MEL) {
        const int M = d.n_mels;
        const bool factored = d.ginv_band != nullptr;
        float4* mel4 = reinterpret_cast<float4*>(mm);     // factored form: (column 0, 1, 2) per filter, kInvBand zero rows on either side
        if (factored) {
            for (int i = tid; i < kInvMel4; i += kGlThreads) mel4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            __syncthreads();
        }
        for (int i = tid; i < 3 * M; i += kGlThreads) {
            const int c = i / M, m = i - c * M;
            float v = mag[(b * 3 + c) * M + m] - diff[(b * 3 + c) * M + m];
            v = v >= 0.0f ? v : 0.2f * v;              // leaky_relu, app3.py:204
            v = fmaxf(fast_expm1(v), 0.0f);            // app3.py:207-208 (as dn_invmel_body.hpp)
            if (factored) reinterpret_cast<float*>(mel4 + kInvBand + m)[c] = v;
            else mm[c * 128 + m] = v;
        }
        __syncthreads();
        constexpr int kRounds = (kBins + kGlThreads - 1) / kGlThreads;
        if (factored) {
            // factored form, in the canonical order of dn_invmel_body.hpp (the FFT tiles are free until the first iteration)
            float4* yp4 = reinterpret_cast<float4*>(smem);
            float4* y4 = yp4 + kInvThirds * kMaxMels;
            static_assert(16 * (kInvThirds + 1) * kMaxMels <= 8 * 3 * kFftTile && 16 * kInvMel4 <= 4 * 2 * kNR, "the fold's scratch fits");
            InvBandFold<kGlThreads> inv;
            inv.fetch(d, tid);
            inv.fold(d, mel4, yp4, y4, tid);
#pragma unroll
            for (int r = 0; r < kRounds; ++r) {
                const int k = tid + kGlThreads * r;
                if (k < kBins) {
                    const float4 o = invmel_bin(d.fb2[k], y4);
                    lmag[0 * kBinPad + k] = o.x; lmag[1 * kBinPad + k] = o.y; lmag[2 * kBinPad + k] = o.z;
                }
            }
            __syncthreads();
        } else {
        // thread <-> bins tid, tid+192, ..: each pinv element is loaded once and used for all 3 columns (pinv_t rows are
        // zero padded to pinv_stride >= kRounds*192, so the tail loads are in bounds)
        float acc[kRounds][3];
#pragma unroll
        for (int r = 0; r < kRounds; ++r) acc[r][0] = acc[r][1] = acc[r][2] = 0.0f;
        const float* p = d.pinv_t + tid;
#pragma unroll 8
        for (int m = 0; m < M; ++m) {
            const float* pm = p + (size_t)m * d.pinv_stride;
            const float m0 = mm[m], m1 = mm[128 + m], m2 = mm[256 + m];
#pragma unroll
            for (int r = 0; r < kRounds; ++r) {
                const float pv = pm[kGlThreads * r];
                acc[r][0] = fmaf(pv, m0, acc[r][0]); acc[r][1] = fmaf(pv, m1, acc[r][1]); acc[r][2] = fmaf(pv, m2, acc[r][2]);
            }
        }
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            const int k = tid + kGlThreads * r;
            if (k < kBins) {
#pragma unroll
                for (int c = 0; c < 3; ++c) lmag[c * kBinPad + k] = fmaxf(acc[r][c], 0.0f);    // relu + clamp, app3.py:210-211
            }
        }
        __syncthreads();
        }
    }

    typename G::Fft::Tw tw;
    G::Fft::template load<!kLean>(tw, reinterpret_cast<const v2f*>(d.twc), lane);
    if (kLean && w == 1) G::Fft::fill_pd(pd_t, reinterpret_cast<const v2f*>(d.twc), lane);

    // lane constants: bin twiddles of the owned pairs, synthesis window / NC, analysis window * 1/envelope
    // and the source sample indices of this column in the rebuilt signal s[0..n_fft)  (H = hop = n_fft/2):
    //   column 0: n < H -> s[H-n] (reflection), else s[n-H]
    //   column 1: s[n]
    //   column 2: n < H -> s[n+H], else s[3H-2-n] (reflection)
    v2f wkh[kNP], wsyn[kLean ? 1 : kNV], cw[kLean ? 1 : kNV];
#pragma unroll
    for (int t = 0; t < kNP; ++t) wkh[t] = cscale(reinterpret_cast<const v2f*>(d.twr)[lane + 64 * t], 0.5f);
#pragma unroll
    for (int t = 0; t < kNV; ++t) {
        const int m = lane + 64 * t;
        const v2f ww = reinterpret_cast<const v2f*>(d.window)[m];
        const v2f ws = cscale(ww, 1.0f / (float)kNC);
        const int n0 = 2 * m, n1 = n0 + 1;
        int i0, i1;
        if (w == 1) { i0 = n0; i1 = n1; }
        else if (w == 0) { i0 = n0 < kHop ? kHop - n0 : n0 - kHop; i1 = n1 < kHop ? kHop - n1 : n1 - kHop; }
        else { i0 = n0 < kHop ? n0 + kHop : 3 * kHop - 2 - n0; i1 = n1 < kHop ? n1 + kHop : 3 * kHop - 2 - n1; }
        const v2f cc = mk2(ww[0] * d.inv_env[i0], ww[1] * d.inv_env[i1]);
        if (kLean) {
            if (w == 0) wsyn_t[m] = ws;           // the same for every column: one table
            cw_t[w * kNC + m] = cc;
        } else {
            wsyn[t] = ws;
            cw[t] = cc;
        }
    }

    // per-lane state: the NP bin pairs (k, NC-k), k = lane + 64 t, plus bin NC/2 (meaningful in lane 0)
    float mlo[kNP], mhi[kNP], mmid;
    v2f alo[kNP], ahi[kNP], amid, plo[kNP], phi[kNP], pmid;
    {
        const size_t row = (b * 3 + w) * kBins;
#pragma unroll
        for (int t = 0; t < kNP; ++t) {
            const int k = lane + 64 * t, kh = kNC - k;
            mlo[t] = FROM_MEL ? lmag[w * kBinPad + k] : (mag != nullptr ? mag[row + k] : 1.0f);
            mhi[t] = FROM_MEL ? lmag[w * kBinPad + kh] : (mag != nullptr ? mag[row + kh] : 1.0f);
            if (it_begin == 0) {
                if (init != nullptr) { alo[t] = init[row + k]; ahi[t] = init[row + kh]; }
                else rand_angle_pair(seed, sid0 + b, w, k, alo[t], ahi[t]);
            }
            plo[t] = mk2(0.0f, 0.0f);
            phi[t] = mk2(0.0f, 0.0f);
        }
        mmid = FROM_MEL ? lmag[w * kBinPad + kNC / 2] : (mag != nullptr ? mag[row + kNC / 2] : 1.0f);
        if (it_begin == 0) amid = init != nullptr ? init[row + kNC / 2] : rand_angle_mid(seed, sid0 + b, w, kNC);
        pmid = mk2(0.0f, 0.0f);
    }

    // a = rebuilt - m * tprev; tprev = rebuilt; angles = a / (|a| + 1e-16).
    // The normalisation is one v_rsq: 1/sqrt(|a|^2 + 1e-32) equals 1/(|a| + 1e-16) to within fp32 rounding unless |a| < ~1e-12, where both
    // forms only decide how a bin that carries no energy is scaled (exactly 0 stays 0 in both).  It saves a sqrt, an add and a
    // quarter-rate op per bin on the critical path of every iteration.
    // The state keeps X = angles * magnitude (what the next istft consumes) instead of the angles themselves.
    auto update = [mom](v2f reb, v2f& prev, v2f& x, float m) {
        const v2f a = reb - prev * mom;
        prev = reb;
        const float inv = __builtin_amdgcn_rsqf(fmaf(a[0], a[0], fmaf(a[1], a[1], 1e-32f)));
        x = a * (inv * m);
    };

    if (FROM_MEL || kLean) __syncthreads();   // the prologue scratch becomes the overlap-add lines; the shared lean tables are complete

    // The iteration loop is instantiated once per column (W is a compile-time constant inside): the overlap-add
    // stores and every other column-dependent choice become straight-line code instead of per-value predicates.
    v2f* mystate = state != nullptr ? state + ((b * 3 + w) * (2 * kNV + 2)) * 64 + lane : nullptr;
    auto iterate = [&](auto wc) {
    constexpr int W = decltype(wc)::value;
    v2f v[kNV], xlo[kNP], xhi[kNP], xmid, rlo[kNP], rhi[kNP], rmid;
    if (it_begin > 0) {
#pragma unroll
        for (int t = 0; t < kNP; ++t) {
            xlo[t] = mystate[64 * t]; xhi[t] = mystate[64 * (kNP + t)];
            plo[t] = mystate[64 * (kNV + 1 + t)]; phi[t] = mystate[64 * (kNV + 1 + kNP + t)];
        }
        xmid = mystate[64 * kNV];
        pmid = mystate[64 * (2 * kNV + 1)];
    } else {
#pragma unroll
        for (int t = 0; t < kNP; ++t) {
            xlo[t] = alo[t] * mlo[t];
            xhi[t] = ahi[t] * mhi[t];
        }
        xmid = amid * mmid;
    }
    const int n_last = max(n_iter, it_begin);       // a chain resumed at or past its end still terminates: it goes straight to the final istft
    for (int it = it_begin;; ++it) {
        if (it == it_stop) {          // hand the chain over (uniform)
#pragma unroll
            for (int t = 0; t < kNP; ++t) {
                mystate[64 * t] = xlo[t]; mystate[64 * (kNP + t)] = xhi[t];
                mystate[64 * (kNV + 1 + t)] = plo[t]; mystate[64 * (kNV + 1 + kNP + t)] = phi[t];
            }
            mystate[64 * kNV] = xmid;
            mystate[64 * (2 * kNV + 1)] = pmid;
            break;
        }
        // ---- istft of X = angles * magnitude: Hermitian merge, inverse FFT, synthesis window, overlap-add lines
        DN_STAMP(0);
        irfft_merge_pairs<kNV>(xlo, xhi, xmid, wkh, lane, v);
        DN_STAMP(1);
        G::Fft::template run<true>(v, tw, mytile, lane, pd_t);
        DN_STAMP(2);
        float* y1 = ybuf[it & 1][0];
        float* yo = ybuf[it & 1][1];
        {
            // column 1 -> y1[n]; column 0 keeps its second half -> yo[n-H]; column 2 its first half -> yo[n+H]
            // (the other halves fall outside the samples the istft trim keeps and are simply not stored)
            float* ydst = W == 1 ? y1 : (W == 0 ? yo - kHop : yo + kHop);
            constexpr int t_lo = W == 0 ? kNP : 0, t_hi = W == 2 ? kNP : kNV;
#pragma unroll
            for (int t = t_lo; t < t_hi; ++t)
                *reinterpret_cast<v2f*>(ydst + 2 * (lane + 64 * t)) = v[t] * (kLean ? wsyn_t[lane + 64 * t] : wsyn[kLean ? 0 : t]);
        }
        DN_STAMP(3);
        __syncthreads();
        DN_STAMP(4);
        if (EXIT_GE ? it >= n_last : it == n_last) {
            // final istft: divide by the window envelope, trim, scale (app3.py:217 `* peak`)
            const float sc = scale != nullptr ? scale[b] : 1.0f;
            if (!STREAM) {
                for (int n = tid; n < kNR; n += kGlThreads)
                    wave[b * kNR + n] = (y1[n] + yo[n]) * d.inv_env[n] * sc;
            } else {
                constexpr int kR = (kNR + kGlThreads - 1) / kGlThreads;
                float* orow = ola + b * kNR;
                float cur[kR], nxt[kR];
#pragma unroll
                for (int r = 0; r < kR; ++r) {
                    const int n = tid + kGlThreads * r;
                    cur[r] = n < kNR ? orow[n] : 0.0f;
                    nxt[r] = n + kNR / 2 < kNR ? orow[n + kNR / 2] : 0.0f;
                }
                __syncthreads();          // every old sample is in registers before the line is rewritten
#pragma unroll
                for (int r = 0; r < kR; ++r) {
                    const int n = tid + kGlThreads * r;
                    if (n < kNR) {
                        orow[n] = fmaf((y1[n] + yo[n]) * d.inv_env[n], sc, nxt[r]);     // (the contraction the compiler makes anyway, spelled out: dn_glw_body.hpp must round alike)
                        if (n < kNR / 2) {
                            if (out_s16) {
                                const float c = fminf(fmaxf(cur[r], -1.0f), 1.0f) * 32767.0f;      // np.clip, * iinfo(int16).max
                                static_cast<short*>(hop_out)[b * (kNR / 2) + n] = (short)c;          // astype(int16): truncation
                            } else {
                                static_cast<float*>(hop_out)[b * (kNR / 2) + n] = cur[r];
                            }
                        }
                    }
                }
            }
            break;
        }
        // ---- stft of the rebuilt signal (centre, reflect): this wave's column, analysis window
#pragma unroll
        for (int t = 0; t < kNV; ++t) {
            // source samples of (n0, n0+1), n0 = 2 (lane + 64 t): n0 < H exactly when t < NP, so the reflection case is
            // known per t at compile time and every address is +-2*lane plus a constant
            const int n0 = 2 * (lane + 64 * t);
            int i0, i1;
            if (W == 1) { i0 = n0; i1 = n0 + 1; }
            else if (W == 0) { if (t < kNP) { i0 = kHop - n0; i1 = i0 - 1; } else { i0 = n0 - kHop; i1 = i0 + 1; } }
            else { if (t < kNP) { i0 = n0 + kHop; i1 = i0 + 1; } else { i0 = 3 * kHop - 2 - n0; i1 = i0 - 1; } }
            v[t] = mk2(y1[i0] + yo[i0], y1[i1] + yo[i1]) * (kLean ? cw_t[W * kNC + lane + 64 * t] : cw[kLean ? 0 : t]);
        }
        DN_STAMP(5);
        G::Fft::template run<false>(v, tw, mytile, lane, pd_t);
        DN_STAMP(6);
        rfft_split_pairs<kNV>(v, wkh, lane, rlo, rhi, rmid);
        DN_STAMP(7);
        // ---- phase update with momentum
#pragma unroll
        for (int t = 0; t < kNP; ++t) {
            update(rlo[t], plo[t], xlo[t], mlo[t]);
            update(rhi[t], phi[t], xhi[t], mhi[t]);
        }
        update(rmid, pmid, xmid, mmid);
        DN_STAMP(8);
    }
    };
    if (w == 0) iterate(std::integral_constant<int, 0>{});
    else if (w == 1) iterate(std::integral_constant<int, 1>{});
    else iterate(std::integral_constant<int, 2>{});
}

}  // namespace dn
