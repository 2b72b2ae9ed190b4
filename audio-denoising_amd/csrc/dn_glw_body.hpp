// dn_glw_body.hpp -- Griffin-Lim with ONE WAVEFRONT PER STREAM: the schedule for the saturated regime (several streams per CU).
//
// dn_gl_body.hpp gives a stream's three STFT columns to three wavefronts: the shortest possible chain for ONE stream (what the batch-256
// launch needs: one stream per CU), but each of those waves is a latency chain -- four LDS tile exchanges, two ds_bpermute rounds and a
// workgroup barrier per iteration that nothing overlaps -- and with several streams per CU the machine stays latency bound: at 1,024 and
// 8,192 streams the counters see the VALU busy 38 % of the time and 1.4 waves per SIMD (profiles/r03_pmc_saturated.txt).
// Here a wavefront owns a whole stream:
//   * the three columns are independent between two overlap-add points, so the wave runs them in lock step (columns 0 and 2 as a pair of
//     transforms, then column 1): every exchange of one transform is in flight while the butterflies of the other issue;
//   * no workgroup barrier at all: the overlap-add is LANE LOCAL -- sample pair (2m, 2m+1) of column 1 meets pair m + 256 of column 0 and
//     pair m - 256 of column 2, the same lane four registers further -- so the rebuilt signal is summed in registers, written to a
//     wave-private LDS line once per iteration, and read back only for the two reflected half columns;
//   * four streams per workgroup, two workgroups per CU: eight streams in flight per CU instead of two.
// Arithmetic per column is that of gl_body, operation for operation (same transforms, same window products, (centre + side) in the
// overlap-add), so the result is bit-identical; only who executes it changes.  n_fft 1024 only (at 1536 the per-lane state does not fit).
#pragma once
#include "dn_gl_body.hpp"

namespace dn {

constexpr int kGlwStreams = 4;        // streams (= wavefronts) per workgroup

template <int NFFT> struct GlwLds {
    using G = Geo<NFFT>;
    static constexpr int kCw = 0;                                       // v2f [3][NC]  analysis window x 1/envelope of each column's source samples
    static constexpr int kWsyn = kCw + 8 * 3 * G::kNC;                  // v2f [NC]     synthesis window / NC
    static constexpr int kWave = kWsyn + 8 * G::kNC;                    // per wavefront: two exchange tiles, the rebuilt signal
    static constexpr int kPerWave = 8 * 2 * G::kTile + 4 * NFFT;
    static constexpr int kTotal = kWave + kGlwStreams * kPerWave;
};
template <int NFFT> constexpr int glw_smem() { return GlwLds<NFFT>::kTotal; }

// the lane-indexed window tables, once per workgroup (every thread of the workgroup calls this; a workgroup barrier follows)
template <int NFFT, int THREADS>
__device__ __forceinline__ void glw_fill_tables(char* smem, const DspDev& d, int tid) {
    using G = Geo<NFFT>;
    constexpr int kNC = G::kNC, kHop = G::kHop;
    v2f* cw_t = reinterpret_cast<v2f*>(smem + GlwLds<NFFT>::kCw);
    v2f* wsyn_t = reinterpret_cast<v2f*>(smem + GlwLds<NFFT>::kWsyn);
    for (int m = tid; m < kNC; m += THREADS) {
        const v2f ww = reinterpret_cast<const v2f*>(d.window)[m];
        wsyn_t[m] = cscale(ww, 1.0f / (float)kNC);
        const int n0 = 2 * m, n1 = n0 + 1;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int i0, i1;
            if (c == 1) { i0 = n0; i1 = n1; }
            else if (c == 0) { i0 = n0 < kHop ? kHop - n0 : n0 - kHop; i1 = n1 < kHop ? kHop - n1 : n1 - kHop; }
            else { i0 = n0 < kHop ? n0 + kHop : 3 * kHop - 2 - n0; i1 = n1 < kHop ? n1 + kHop : 3 * kHop - 2 - n1; }
            cw_t[c * kNC + m] = mk2(ww[0] * d.inv_env[i0], ww[1] * d.inv_env[i1]);
        }
    }
}

// One wavefront (`lane`, wave `wv` of its workgroup) runs the whole chain of stream `b`.  mag: linear magnitudes [B][3][K].
template <int NFFT, bool STREAM>
__device__ __forceinline__ void glw_body(char* smem, const DspDev& d, const float* __restrict__ mag, const v2f* __restrict__ init,
                                         uint64_t seed, uint64_t sid0, const float* __restrict__ scale, float* __restrict__ wave,
                                         int n_iter, float mom, size_t b, int lane, int wv, float* ola, void* hop_out, int out_s16,
                                         int it_begin, const v2f* __restrict__ state) {
    using G = Geo<NFFT>;
    using L = GlwLds<NFFT>;
    static_assert(NFFT == 1024, "one wavefront per stream is built for n_fft 1024");
    constexpr int kNR = G::kNR, kNC = G::kNC, kHop = G::kHop, kBins = G::kBins, kNV = G::kNV, kNP = G::kNP, kTile = G::kTile;
    constexpr int kHalf = kNV / 2;                       // registers between a sample pair and the one a hop further
    const v2f* cw_t = reinterpret_cast<const v2f*>(smem + L::kCw);
    const v2f* wsyn_t = reinterpret_cast<const v2f*>(smem + L::kWsyn);
    char* mine = smem + L::kWave + wv * L::kPerWave;
    v2f* tile0 = reinterpret_cast<v2f*>(mine);
    v2f* tile1 = tile0 + kTile;
    float* sl = reinterpret_cast<float*>(mine + 8 * 2 * kTile);      // rebuilt signal s[0..n_fft), as of the last overlap-add

    typename G::Fft::Tw tw;
    G::Fft::template load<true>(tw, reinterpret_cast<const v2f*>(d.twc), lane);
    v2f wkh[kNP];
#pragma unroll
    for (int t = 0; t < kNP; ++t) wkh[t] = cscale(reinterpret_cast<const v2f*>(d.twr)[lane + 64 * t], 0.5f);

    // per-lane state of the three columns: bin pairs (k, NC-k), k = lane + 64 t, plus bin NC/2 (meaningful in lane 0)
    float mlo[3][kNP], mhi[3][kNP], mmid[3];
    v2f plo[3][kNP], phi[3][kNP], pmid[3];
    v2f snew[kNV];            // the rebuilt signal: sample pairs (2m, 2m+1), m = lane + 64 t

    auto update = [mom](v2f reb, v2f& prev, v2f& x, float m) {      // as gl_body
        const v2f a = reb - prev * mom;
        prev = reb;
        const float inv = __builtin_amdgcn_rsqf(fmaf(a[0], a[0], fmaf(a[1], a[1], 1e-32f)));
        x = a * (inv * m);
    };
    // istft of the side columns (0, 2) as a pair, then the centre column, from X = angles * magnitude; leaves the overlap-add in snew
    auto synthesize = [&](v2f (&xlo)[3][kNP], v2f (&xhi)[3][kNP], v2f (&xmid)[3]) {
        {
            v2f v[2][kNV];
            irfft_merge_pairs<kNV>(xlo[0], xhi[0], xmid[0], wkh, lane, v[0]);
            irfft_merge_pairs<kNV>(xlo[2], xhi[2], xmid[2], wkh, lane, v[1]);
            v2f* const tiles[2] = {tile0, tile1};
            G::Fft::template run_n<true, 2>(v, tw, tiles, lane);
            // column 0 keeps its second half -> s[n - H]; column 2 its first half -> s[n + H] (the other halves fall outside the istft trim)
#pragma unroll
            for (int t = 0; t < kHalf; ++t) {
                snew[t] = cmul_elem(v[0][t + kHalf], wsyn_t[lane + 64 * (t + kHalf)]);
                snew[t + kHalf] = cmul_elem(v[1][t], wsyn_t[lane + 64 * t]);
            }
        }
        {
            v2f v[1][kNV];
            irfft_merge_pairs<kNV>(xlo[1], xhi[1], xmid[1], wkh, lane, v[0]);
            v2f* const tiles[1] = {tile0};
            G::Fft::template run_n<true, 1>(v, tw, tiles, lane);
#pragma unroll
            for (int t = 0; t < kNV; ++t) snew[t] = cadd(cmul_elem(v[0][t], wsyn_t[lane + 64 * t]), snew[t]);       // (centre + side) of rounded products, as gl_body
        }
        // the reflected half columns of the next analysis read the line; this wave's LDS operations execute in order
        wave_sync();
#pragma unroll
        for (int t = 0; t < kNV; ++t) *reinterpret_cast<v2f*>(sl + 2 * (lane + 64 * t)) = snew[t];
        wave_sync();
    };

    {   // ---- prologue: magnitudes, initial phases (injected, drawn, or the chain a head start parked), first synthesis
        v2f xlo[3][kNP], xhi[3][kNP], xmid[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const size_t row = (b * 3 + c) * kBins;
            const v2f* st = state != nullptr ? state + ((b * 3 + c) * (2 * kNV + 2)) * 64 + lane : nullptr;
#pragma unroll
            for (int t = 0; t < kNP; ++t) {
                const int k = lane + 64 * t, kh = kNC - k;
                mlo[c][t] = mag != nullptr ? mag[row + k] : 1.0f;
                mhi[c][t] = mag != nullptr ? mag[row + kh] : 1.0f;
                if (it_begin > 0) {
                    xlo[c][t] = st[64 * t]; xhi[c][t] = st[64 * (kNP + t)];
                    plo[c][t] = st[64 * (kNV + 1 + t)]; phi[c][t] = st[64 * (kNV + 1 + kNP + t)];
                } else {
                    const v2f alo = init != nullptr ? init[row + k] : rand_angle(seed, sid0 + b, c, k);
                    const v2f ahi = init != nullptr ? init[row + kh] : rand_angle(seed, sid0 + b, c, kh);
                    xlo[c][t] = alo * mlo[c][t];
                    xhi[c][t] = ahi * mhi[c][t];
                    plo[c][t] = mk2(0.0f, 0.0f);
                    phi[c][t] = mk2(0.0f, 0.0f);
                }
            }
            mmid[c] = mag != nullptr ? mag[row + kNC / 2] : 1.0f;
            if (it_begin > 0) {
                xmid[c] = st[64 * kNV];
                pmid[c] = st[64 * (2 * kNV + 1)];
            } else {
                const v2f amid = init != nullptr ? init[row + kNC / 2] : rand_angle(seed, sid0 + b, c, kNC / 2);
                xmid[c] = amid * mmid[c];
                pmid[c] = mk2(0.0f, 0.0f);
            }
        }
        synthesize(xlo, xhi, xmid);
    }

    for (int it = it_begin; it < n_iter; ++it) {
        // ---- stft of the rebuilt signal (centre, reflect) -> phase update -> istft, side columns as a pair, then the centre column
        v2f xlo[3][kNP], xhi[3][kNP], xmid[3];
        {
            v2f v[2][kNV];
#pragma unroll
            for (int t = 0; t < kNV; ++t) {
                const int n0 = 2 * (lane + 64 * t);
                // column 0: n0 < H -> s[H - n0], s[H - n0 - 1] (reflection), else the pair a hop earlier (same lane, register t - 4)
                // column 2: n0 < H -> the pair a hop later (register t + 4), else s[3H - 2 - n0], s[3H - 3 - n0] (reflection)
                const v2f s0 = t < kNP ? mk2(sl[kHop - n0], sl[kHop - n0 - 1]) : snew[t - kHalf];
                const v2f s2 = t < kNP ? snew[t + kHalf] : mk2(sl[3 * kHop - 2 - n0], sl[3 * kHop - 3 - n0]);
                v[0][t] = s0 * cw_t[0 * kNC + lane + 64 * t];
                v[1][t] = s2 * cw_t[2 * kNC + lane + 64 * t];
            }
            v2f* const tiles[2] = {tile0, tile1};
            G::Fft::template run_n<false, 2>(v, tw, tiles, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = 2 * j;
                v2f rlo[kNP], rhi[kNP], rmid;
                rfft_split_pairs<kNV>(v[j], wkh, lane, rlo, rhi, rmid);
#pragma unroll
                for (int t = 0; t < kNP; ++t) {
                    update(rlo[t], plo[c][t], xlo[c][t], mlo[c][t]);
                    update(rhi[t], phi[c][t], xhi[c][t], mhi[c][t]);
                }
                update(rmid, pmid[c], xmid[c], mmid[c]);
            }
        }
        {
            v2f v[1][kNV];
#pragma unroll
            for (int t = 0; t < kNV; ++t) v[0][t] = snew[t] * cw_t[1 * kNC + lane + 64 * t];
            v2f* const tiles[1] = {tile0};
            G::Fft::template run_n<false, 1>(v, tw, tiles, lane);
            v2f rlo[kNP], rhi[kNP], rmid;
            rfft_split_pairs<kNV>(v[0], wkh, lane, rlo, rhi, rmid);
#pragma unroll
            for (int t = 0; t < kNP; ++t) {
                update(rlo[t], plo[1][t], xlo[1][t], mlo[1][t]);
                update(rhi[t], phi[1][t], xhi[1][t], mhi[1][t]);
            }
            update(rmid, pmid[1], xmid[1], mmid[1]);
        }
        synthesize(xlo, xhi, xmid);
    }

    // ---- final istft: divide by the window envelope, trim, scale (app3.py:217 `* peak`); streaming: fold into the overlap-add line (P12)
    const float sc = scale != nullptr ? scale[b] : 1.0f;
    if (!STREAM) {
#pragma unroll
        for (int t = 0; t < kNV; ++t) {
            const int n = 2 * (lane + 64 * t);
            const v2f e = *reinterpret_cast<const v2f*>(d.inv_env + n);
            *reinterpret_cast<v2f*>(wave + b * kNR + n) = mk2(snew[t][0] * e[0] * sc, snew[t][1] * e[1] * sc);
        }
    } else {
        // hop_out <- ola[:hop]; ola <- concat(ola[hop:], 0) + frame   (app3.py:219-224).  The lane that owns sample pair n also owns pair
        // n + hop (four registers further): every old sample is in registers before the line is rewritten, by construction.
        float* orow = ola + b * kNR;
        v2f cur[kHalf], nxt[kHalf];
#pragma unroll
        for (int t = 0; t < kHalf; ++t) {
            const int n = 2 * (lane + 64 * t);
            cur[t] = *reinterpret_cast<const v2f*>(orow + n);
            nxt[t] = *reinterpret_cast<const v2f*>(orow + n + kHop);
        }
#pragma unroll
        for (int t = 0; t < kNV; ++t) {
            const int n = 2 * (lane + 64 * t);
            const v2f e = *reinterpret_cast<const v2f*>(d.inv_env + n);
            const v2f old = t < kHalf ? nxt[t % kHalf] : mk2(0.0f, 0.0f);
            *reinterpret_cast<v2f*>(orow + n) = mk2(fmaf(snew[t][0] * e[0], sc, old[0]), fmaf(snew[t][1] * e[1], sc, old[1]));     // as gl_body
        }
#pragma unroll
        for (int t = 0; t < kHalf; ++t) {
            const int n = 2 * (lane + 64 * t);
            if (out_s16) {
                const float c0 = fminf(fmaxf(cur[t][0], -1.0f), 1.0f) * 32767.0f, c1 = fminf(fmaxf(cur[t][1], -1.0f), 1.0f) * 32767.0f;   // np.clip, * iinfo(int16).max
                short* q = static_cast<short*>(hop_out) + b * (kNR / 2) + n;
                q[0] = (short)c0; q[1] = (short)c1;                                                                                         // astype(int16): truncation
            } else {
                *reinterpret_cast<v2f*>(static_cast<float*>(hop_out) + b * (kNR / 2) + n) = cur[t];
            }
        }
    }
}

}  // namespace dn
