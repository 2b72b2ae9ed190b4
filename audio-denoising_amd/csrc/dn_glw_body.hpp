// dn_glw_body.hpp -- Griffin-Lim with ONE WAVEFRONT PER STREAM: the schedule for the saturated regime (several streams per CU).
//
// dn_gl_body.hpp gives a stream's three STFT columns to three wavefronts: the shortest possible chain for ONE stream (what the batch-256
// launch needs: one stream per CU), but each of those waves is a latency chain -- four LDS tile exchanges, two ds_bpermute rounds and a
// workgroup barrier per iteration that nothing overlaps -- and with several streams per CU the machine stays latency bound: at 1,024 and
// 8,192 streams the counters see the VALU busy 38 % of the time and 1.4 waves per SIMD (profiles/r03_pmc_saturated.txt).
// Here a wavefront owns a whole stream:
//   * the three columns are independent between two overlap-add points, so the wave runs their transforms as one skewed group
//     (WaveFft::run_n: every exchange of one transform is in flight while the butterflies of the next issue; two LDS tiles serve the three);
//   * no workgroup barrier at all: the overlap-add is LANE LOCAL -- sample pair (2m, 2m+1) of column 1 meets pair m + 256 of column 0 and
//     pair m - 256 of column 2, the same lane four registers further -- so the rebuilt signal is summed in registers, written to a
//     wave-private LDS line once per iteration, and read back only for the two reflected half columns;
//   * four streams per workgroup, two workgroups per CU: eight streams in flight per CU instead of two.
// Arithmetic per column is that of gl_body, operation for operation (same transforms, same window products, (centre + side) in the
// overlap-add), so the result is bit-identical; only who executes it changes.  n_fft 1024 only (at 1536 the per-lane state does not fit).
#pragma once
#include "dn_gl_body.hpp"

namespace dn {

// n_fft 1024 only.  At 1536 (12 values a lane) the per-lane state of three columns -- 78 registers of previous spectra, 39 magnitudes, the old and the
// new signal, one transform -- does not fit the 256 registers of a wave that shares its SIMD: built and measured in round 3 (columns one after the other,
// radix-12 twiddles in LDS, three streams a workgroup): 360-400 B of scratch a lane, 46 scratch accesses in every iteration, and 1.6x SLOWER than a
// wavefront per column at 1,024 streams (662 against 410 us per hop), 1.7-2.3x slower as a deep pipe at 256.  Removed again; gl_body serves n_fft 1536.
// (Round 4 tried it once more with a SIMD's registers to itself -- a chains-only kernel, 420 registers, nothing spilled: still slower, dn_group.hip.)
constexpr int kGlwWaves = 4;          // wavefronts of a workgroup that run a chain (streams x chain segments)

#ifdef DN_PROBE
// diagnostic build only: s_memtime stamps of the four wavefronts of workgroup 0 (tools/glw_probe.py)
static __device__ unsigned long long g_glw_probe[4][8];
#define DN_WSTAMP(id) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_glw_probe[threadIdx.x >> 6][id] = t_; \
    __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DN_WSTAMP(id) do { } while (0)
#endif

template <int NFFT> struct GlwLds {
    using G = Geo<NFFT>;
    static constexpr int kCw = 0;                                       // v2f [3][NC]  analysis window x 1/envelope of each column's source samples
    static constexpr int kWsyn = kCw + 8 * 3 * G::kNC;                  // v2f [NC]     synthesis window / NC
    static constexpr int kWave = kWsyn + 8 * G::kNC;                    // per wavefront: two exchange tiles, the rebuilt signal
    static constexpr int kPerWave = 8 * 2 * G::kTile + 4 * NFFT;
    static constexpr int kTotal = kWave + kGlwWaves * kPerWave;
};
template <int NFFT> constexpr int glw_smem() { return GlwLds<NFFT>::kTotal; }

// the lane-indexed window tables (cw [3][NC], wsyn [NC]: 16 KB the plan holds ready-made), once per workgroup: every thread of the workgroup copies
// its share; an LDS-only workgroup barrier follows -- in glw_body for the waves that run a chain (their own loads stay in flight across it), in the
// kernel for the others
template <int NFFT, int THREADS>
__device__ __forceinline__ void glw_fill_tables(char* smem, const DspDev& d, int tid) {
    constexpr int kVec = 4 * Geo<NFFT>::kNC * 8 / 16;              // float4s
    static_assert(GlwLds<NFFT>::kWsyn == GlwLds<NFFT>::kCw + 8 * 3 * Geo<NFFT>::kNC && kVec % THREADS == 0, "the LDS tables are laid out as the plan's");
    const float4* src = reinterpret_cast<const float4*>(d.glw_tables);
    float4* dst = reinterpret_cast<float4*>(smem + GlwLds<NFFT>::kCw);
    float4 q[kVec / THREADS];
#pragma unroll
    for (int i = 0; i < kVec / THREADS; ++i) q[i] = src[tid + THREADS * i];
#pragma unroll
    for (int i = 0; i < kVec / THREADS; ++i) dst[tid + THREADS * i] = q[i];
}

// One wavefront (`lane`, wave `wv` of its workgroup) runs iterations [it_begin, it_stop) of the chain of stream `b` -- or, with it_stop < 0, everything
// from it_begin to the end, and emits the frame.  mag: linear magnitudes [B][3][K].  The chain can be cut between any two iterations without changing
// a bit (chain segments of a deep pipe).  A segment that stops early parks the chain in `state`; `resume` says where this one starts from:
//   kGlwFresh    the initial phases (injected or drawn);
//   kGlwFromX    what gl_body parks (a front workgroup's head start): X = angles * magnitude and the previous rebuilt spectrum of every lane;
//   kGlwFromSeg  what an earlier segment of this body parked: the rebuilt SIGNAL (n_fft floats -- the spectrum X is one synthesis away from it) and
//                the previous rebuilt spectrum, 17.5 KB a stream in 16-byte rows instead of gl_body's 27.6 KB in 8-byte rows.  Every launch of a
//                deep pipe moves this through HBM for every stream and segment boundary, all CUs at once, at the head of the launch.
constexpr int kGlwFresh = 0, kGlwFromX = 1, kGlwFromSeg = 2;
// EMIT: what the wave does with a finished frame:
//   kEmitFrame   store it (frame mode);
//   kEmitStream  fold it into the stream's overlap-add line and emit the hop (P12; one chain a stream and launch);
//   kEmitStage   leave frame x 1/envelope in the wave's LDS line: the workgroup folds the frames of its stream in order afterwards (hop groups:
//                several chains of ONE stream finish in the same launch and the overlap-add line is sequential)
constexpr int kEmitFrame = 0, kEmitStream = 1, kEmitStage = 2;
template <int NFFT> __device__ __forceinline__ float* glw_signal_line(char* smem, int wv) {
    return reinterpret_cast<float*>(smem + GlwLds<NFFT>::kWave + wv * GlwLds<NFFT>::kPerWave + 8 * 2 * Geo<NFFT>::kTile);
}
template <int NFFT, int EMIT>
__device__ __forceinline__ void glw_body(char* smem, const DspDev& d, const float* __restrict__ mag, const v2f* __restrict__ init,
                                         uint64_t seed, uint64_t sid0, const float* __restrict__ scale, float* __restrict__ wave,
                                         int n_iter, float mom, size_t b, int lane, int wv, float* ola, void* hop_out, int out_s16,
                                         int it_begin, int it_stop, int resume, v2f* __restrict__ state, int tid) {
    using G = Geo<NFFT>;
    using L = GlwLds<NFFT>;
    constexpr int kNR = G::kNR, kNC = G::kNC, kHop = G::kHop, kBins = G::kBins, kNV = G::kNV, kNP = G::kNP, kTile = G::kTile;
    constexpr int kHalf = kNV / 2;                       // registers between a sample pair and the one a hop further
    const v2f* cw_t = reinterpret_cast<const v2f*>(smem + L::kCw);
    const v2f* wsyn_t = reinterpret_cast<const v2f*>(smem + L::kWsyn);
    static_assert(NFFT == 1024, "one wavefront per stream is built for n_fft 1024 (see above)");
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
    // the three transforms of this wave, skewed (run_n); two tiles do: transform 2 takes over tile 0 after transform 0's last read of it (in order)
    auto fft = [&](auto inv, v2f (&v)[3][kNV]) {
        constexpr bool INV = decltype(inv)::value;
        v2f* const tiles[3] = {tile0, tile1, tile0};
        G::Fft::template run_n<INV, 3>(v, tw, tiles, lane);
    };
    // istft from X = angles * magnitude of the three columns (order in the batch: 0, 2, 1) into the NEXT rebuilt signal: column 0 keeps its second half
    // -> s[n - H], column 2 its first half -> s[n + H] (the other halves fall outside the istft trim), then the centre column on top: (centre + side)
    // of rounded products, as gl_body.  The new signal becomes THE signal: registers, and the wave's LDS line for the reflected half columns of the
    // next analysis (this wave's LDS operations execute in order).
    using XP = v2f[kNP];
    auto synthesize = [&](XP& x0lo, XP& x0hi, v2f x0mid, XP& x2lo, XP& x2hi, v2f x2mid, XP& x1lo, XP& x1hi, v2f x1mid) {
        v2f v[3][kNV];
        irfft_merge_pairs<kNV>(x0lo, x0hi, x0mid, wkh, lane, v[0]);
        irfft_merge_pairs<kNV>(x2lo, x2hi, x2mid, wkh, lane, v[1]);
        irfft_merge_pairs<kNV>(x1lo, x1hi, x1mid, wkh, lane, v[2]);
        fft(std::true_type{}, v);
#pragma unroll
        for (int t = 0; t < kHalf; ++t) {
            snew[t] = cmul_elem(v[0][t + kHalf], wsyn_t[lane + 64 * (t + kHalf)]);
            snew[t + kHalf] = cmul_elem(v[1][t], wsyn_t[lane + 64 * t]);
        }
#pragma unroll
        for (int t = 0; t < kNV; ++t) snew[t] = cadd(cmul_elem(v[2][t], wsyn_t[lane + 64 * t]), snew[t]);
        wave_sync();
#pragma unroll
        for (int t = 0; t < kNV; ++t) *reinterpret_cast<v2f*>(sl + 2 * (lane + 64 * t)) = snew[t];
        wave_sync();
    };

    const bool park = it_stop >= 0;
    const int it_last = park ? it_stop : n_iter;
    // segment format, per stream, rows of 64 lanes: 3 x kNP float4 rows (column c: plo pairs, then phi pairs), kNV / 2 float4 rows (signal pairs),
    // then three v2f rows (pmid)
    constexpr int kPrevRows = 3 * kNP, kSigRows = kNV / 2, kSegBytes = (kPrevRows + kSigRows) * 1024 + 3 * 512;
    static_assert(kNP % 2 == 0 && kSegBytes <= 8 * 3 * (2 * kNV + 2) * 64, "a parked segment fits the slot gl_body's format needs");
    char* seg = state != nullptr ? reinterpret_cast<char*>(state) + b * (size_t)(8 * 3 * (2 * kNV + 2) * 64) : nullptr;
    auto park_segment = [&]() {
        float4* q = reinterpret_cast<float4*>(seg) + lane;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int h = 0; h < kNP / 2; ++h) {
                q[64 * (kNP * c + h)] = make_float4(plo[c][2 * h][0], plo[c][2 * h][1], plo[c][2 * h + 1][0], plo[c][2 * h + 1][1]);
                q[64 * (kNP * c + kNP / 2 + h)] = make_float4(phi[c][2 * h][0], phi[c][2 * h][1], phi[c][2 * h + 1][0], phi[c][2 * h + 1][1]);
            }
        }
#pragma unroll
        for (int h = 0; h < kSigRows; ++h) q[64 * (kPrevRows + h)] = make_float4(snew[2 * h][0], snew[2 * h][1], snew[2 * h + 1][0], snew[2 * h + 1][1]);
        v2f* r = reinterpret_cast<v2f*>(seg + (kPrevRows + kSigRows) * 1024) + lane;
#pragma unroll
        for (int c = 0; c < 3; ++c) r[64 * c] = pmid[c];
    };
    {   // ---- prologue: magnitudes, initial phases (injected, drawn, or the chain a head start / an earlier segment parked), first synthesis
        // (three straight-line variants behind ONE uniform branch each: every load of a variant is in flight before the first is waited for)
        v2f xlo[3][kNP], xhi[3][kNP], xmid[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const size_t row = (b * 3 + c) * kBins;
#pragma unroll
            for (int t = 0; t < kNP; ++t) {
                const int k = lane + 64 * t, kh = kNC - k;
                mlo[c][t] = mag != nullptr ? mag[row + k] : 1.0f;
                mhi[c][t] = mag != nullptr ? mag[row + kh] : 1.0f;
            }
            mmid[c] = mag != nullptr ? mag[row + kNC / 2] : 1.0f;
        }
        if (resume == kGlwFromSeg) {
            const float4* q = reinterpret_cast<const float4*>(seg) + lane;
            float4 pq[kPrevRows], sq[kSigRows];
#pragma unroll
            for (int i = 0; i < kPrevRows; ++i) pq[i] = q[64 * i];
#pragma unroll
            for (int i = 0; i < kSigRows; ++i) sq[i] = q[64 * (kPrevRows + i)];
            const v2f* r = reinterpret_cast<const v2f*>(seg + (kPrevRows + kSigRows) * 1024) + lane;
#pragma unroll
            for (int c = 0; c < 3; ++c) pmid[c] = r[64 * c];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int h = 0; h < kNP / 2; ++h) {
                    const float4 lo4 = pq[kNP * c + h], hi4 = pq[kNP * c + kNP / 2 + h];
                    plo[c][2 * h] = mk2(lo4.x, lo4.y); plo[c][2 * h + 1] = mk2(lo4.z, lo4.w);
                    phi[c][2 * h] = mk2(hi4.x, hi4.y); phi[c][2 * h + 1] = mk2(hi4.z, hi4.w);
                }
#pragma unroll
            for (int h = 0; h < kSigRows; ++h) { snew[2 * h] = mk2(sq[h].x, sq[h].y); snew[2 * h + 1] = mk2(sq[h].z, sq[h].w); }
        } else if (resume == kGlwFromX) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const v2f* st = state + ((b * 3 + c) * (2 * kNV + 2)) * 64 + lane;
#pragma unroll
                for (int t = 0; t < kNP; ++t) {
                    xlo[c][t] = st[64 * t]; xhi[c][t] = st[64 * (kNP + t)];
                    plo[c][t] = st[64 * (kNV + 1 + t)]; phi[c][t] = st[64 * (kNV + 1 + kNP + t)];
                }
                xmid[c] = st[64 * kNV];
                pmid[c] = st[64 * (2 * kNV + 1)];
            }
        } else {
            if (init != nullptr) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const size_t row = (b * 3 + c) * kBins;
#pragma unroll
                    for (int t = 0; t < kNP; ++t) {
                        xlo[c][t] = init[row + lane + 64 * t];
                        xhi[c][t] = init[row + kNC - (lane + 64 * t)];
                    }
                    xmid[c] = init[row + kNC / 2];
                }
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
#pragma unroll
                    for (int t = 0; t < kNP; ++t) {
                        rand_angle_pair(seed, sid0 + b, c, lane + 64 * t, xlo[c][t], xhi[c][t]);
                    }
                    xmid[c] = rand_angle_mid(seed, sid0 + b, c, kNC);
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
#pragma unroll
                for (int t = 0; t < kNP; ++t) {
                    xlo[c][t] = xlo[c][t] * mlo[c][t];
                    xhi[c][t] = xhi[c][t] * mhi[c][t];
                    plo[c][t] = mk2(0.0f, 0.0f);
                    phi[c][t] = mk2(0.0f, 0.0f);
                }
                xmid[c] = xmid[c] * mmid[c];
                pmid[c] = mk2(0.0f, 0.0f);
            }
        }
        // every load of this wave's prologue is in flight: now the workgroup's window tables (a straight 16 KB copy) and the LDS-only barrier
        // that publishes them -- one trip to memory for all of it instead of four in a row
        glw_fill_tables<NFFT, 64 * kGlwWaves>(smem, d, tid);
        DN_LDS_BARRIER();
        if (resume == kGlwFromSeg) {
            // the reflected half columns of the analysis read the signal from the wave's LDS line
#pragma unroll
            for (int t = 0; t < kNV; ++t) *reinterpret_cast<v2f*>(sl + 2 * (lane + 64 * t)) = snew[t];
            wave_sync();
        }
        DN_WSTAMP(2);
        if (resume != kGlwFromSeg) synthesize(xlo[0], xhi[0], xmid[0], xlo[2], xhi[2], xmid[2], xlo[1], xhi[1], xmid[1]);
        DN_WSTAMP(3);
        if (park && it_begin >= it_last) {        // an empty segment: the chain stays (or, for a new one, is put) where the next segment expects it
            if (resume != kGlwFromSeg) park_segment();
            return;
        }
    }

    for (int it = it_begin; it < it_last; ++it) {
        if (it == it_begin + 1) DN_WSTAMP(4);
        // ---- stft of the rebuilt signal (centre, reflect) -> phase update with momentum -> istft: the three columns as one skewed group of
        // transforms in each direction (batch order 0, 2, 1).
        // column 0: n0 < H -> s[H - n0], s[H - n0 - 1] (reflection), else the pair a hop earlier (same lane, register t - kHalf)
        // column 2: n0 < H -> the pair a hop later (register t + kHalf), else s[3H - 2 - n0], s[3H - 3 - n0] (reflection)
        auto build0 = [&](v2f (&v)[kNV]) {
#pragma unroll
            for (int t = 0; t < kNV; ++t) {
                const int n0 = 2 * (lane + 64 * t);
                const v2f s0 = t < kNP ? mk2(sl[kHop - n0], sl[kHop - n0 - 1]) : snew[t - kHalf];
                v[t] = s0 * cw_t[0 * kNC + lane + 64 * t];
            }
        };
        auto build2 = [&](v2f (&v)[kNV]) {
#pragma unroll
            for (int t = 0; t < kNV; ++t) {
                const int n0 = 2 * (lane + 64 * t);
                const v2f s2 = t < kNP ? snew[t + kHalf] : mk2(sl[3 * kHop - 2 - n0], sl[3 * kHop - 3 - n0]);
                v[t] = s2 * cw_t[2 * kNC + lane + 64 * t];
            }
        };
        auto advance = [&](const v2f (&v)[kNV], auto col, XP& xlo, XP& xhi, v2f& xmid) {       // split, phase update
            constexpr int c = decltype(col)::value;
            v2f rlo[kNP], rhi[kNP], rmid;
            rfft_split_pairs<kNV>(v, wkh, lane, rlo, rhi, rmid);
#pragma unroll
            for (int t = 0; t < kNP; ++t) {
                update(rlo[t], plo[c][t], xlo[t], mlo[c][t]);
                update(rhi[t], phi[c][t], xhi[t], mhi[c][t]);
            }
            update(rmid, pmid[c], xmid, mmid[c]);
        };
        v2f x0lo[kNP], x0hi[kNP], x0mid, x2lo[kNP], x2hi[kNP], x2mid, x1lo[kNP], x1hi[kNP], x1mid;
        {
            v2f v[3][kNV];
            build0(v[0]);
            build2(v[1]);
#pragma unroll
            for (int t = 0; t < kNV; ++t) v[2][t] = snew[t] * cw_t[1 * kNC + lane + 64 * t];
            fft(std::false_type{}, v);
            advance(v[0], std::integral_constant<int, 0>{}, x0lo, x0hi, x0mid);
            advance(v[1], std::integral_constant<int, 2>{}, x2lo, x2hi, x2mid);
            advance(v[2], std::integral_constant<int, 1>{}, x1lo, x1hi, x1mid);
        }
        synthesize(x0lo, x0hi, x0mid, x2lo, x2hi, x2mid, x1lo, x1hi, x1mid);
    }
    if (park) {          // hand the chain over (uniform)
        DN_WSTAMP(5);
        park_segment();
        DN_WSTAMP(6);
        return;
    }

    DN_WSTAMP(5);
    // ---- final istft: divide by the window envelope, trim, scale (app3.py:217 `* peak`); streaming: fold into the overlap-add line (P12)
    if (EMIT == kEmitStage) {
        wave_sync();
#pragma unroll
        for (int t = 0; t < kNV; ++t) {
            const int n = 2 * (lane + 64 * t);
            const v2f e = *reinterpret_cast<const v2f*>(d.inv_env + n);
            *reinterpret_cast<v2f*>(sl + n) = mk2(snew[t][0] * e[0], snew[t][1] * e[1]);          // the rounded product the fold's fma takes (as below)
        }
        return;
    }
    const float sc = scale != nullptr ? scale[b] : 1.0f;
    if (EMIT == kEmitFrame) {
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
                // astype(int16): truncation; the pair goes out as ONE 4-byte store (the buffer may be host memory behind PCIe)
                const unsigned int pair = (unsigned int)(unsigned short)(short)c0 | ((unsigned int)(unsigned short)(short)c1 << 16);
                *reinterpret_cast<unsigned int*>(static_cast<short*>(hop_out) + b * (kNR / 2) + n) = pair;
            } else {
                *reinterpret_cast<v2f*>(static_cast<float*>(hop_out) + b * (kNR / 2) + n) = cur[t];
            }
        }
    }
}

}  // namespace dn
