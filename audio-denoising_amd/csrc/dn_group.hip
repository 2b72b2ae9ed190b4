// dn_group.hip -- hop groups: H consecutive hops of every stream in ONE launch, every Griffin-Lim chain whole (dn_pipe_set_group).
#include "dn_hop_common.hpp"

namespace dn {

// ---- hop groups: the whole Griffin-Lim chain of a frame inside ONE launch (dn_pipe_set_group; n_fft 1024)
// The deep pipe above keeps D hops of a stream in flight by cutting every chain into D segments, one per launch -- and pays for it at every
// launch boundary: the chains park in HBM (17.5 KB a stream and boundary, all CUs fetching theirs in the same microsecond at the head of the next
// launch: ~10 % of a launch, 17x the path's compulsory traffic).  Nothing but a launch boundary forces a chain off the chip, so a launch here is as
// long as a chain: it carries H consecutive new hops of every stream -- their front halves one after the other in the stream's front workgroup,
// hx handed from hop to hop -- beside the WHOLE chains of the H hops the previous launch fronted, one wavefront each.  A chain is never cut, nothing
// is parked, a frame's phases are drawn by its own chain wave; between launches only the slots travel (magnitudes, peak, meta: 6.5 KB a frame).
//   blocks [0, back_blocks)            the pending frames of spb = 4 / H streams: wavefront w runs the chain of the (w % per)-th oldest frame of stream
//                                      blockIdx.x * spb + w / per (per = 4 / spb) from its first iteration to its last
//   blocks [back_blocks, back_blocks + B)  P1-P10 of stream b's `group_hops` new hops, in order, into slots slot_next, slot_next + 1, ...
// Same arithmetic, same seeds (seed + frame index, stream id), same slots as the one-hop pipe: frames, hx and emitted hops are bit-identical to it.
// The price is granularity, not latency: input arrives H hops at a time and a frame is complete one launch (H hops) after its group was submitted.
#ifdef DN_PROBE
// diagnostic build only (tools/group_probe.py): the phases of front workgroup 0 for every hop of the group, start / end of every workgroup
static __device__ unsigned long long g_group_front[DN_PIPE_MAX_GROUP][4];
static __device__ unsigned long long g_group_wg[2048][2];
#define DN_GSTAMP(h, id) do { if (tid == 0 && (int)blockIdx.x == a.back_blocks) { unsigned long long t_; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); g_group_front[h][id] = t_; } } while (0)
#define DN_GWG(id) do { if (tid == 0 && blockIdx.x < 2048) { unsigned long long t_; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); g_group_wg[blockIdx.x][id] = t_; } } while (0)
#else
#define DN_GSTAMP(h, id) do { } while (0)
#define DN_GWG(id) do { } while (0)
#endif

template <int NFFT, bool STREAM, bool BF16, int CT>
__global__ __launch_bounds__(kHopPipeThreads, 2) void group_kernel(DspDev d, const DspDev* __restrict__ dp, const CellDev* __restrict__ cp, HopArgs a) {
    // n_fft 1024 only.  At 1536 a chain wave needs the state of three columns of 768 complex values: 420 registers.  Built in round 4 as a split group
    // (the chains as a launch of their own, one wave a SIMD, nothing spilled; the front halves as a second launch) and measured SLOWER than the one-hop
    // pipe's wavefront per column: 118.6 against 95.8 us per batch-256 hop, 443 against 370 us at 1,024 streams -- one wave runs six 768-point
    // transforms an iteration back to back (24 k ticks) where three column waves overlap theirs, and at 420 registers no front wave fits beside it.
    static_assert(NFFT == 1024, "whole chains run one wavefront per stream beside a front wave (dn_glw_body.hpp): n_fft 1024");
    constexpr int kNR = NFFT, kBins = Geo<NFFT>::kBins, kHop = kNR / 2;
    constexpr int kPairs = kHop / 2, kFoldN = (kPairs + kHopPipeThreads - 1) / kHopPipeThreads;   // sample pairs of half an overlap-add line, per thread (1 at n_fft 1024, 2 at 1536)
    __shared__ __attribute__((aligned(16))) char smem[hop_smem<NFFT>()];
    const int tid = threadIdx.x;
    const unsigned long long pushes = a.ctl->pushes, frames = a.ctl->frames;
    const unsigned int pending = a.ctl->pending, slot_next = a.ctl->slot_next;
    const SlotLayout sl(a.B, a.n_mels, kBins);
    // new hops that only fill the ring (the first n_fft / hop - 1 pushes of a stream, app3.py:174-178), the others are fronted
    int prime_hops = 0;
    if (STREAM) {
        const long long left = (long long)a.prime - (long long)pushes;
        prime_hops = left <= 0 ? 0 : left < (long long)a.group_hops ? (int)left : a.group_hops;
    }
    const int fronted = a.front_B > 0 ? a.group_hops - prime_hops : 0;
    DN_GWG(0);
    if ((int)blockIdx.x < a.back_blocks) {
      {
        // a workgroup holds a.spb streams x `per` pending frames of each (spb x per = 4): wave w runs frame w % per of stream blockIdx.x * spb + w / per
        const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int per = kGlwWaves / a.spb, sidx = wv / per, j = wv - sidx * per;
        const size_t b = (size_t)blockIdx.x * a.spb + sidx;
        DN_WSTAMP(0);
        const bool runs = j < (int)pending && b < (size_t)a.back_B;
        if (pending && !runs) {            // (the workgroup's window tables: every wave fills its share; a wave with a chain does it inside glw_body)
            glw_fill_tables<NFFT, kHopPipeThreads>(smem, d, tid);
            DN_LDS_BARRIER();
        }
        if (runs) {
            const int s = slot_behind(slot_next, (int)pending - j, a.n_slots);
            const float* slot = a.slots + (size_t)s * a.slot_stride;
            const uint32_t* meta = reinterpret_cast<const uint32_t*>(slot + sl.meta) + kSlotMeta * b;
            const v2f* init = meta[0] ? reinterpret_cast<const v2f*>(a.slot_init + (size_t)s * a.init_stride) : nullptr;
            const uint64_t seed = (uint64_t)meta[1] | ((uint64_t)meta[2] << 32);
            const uint64_t sid0 = (uint64_t)meta[3] | ((uint64_t)meta[4] << 32);
            const int n_iter = __builtin_amdgcn_readfirstlane((int)meta[6]);
            const float mom = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((int)meta[7]));
            float* gl_out = reinterpret_cast<float*>((uint64_t)meta[8] | ((uint64_t)meta[9] << 32));
#ifdef DN_GLW_PRIO
            __builtin_amdgcn_s_setprio(DN_GLW_PRIO);          // (experiment knob: measured below)
#endif
            glw_body<NFFT, STREAM ? kEmitStage : kEmitFrame>(smem, d, slot + sl.lin, init, seed, sid0, slot + sl.peak, gl_out, n_iter, mom, b, lane, wv,
                                                             nullptr, nullptr, 0, 0, -1, kGlwFresh, nullptr, tid);
            DN_WSTAMP(7);
        }
        if (STREAM) {
            // P12 for the frames of a stream in order: hop_out <- ola[:hop]; ola <- concat(ola[hop:], 0) + frame * peak (app3.py:217-224).  The chains
            // left frame x 1/envelope in their LDS lines; a thread owns sample pairs of either half of the line and keeps both in registers.
            __syncthreads();
            const int P = (int)pending;
            const int lead = a.filler_first ? a.group_out - P : 0;          // hops with no frame behind them: zero, as the reference's ola still is
            for (int si = 0; si < a.spb; ++si) {
                const size_t bs = (size_t)blockIdx.x * a.spb + si;
                if (bs >= (size_t)a.back_B) break;
                float* orow = a.ola + bs * kNR;
                v2f lo[kFoldN], hi[kFoldN];
#pragma unroll
                for (int r = 0; r < kFoldN; ++r) {
                    const int m = tid + kHopPipeThreads * r;
                    if (m < kPairs) { lo[r] = *reinterpret_cast<const v2f*>(orow + 2 * m); hi[r] = *reinterpret_cast<const v2f*>(orow + 2 * m + kHop); }
                }
                for (int i = 0; i < a.group_out; ++i) {
                    const int f = i - lead;
                    const bool has = f >= 0 && f < P;
                    const float* line = glw_signal_line<NFFT>(smem, si * per + (has ? f : 0));
                    const float sc = has ? (a.slots + (size_t)slot_behind(slot_next, P - f, a.n_slots) * a.slot_stride + sl.peak)[bs] : 0.0f;
#pragma unroll
                    for (int r = 0; r < kFoldN; ++r) {
                        const int m = tid + kHopPipeThreads * r;
                        if (m >= kPairs) continue;
                        v2f emit = mk2(0.0f, 0.0f);
                        if (has) {
                            emit = lo[r];
                            const v2f p0 = *reinterpret_cast<const v2f*>(line + 2 * m), p1 = *reinterpret_cast<const v2f*>(line + 2 * m + kHop);
                            lo[r] = mk2(fmaf(p0[0], sc, hi[r][0]), fmaf(p0[1], sc, hi[r][1]));
                            hi[r] = mk2(fmaf(p1[0], sc, 0.0f), fmaf(p1[1], sc, 0.0f));
                        }
                        const size_t at = (size_t)i * (size_t)a.hop_out_stride + bs * kHop + 2 * m;
                        if (a.out_s16) {
                            const float c0 = fminf(fmaxf(emit[0], -1.0f), 1.0f) * 32767.0f, c1 = fminf(fmaxf(emit[1], -1.0f), 1.0f) * 32767.0f;   // app3.py:244-245
                            *reinterpret_cast<unsigned int*>(static_cast<short*>(a.hop_out) + at) = (unsigned int)(unsigned short)(short)c0 | ((unsigned int)(unsigned short)(short)c1 << 16);
                        } else {
                            *reinterpret_cast<v2f*>(static_cast<float*>(a.hop_out) + at) = emit;
                        }
                    }
                }
                if (P > 0) {
#pragma unroll
                    for (int r = 0; r < kFoldN; ++r) {
                        const int m = tid + kHopPipeThreads * r;
                        if (m < kPairs) { *reinterpret_cast<v2f*>(orow + 2 * m) = lo[r]; *reinterpret_cast<v2f*>(orow + 2 * m + kHop) = hi[r]; }
                    }
                }
            }
        }
      }
    } else {
      {
        const size_t b = blockIdx.x - a.back_blocks;
#pragma unroll 1
        for (int h = 0; h < a.group_hops; ++h) {
            // One hop of the loop is three large straight-line stages.  Everything they read through the plan's pointers is loop-invariant, and hoisted
            // out of the loop it would all be live at once (every table pointer, the lane's twiddles, weight fragments: ~190 VGPRs and ~300 SGPRs of
            // spills): the pointers of each hop are rebased by a zero the compiler cannot see through, so each stage loads what it needs where it
            // needs it, as in the single-hop kernels.
            int z;
            DN_OPAQUE_ZERO(z);
            const DspDev& dz = dp[z];
            const CellDev& cz = cp[z];
            const int tidz = tid + z;
            const float* frames_in = STREAM ? a.ring : a.frames + (size_t)h * (size_t)a.frames_stride;
            if (STREAM) {
                const char* in = static_cast<const char*>(a.hop_in) + (size_t)h * (size_t)a.hop_in_stride * (a.in_s16 ? 2 : 4);
                ring_shift<NFFT, kHopPipeThreads>(a.ring, in, a.in_s16, b, tid);
            }
            if (h < prime_hops) continue;
            const int q = h - prime_hops;
            int s = (int)slot_next + q;
            if (s >= a.n_slots) s -= a.n_slots;
            float* slot = a.slots + (size_t)s * a.slot_stride;
            DN_GSTAMP(q, 0);
            stft_body<NFFT, false, true, kHopPipeThreads>(smem, dz, frames_in, nullptr, slot, slot + sl.peak, DN_PEAK_NORMALIZE | DN_PRE_WINDOW, b, tidz);   // P1-P6
            __syncthreads();
            DN_GSTAMP(q, 1);
            cell_body<kHopPipeThreads / 64, BF16, CT>(smem, cz, slot, a.hx + z, slot + sl.diff, a.hx + z, 3, a.C, b, tidz);                               // P7
            __syncthreads();
            DN_GSTAMP(q, 2);
            invmel_body<NFFT, true, kHopPipeThreads>(smem, dz, slot, slot + sl.diff, slot + sl.lin, 3 * a.B, b * 3, tidz);                              // P8-P10
            DN_GSTAMP(q, 3);
            if (tid == 0) {
                uint32_t* meta = reinterpret_cast<uint32_t*>(slot + sl.meta) + kSlotMeta * b;
                const uint64_t seed = a.seed + frames + (unsigned long long)q;
                meta[0] = a.init_in != nullptr ? 1u : 0u;
                meta[1] = (uint32_t)seed;
                meta[2] = (uint32_t)(seed >> 32);
                meta[3] = (uint32_t)a.sid0;
                meta[4] = (uint32_t)(a.sid0 >> 32);
                meta[5] = 0u;
                meta[6] = (uint32_t)a.n_iter;
                meta[7] = __builtin_bit_cast(uint32_t, a.mom);
                const uint64_t dst = reinterpret_cast<uint64_t>(STREAM ? nullptr : a.gl_out + (size_t)h * (size_t)a.out_stride);
                meta[8] = (uint32_t)dst;
                meta[9] = (uint32_t)(dst >> 32);
            }
            if (a.init_in != nullptr) {
                const float2* src = reinterpret_cast<const float2*>(a.init_in + (size_t)h * (size_t)a.init_in_stride) + b * 3 * kBins;
                float2* dst = a.slot_init + (size_t)s * a.init_stride + b * 3 * kBins;
                for (int i = tid; i < 3 * kBins; i += kHopPipeThreads) dst[i] = src[i];
            }
            __syncthreads();          // the next hop reuses the LDS stages and reads the hx this one stored
        }
      }
    }
    // ---- ticket: the last workgroup of the launch advances the control block (every workgroup has read it by then)
    __syncthreads();
    DN_GWG(1);
    if (tid == 0) {
        const unsigned int t = atomicAdd(&a.ctl->done, 1u);
        if (t == gridDim.x - 1) {
            a.ctl->done = 0;
            a.ctl->launches = a.ctl->launches + 1;
            if (a.front_B > 0) a.ctl->pushes = pushes + (unsigned long long)a.group_hops;
            a.ctl->frames = frames + (unsigned long long)fronted;
            unsigned int sn = slot_next + (unsigned int)fronted;
            if (sn >= (unsigned int)a.n_slots) sn -= (unsigned int)a.n_slots;
            a.ctl->slot_next = sn;
            a.ctl->pending = (unsigned int)fronted;          // every pending chain ran to its end in this launch
        }
    }
}

void launch_group(const DspDev& d, const CellDev& c, const HopArgs& a, bool bf16, hipStream_t st) {
    const dim3 grid(a.back_blocks + a.front_B), block(kHopPipeThreads);
    const bool stream = a.ola != nullptr, usual = a.C == 5;
    auto go = [&](auto k) { hipLaunchKernelGGL(k, grid, block, 0, st, d, a.d_dev, a.c_dev, a); };
    if (stream) {
        if (usual) { if (bf16) go(group_kernel<1024, true, true, 5>); else go(group_kernel<1024, true, false, 5>); }
        else { if (bf16) go(group_kernel<1024, true, true, 0>); else go(group_kernel<1024, true, false, 0>); }
    } else {
        if (usual) { if (bf16) go(group_kernel<1024, false, true, 5>); else go(group_kernel<1024, false, false, 5>); }
        else { if (bf16) go(group_kernel<1024, false, true, 0>); else go(group_kernel<1024, false, false, 0>); }
    }
}

}  // namespace dn

#ifdef DN_PROBE
extern "C" int dn_probe_read_group_glw(unsigned long long* host32) {
    return (int)hipMemcpyFromSymbol(host32, HIP_SYMBOL(dn::g_glw_probe), sizeof(dn::g_glw_probe));
}
extern "C" int dn_probe_read_group_front(unsigned long long* host16) {
    return (int)hipMemcpyFromSymbol(host16, HIP_SYMBOL(dn::g_group_front), sizeof(dn::g_group_front));
}
extern "C" int dn_probe_read_group_wg(unsigned long long* host4096) {
    return (int)hipMemcpyFromSymbol(host4096, HIP_SYMBOL(dn::g_group_wg), sizeof(dn::g_group_wg));
}
#endif
