// dn_hop.hip -- the whole hop as ONE launch (app3.py:178-226 for B streams), in two schedules:
//
// frame_kernel   nothing overlapped: one workgroup per stream runs P1-P12 back to back (dn_process_frame, dn_stream_step).
//
// hop_kernel     software-pipelined: ONE launch per hop that overlaps hop n's Griffin-Lim with hop n+1's analysis + model +
//                inverse mel (consecutive loop iterations of app3.py:178 overlapped; dn_pipe_*).
//   Consecutive hops depend on each other only through hx (the model); hop n's Griffin-Lim (~3/4 of a hop, a
//   strictly serial chain per stream that occupies three wavefronts of a CU) does not depend on hop n+1's
//   front half.  A launch therefore carries two kinds of workgroups:
//     blocks [0, back_B)           Griffin-Lim of the PENDING hop, reading scratch slot (frames-1) & 1
//     blocks [back_B, back_B + B)  P1-P10 of THIS hop (stft -> GRUUNet2 -> inverse mel, one stream per workgroup,
//                                  stages separated by workgroup barriers), writing scratch slot frames & 1
//   Both kinds are resident together (192 threads, <= 35 KB LDS: a Griffin-Lim and a front workgroup share a CU),
//   so the front half fills issue slots and the fourth SIMD the latency-bound Griffin-Lim leaves idle.  Everything is on
//   the caller's stream: launch k+1 is ordered behind launch k, which is all the synchronisation the slot hand-over
//   needs -- no events, no second stream (a two-stream/event version lost ~13 us per hop to cross-queue signalling).
//   What changes from hop to hop -- the slot parity, the Griffin-Lim seed, whether a hop is pending, the ring priming
//   of a new stream -- lives in a device-resident control block (PipeCtl) that the last workgroup of every launch
//   advances, so one captured launch replays indefinitely under hipGraph (BASELINE config 5).
#include "dn_hop_common.hpp"

// The kernels below are instantiated in THREE translation units, because LLVM's GCN scheduling strategies suit them differently (Makefile; measured,
// profiles/r04_group_sweep.txt):
//   dn_hop.hip      (this file)  n_fft 1024, a wavefront per STFT column (the one-hop pipe, the unpipelined hop): max-ILP, +2..3 %;
//   dn_hop_glw.hip  (DN_HOP_TU_GLW)   n_fft 1024, a wavefront per stream (deep pipes, the saturated regime, the front-only launch of a split hop):
//                                     iterative-ILP, +3..7 % (1,024 streams 6.73 -> 7.00 M frames/s, the captured streaming step 6.34 -> 6.79 M);
//   dn_hop1536.hip  (DN_HOP_TU_1536)  n_fft 1536: the default strategy (max-ILP costs it 12 %: it sits at the 256-register cap and spills more).
// The stamped diagnostic build keeps everything in this file (its probe arrays are per translation unit).
#if !defined(DN_PROBE)
#define DN_HOP_SPLIT_TUS 1
#endif

namespace dn {

// n_fft 1536: the Griffin-Lim body would take 330 registers and shut the front workgroup out of the CU; capping the
// kernel at two waves per SIMD (256 registers, ~80 values spilled to scratch) keeps both halves resident (+30 % at 1024
// streams).  n_fft 1024 fits in 229 registers without a cap (capping it costs 8 %).
// The pipelined launch uses workgroups of FOUR wavefronts.  A Griffin-Lim workgroup needs three (one per column): its fourth exits at
// once.  A front workgroup uses all four: its waves share SIMDs with the Griffin-Lim waves of the same CU (six or seven waves on four
// SIMDs) and its phases end at workgroup barriers, so the front half -- not the Griffin-Lim chain -- was what ended a batch-256 launch
// (measured: two iterations moved INTO the front workgroup cost 4.8 us); a fourth wave shortens every conv phase by a quarter.
#ifdef DN_PROBE
// diagnostic build only: when the two kinds of workgroups of one launch start and finish (workgroups 0 and back_B of the last launch)
static __device__ unsigned long long g_hop_wg_probe[8];
static __device__ unsigned long long g_hop_blk_t[2048][2];   // start / end of every workgroup of the last launch (s_memtime)
static __device__ unsigned int g_hop_blk_hw[2048];          // where every workgroup of the last launch ran: HW_ID | XCC_ID << 28 (tools/glw_probe.py census)
#define DN_HSTAMP(id) do { if (tid == 0 && (blockIdx.x == 0 || (int)blockIdx.x == a.back_blocks)) { unsigned long long t_; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); g_hop_wg_probe[id] = t_; } } while (0)
#else
#define DN_HSTAMP(id) do { } while (0)
#endif

// CT: the number of compressed mel bins when the plan has the usual one (80 mels at n_fft 1024, 64 at 1536), 0 = any (run-time lengths in the model)
// GLW: the pending hops' Griffin-Lim runs one wavefront per stream and chain segment (dn_glw_body.hpp) instead of one wavefront per column (one
//      stream a workgroup, one pending hop).  A workgroup then holds a.spb streams x a.depth chain segments (spb x depth <= 4):
//        depth 1  four streams a workgroup: the saturated regime (several streams per CU);
//        depth D  a stream's chain is cut into D segments that run in D consecutive launches, so D hops of the SAME stream are in flight
//                 (wave j of the workgroup advances frame frames-1-j by its next segment and parks it in HBM, the last segment emits): with
//                 about one stream per CU this is what fills the CU -- throughput of the saturated regime at batch 256 for D - 1 more
//                 hops of latency.
//      Capped at two waves per SIMD.
// FRONT: a launch of front workgroups only, the second of the TWO launches of a split hop (a.front_only; the first is this kernel's
//      ordinary form with front_B = 0: the chains).  Where a launch carries many more workgroups than the chip holds at once, its Griffin-Lim
//      workgroups (the low block ids) run first and its front workgroups after them anyway -- two phases, not a mix -- and a front workgroup
//      compiled beside the chain inherits the chain's 243 registers: two workgroups a CU.  On its own the front half is capped at
//      kFrontPerCu workgroups a CU (its phases end at workgroup barriers and wait on memory: more of them in flight is what it wants).
template <int NFFT, bool STREAM, bool BF16, int CT, bool GLW, bool FRONT = false>
__global__ __launch_bounds__(kHopPipeThreads, FRONT ? kFrontPerCu : (NFFT == 1536 || GLW) ? 2 : 1) void hop_kernel(DspDev d, CellDev cd, HopArgs a) {
    constexpr int kNR = NFFT, kBins = Geo<NFFT>::kBins;
    __shared__ __attribute__((aligned(16))) char smem[FRONT ? front_smem<NFFT>() : hop_smem<NFFT>()];
    const int tid = threadIdx.x;
    // control block as the previous launch left it (uniform: scalar loads)
    const unsigned long long pushes = a.ctl->pushes, frames = a.ctl->frames, launches = a.ctl->launches;
    const unsigned int pending = a.ctl->pending, slot_next = a.ctl->slot_next;
    const SlotLayout sl(a.B, d.n_mels, kBins);
    const bool priming = STREAM && pushes < (unsigned long long)a.prime;
#ifdef DN_PROBE
    if (tid == 0 && blockIdx.x < 2048) {
        unsigned int hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_hop_blk_hw[blockIdx.x] = (hw & 0x0fffffffu) | (xcc << 28);
        unsigned long long t_;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
        g_hop_blk_t[blockIdx.x][0] = t_;
    }
#endif
    // the oldest frame in flight completes in this launch when this is its last segment (wavefront per column: depth 1, always)
    const int depth = GLW ? a.depth : 1;
    const bool completes = !FRONT && pending > 0 && launches - a.ctl->front_launch[(frames - pending) & 7] == (unsigned long long)depth;
    if (!FRONT && GLW && (int)blockIdx.x < a.back_blocks) {
        if constexpr (GLW && !FRONT) {
            const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
            const int sidx = wv / depth, j = wv - sidx * depth;          // stream of the workgroup, frames behind the newest (uniform)
            const size_t b = (size_t)blockIdx.x * a.spb + sidx;
            DN_WSTAMP(0);
            // the workgroup's window tables: every wave fills its share and meets the others at ONE LDS-only barrier -- a wave that runs a chain
            // does both inside glw_body, behind the loads of its own prologue; the others here
            const bool runs = pending && sidx < a.spb && b < (size_t)a.back_B && j < (int)pending;
            if (pending && !runs) {
                glw_fill_tables<NFFT, kHopPipeThreads>(smem, d, tid);
                DN_LDS_BARRIER();
            }
            DN_WSTAMP(1);
            if (sidx < a.spb && b < (size_t)a.back_B) {          // (a wave without work skips to the ticket: wave 0 always has a stream)
                if (j < (int)pending) {
                    const unsigned long long g = frames - 1 - j;                                          // the frame
                    const int seg = (int)(launches - a.ctl->front_launch[g & 7]) - 1;                     // its next segment: 0 .. depth-1
                    const int s = slot_behind(slot_next, 1 + j, a.n_slots);
                    const float* slot = a.slots + (size_t)s * a.slot_stride;
                    const uint32_t* meta = reinterpret_cast<const uint32_t*>(slot + sl.meta) + kSlotMeta * b;
                    const v2f* init = meta[0] ? reinterpret_cast<const v2f*>(a.slot_init + (size_t)s * a.init_stride) : nullptr;
                    const uint64_t seed = (uint64_t)meta[1] | ((uint64_t)meta[2] << 32);
                    const uint64_t sid0 = (uint64_t)meta[3] | ((uint64_t)meta[4] << 32);
                    // (the slot words come back through the vector memory path: readfirstlane tells the compiler they are wave-uniform, so the
                    // iteration loop keeps a scalar trip count and a scalar branch)
                    const int it0 = __builtin_amdgcn_readfirstlane((int)meta[5]), n_iter = __builtin_amdgcn_readfirstlane((int)meta[6]);
                    const float mom = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((int)meta[7]));
                    float* gl_out = reinterpret_cast<float*>((uint64_t)meta[8] | ((uint64_t)meta[9] << 32));
                    // iterations [lo, hi) of the frame's chain; the last segment runs to the end and emits.  (A frame of a deep pipe comes with its
                    // initial phases in the slot -- injected, or drawn by its front workgroup's spare wave -- so every segment starts alike.)
                    const int span = n_iter > it0 ? n_iter - it0 : 0;
                    auto cut = [&](int k) { return k <= 0 ? 0 : k >= depth ? span : (span * k + depth / 2) / depth; };
                    const int lo = it0 + cut(seg), hi = it0 + cut(seg + 1);
                    const bool last = seg == depth - 1;
#ifdef DN_GLW_PRIO
                    __builtin_amdgcn_s_setprio(DN_GLW_PRIO);
#endif
                    glw_body<NFFT, STREAM>(smem, d, slot + sl.lin, init, seed, sid0, slot + sl.peak, STREAM ? nullptr : gl_out, n_iter, mom, b, lane, wv,
                                           a.ola, a.hop_out, a.out_s16, lo, last ? -1 : hi, seg > 0 ? kGlwFromSeg : it0 > 0 ? kGlwFromX : kGlwFresh,
                                           reinterpret_cast<v2f*>(a.gl_state + (size_t)s * a.state_stride), tid);
                    DN_WSTAMP(7);
                }
                if (STREAM && j == 0 && !completes) {
                    // nothing to emit in this launch: the reference's ola[:hop] is still zero (app3.py:133,219)
                    for (int n = lane; n < kNR / 2; n += 64) {
                        if (a.out_s16) static_cast<short*>(a.hop_out)[b * (kNR / 2) + n] = 0;
                        else static_cast<float*>(a.hop_out)[b * (kNR / 2) + n] = 0.0f;
                    }
                }
            }
        }
    } else if (!FRONT && (int)blockIdx.x < a.back_blocks) {
      if constexpr (!FRONT) {
        const size_t b = blockIdx.x;
        if (tid >= kHopThreads) return;         // (before any barrier: a terminated wave no longer counts at s_barrier)
        DN_HSTAMP(0);
        if (pending) {
            // the Griffin-Lim chain is the critical path of the launch: let its waves win issue arbitration against the
            // front-half waves they share SIMDs with
            __builtin_amdgcn_s_setprio(DN_GL_PRIO);
            const int s = slot_behind(slot_next, 1, a.n_slots);
            const float* slot = a.slots + (size_t)s * a.slot_stride;
            const uint32_t* meta = reinterpret_cast<const uint32_t*>(slot + sl.meta) + kSlotMeta * b;
            const v2f* init = meta[0] ? reinterpret_cast<const v2f*>(a.slot_init + (size_t)s * a.init_stride) : nullptr;
            const uint64_t seed = (uint64_t)meta[1] | ((uint64_t)meta[2] << 32);
            const uint64_t sid0 = (uint64_t)meta[3] | ((uint64_t)meta[4] << 32);
            // (the slot words come back through the vector memory path: readfirstlane tells the compiler they are wave-uniform, so the iteration
            // loop keeps a scalar trip count and a scalar branch)
            const int it0 = __builtin_amdgcn_readfirstlane((int)meta[5]);          // iterations the frame's front workgroup already ran (head start)
            // the frame's own n_iter / momentum (those of its submit, not of this call: it0 <= n_iter by construction) and destination
            const int n_iter = __builtin_amdgcn_readfirstlane((int)meta[6]);
            const float mom = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((int)meta[7]));
            v2f* st = reinterpret_cast<v2f*>(a.gl_state + (size_t)s * a.state_stride);
            if (!STREAM) {
                float* gl_out = reinterpret_cast<float*>((uint64_t)meta[8] | ((uint64_t)meta[9] << 32));
                gl_body<NFFT, false, false>(smem, d, slot + sl.lin, nullptr, init, seed, sid0, slot + sl.peak, gl_out, n_iter, mom, b, tid,
                                            nullptr, nullptr, 0, it0, -1, st);
            } else
                gl_body<NFFT, false, true>(smem, d, slot + sl.lin, nullptr, init, seed, sid0, slot + sl.peak, nullptr, n_iter, mom, b, tid,
                                           a.ola, a.hop_out, a.out_s16, it0, -1, st);
            __builtin_amdgcn_s_setprio(0);
            DN_HSTAMP(1);
        } else if (STREAM) {
            // nothing to emit yet: the reference's ola[:hop] is still zero (app3.py:133,219)
            for (int n = tid; n < kNR / 2; n += kHopThreads) {      // (three waves are left)
                if (a.out_s16) static_cast<short*>(a.hop_out)[b * (kNR / 2) + n] = 0;
                else static_cast<float*>(a.hop_out)[b * (kNR / 2) + n] = 0.0f;
            }
        }
      }
    } else {
        const size_t b = blockIdx.x - (FRONT ? 0 : a.back_blocks);
        DN_HSTAMP(2);
        const float* frames_in = a.frames;
        if (STREAM) {
            if (__builtin_expect(a.host_copy_src != nullptr, 0)) {
                // deferred host output: the hop the PREVIOUS push emitted for this stream, device staging buffer -> its page-locked host buffer.
                // The front workgroups of a launch start all through it, so the posted stores drain while the hop computes instead of ending the
                // previous launch as one PCIe burst; the last workgroup's system-scope fence below covers them.  (Here and not at the head of the
                // kernel: there the same lines cost the chain waves of EVERY streaming launch 2 % -- 155.8 -> 159.2 us at 1,024 streams.)
                const unsigned int per = a.host_copy_n16 / (unsigned int)a.B;          // 16-byte units of one stream's hop
                for (unsigned int i = tid; i < per; i += kHopPipeThreads) a.host_copy_dst[b * per + i] = a.host_copy_src[b * per + i];
            }
            ring_shift<NFFT, kHopPipeThreads>(a.ring, a.hop_in, a.in_s16, b, tid);
            frames_in = a.ring;
        }
        if (!priming) {
            const int s = (int)slot_next;
            float* slot = a.slots + (size_t)s * a.slot_stride;
            float2* slot_init = a.slot_init + (size_t)s * a.init_stride;
            stft_body<NFFT, false, true, kHopPipeThreads>(smem, d, frames_in, nullptr, slot, slot + sl.peak, DN_PEAK_NORMALIZE | DN_PRE_WINDOW, b, tid);   // P1-P6
            const int split = FRONT ? 0 : min(a.gl_split, a.n_iter);        // (a split hop has no head start)
            // (measured: n_fft 1536, 13 bins a lane: 106 -> 100 us per batch-256 hop; n_fft 1024, 9 bins a lane: 54.7 -> 55.2 us -- the draw's 12 KB
            // a stream through HBM cost more than the multiplies it moved off the chain, so only the long transform uses it)
            // A deep pipe (a.depth > 1: the chain runs as segments, one wavefront per stream) draws here too, at either transform length: there the
            // draw is 27 Philox blocks a lane (~12 k cycles) on the first segment's wave, which then has to get an iteration less than the others.
            const bool draw = a.init_in == nullptr && (NFFT == 1536 ? split > 0 : a.depth > 1);
            if (draw && tid >= kHopThreads) {
                // The fourth wave has no column to transform and would wait here: it draws the random initial phases the head start
                // below begins with (the same Philox blocks, so the same bits) into the slot.  A block is ten rounds of quarter-rate integer
                // multiplies and serves one bin pair -- drawn by the head start itself they were ~7 k cycles on the launch's critical path.
                float2* dst = slot_init + b * 3 * kBins;
                constexpr int kPairs = (kBins - 1) / 2 + 1;                 // one block per bin pair (m, NC - m), m = 0..NC/2
                for (int i = tid - kHopThreads; i < 3 * kPairs; i += 64) {
                    const int col = i / kPairs, m = i - col * kPairs;
                    v2f lo, hi;
                    rand_angle_pair(a.seed + frames, a.sid0 + b, col, m, lo, hi);
                    dst[col * kBins + m] = make_float2(lo[0], lo[1]);
                    if (2 * m != kBins - 1) dst[col * kBins + kBins - 1 - m] = make_float2(hi[0], hi[1]);
                }
            }
            __syncthreads();
            DN_HSTAMP(5);
            cell_body<kHopPipeThreads / 64, BF16, CT, FRONT ? false : kCellStageDefault>(smem, cd, slot, a.hx, slot + sl.diff, a.hx, 3, a.C, b, tid);   // P7
            __syncthreads();
            DN_HSTAMP(6);
            invmel_body<NFFT, true, kHopPipeThreads>(smem, d, slot, slot + sl.diff, slot + sl.lin, 3 * a.B, b * 3, tid);                  // P8-P10
            // what this frame's Griffin-Lim (next launch) needs besides the magnitudes: its seed, its stream ids and, in parity mode, its phases
            DN_HSTAMP(3);                          // front half (P1-P10) done
            if (tid == 0) {
                uint32_t* meta = reinterpret_cast<uint32_t*>(slot + sl.meta) + kSlotMeta * b;
                const uint64_t seed = a.seed + frames;
                meta[0] = (a.init_in != nullptr || (draw && a.depth > 1)) ? 1u : 0u;       // the frame's phases are in the slot
                meta[1] = (uint32_t)seed;
                meta[2] = (uint32_t)(seed >> 32);
                meta[3] = (uint32_t)a.sid0;
                meta[4] = (uint32_t)(a.sid0 >> 32);
                meta[5] = (uint32_t)split;
                meta[6] = (uint32_t)a.n_iter;
                meta[7] = __builtin_bit_cast(uint32_t, a.mom);
                const uint64_t dst = reinterpret_cast<uint64_t>(a.gl_out);
                meta[8] = (uint32_t)dst;
                meta[9] = (uint32_t)(dst >> 32);
            }
            if (a.init_in != nullptr) {
                const float2* src = reinterpret_cast<const float2*>(a.init_in) + b * 3 * kBins;
                float2* dst = slot_init + b * 3 * kBins;
                for (int i = tid; i < 3 * kBins; i += kHopPipeThreads) dst[i] = src[i];
            }
            if constexpr (!FRONT) if (split > 0) {
                // head start: this workgroup would idle for the rest of the launch (the pending hop's chain is ~1.5x longer than P1-P10)
                __syncthreads();                       // the magnitudes are in the slot
                if (tid >= kHopThreads) return;        // the chain is three waves wide
                __builtin_amdgcn_s_setprio(DN_HS_PRIO); // below the pending hop's chain (3): that one ends the launch
                gl_body<NFFT, false, false, NFFT == 1536>(smem, d, slot + sl.lin, nullptr,
                                            draw ? reinterpret_cast<const v2f*>(slot_init) : reinterpret_cast<const v2f*>(a.init_in), a.seed + frames, a.sid0,
                                            nullptr, nullptr, a.n_iter, a.mom, b, tid, nullptr, nullptr, 0, 0, split,
                                            reinterpret_cast<v2f*>(a.gl_state + (size_t)s * a.state_stride));
                __builtin_amdgcn_s_setprio(0);
                DN_HSTAMP(4);                      // head start done
            }
        }
    }
    // ---- ticket: the last workgroup of the launch advances the control block (every workgroup has read it by then)
    // Zero-copy host transport: the completion word the last workgroup publishes must not overtake ANY workgroup's samples.  A workgroup barrier
    // orders LDS and (in this wave-sharing mode) does not have to wait for outstanding stores: every wave therefore waits until its own stores are
    // ACKNOWLEDGED (s_waitcnt vmcnt(0)) before the barrier that precedes its workgroup's ticket.  The samples go to page-locked host memory, which
    // the L2s do not cache: an acknowledged store has left the chip's caches for the fabric, and the completion word -- written by the last
    // workgroup after it has seen every ticket, behind a system-scope fence -- follows them through the same PCIe port, where posted writes of one
    // device stay in order.  The strict form of the memory model -- the ticket as a release / acquire pair at SYSTEM scope -- is kept behind
    // DN_HOST_STRICT_RELEASE: it writes an XCD's whole L2 back once per workgroup (the slots of the pipe are in there) and measured 10-14 % of the
    // host-fed rate (256 streams 59.5 -> 68.0 us per hop, 1,024 streams 163.6 -> 186.5: profiles/r04_v4_side_measurements.jsonl) for stores that
    // are not in the L2 in the first place.
    if (a.host_done != nullptr) DN_WAIT_VMEM();
    __syncthreads();
#ifdef DN_PROBE
    if (tid == 0 && blockIdx.x < 2048) {
        unsigned long long t_;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
        g_hop_blk_t[blockIdx.x][1] = t_;
    }
#endif
    if (tid == 0) {
#ifdef DN_HOST_STRICT_RELEASE
        const unsigned int t = a.host_done != nullptr ? __hip_atomic_fetch_add(&a.ctl->done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_SYSTEM)
                                                      : atomicAdd(&a.ctl->done, 1u);
#else
        const unsigned int t = atomicAdd(&a.ctl->done, 1u);
#endif
        if (t == gridDim.x - 1) {
            a.ctl->done = 0;
            if (!FRONT) a.ctl->launches = launches + 1;     // (the chains' launch of a split hop has counted it: this one is launch `launches - 1` still)
            const bool fronted = a.front_B > 0 && !priming;
            if (a.front_B > 0) a.ctl->pushes = pushes + 1;
            if (fronted) {
                a.ctl->front_launch[frames & 7] = FRONT ? launches - 1 : launches;
                a.ctl->frames = frames + 1;
                a.ctl->slot_next = (int)slot_next + 1 == a.n_slots ? 0u : slot_next + 1;
            }
            a.ctl->pending = pending - (completes ? 1u : 0u) + (fronted ? 1u : 0u);
            if (a.host_done != nullptr) {
                __threadfence_system();
                __hip_atomic_store(a.host_done, a.host_done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

template <int NFFT, bool STREAM, bool GLW>
static void launch_hop_n(const DspDev& d, const CellDev& c, const HopArgs& a, bool bf16, hipStream_t st) {
    const dim3 block(kHopPipeThreads);
    constexpr int kUsualC = NFFT == 1536 ? 4 : 5;
    if constexpr (GLW) {
        if (a.front_only) {             // the second launch of a split hop: front workgroups alone, under their own register budget
            const dim3 fgrid(a.front_B);
            if (a.C == kUsualC) {
                if (bf16) hipLaunchKernelGGL((hop_kernel<NFFT, STREAM, true, kUsualC, true, true>), fgrid, block, 0, st, d, c, a);
                else hipLaunchKernelGGL((hop_kernel<NFFT, STREAM, false, kUsualC, true, true>), fgrid, block, 0, st, d, c, a);
            } else {
                if (bf16) hipLaunchKernelGGL((hop_kernel<NFFT, STREAM, true, 0, true, true>), fgrid, block, 0, st, d, c, a);
                else hipLaunchKernelGGL((hop_kernel<NFFT, STREAM, false, 0, true, true>), fgrid, block, 0, st, d, c, a);
            }
            return;
        }
    }
    const dim3 grid(a.back_blocks + a.front_B);
    if (a.C == kUsualC) {
        if (bf16) hipLaunchKernelGGL((hop_kernel<NFFT, STREAM, true, kUsualC, GLW>), grid, block, 0, st, d, c, a);
        else hipLaunchKernelGGL((hop_kernel<NFFT, STREAM, false, kUsualC, GLW>), grid, block, 0, st, d, c, a);
    } else {
        if (bf16) hipLaunchKernelGGL((hop_kernel<NFFT, STREAM, true, 0, GLW>), grid, block, 0, st, d, c, a);
        else hipLaunchKernelGGL((hop_kernel<NFFT, STREAM, false, 0, GLW>), grid, block, 0, st, d, c, a);
    }
}

#if !defined(DN_HOP_TU_1536) && !defined(DN_HOP_TU_GLW)
// a.glw: the caller laid the grid out for a wavefront per stream and chain segment (n_fft 1024 only) instead of a wavefront per column
// The deferred host output of the LAST push (no further launch will carry it): copy, then publish (dn_pipe_stream_host_wait).
__global__ void host_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, unsigned int n16) {
    for (unsigned int u = blockIdx.x * blockDim.x + threadIdx.x; u < n16; u += gridDim.x * blockDim.x) dst[u] = src[u];
}
__global__ void host_publish_kernel(unsigned long long* done, unsigned long long value) {
    __threadfence_system();
    __hip_atomic_store(done, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void launch_host_copy(const uint4* src, uint4* dst, unsigned int n16, unsigned long long* done, unsigned long long value, hipStream_t st) {
    const unsigned int blocks = (n16 + 255) / 256;
    hipLaunchKernelGGL(host_copy_kernel, dim3(blocks < 1024 ? (blocks ? blocks : 1) : 1024), dim3(256), 0, st, src, dst, n16);
    hipLaunchKernelGGL(host_publish_kernel, dim3(1), dim3(1), 0, st, done, value);      // (stream order: the copy has completed)
}

void launch_hop_1536(const DspDev& d, const CellDev& c, const HopArgs& a, bool bf16, hipStream_t st);          // (dn_hop1536.hip)
void launch_hop_glw(const DspDev& d, const CellDev& c, const HopArgs& a, bool bf16, hipStream_t st);           // (dn_hop_glw.hip)
void launch_frame_1536(const DspDev& d, const CellDev& c, const FrameArgs& a, int B, bool bf16, hipStream_t st);

void launch_hop(const DspDev& d, const CellDev& c, const HopArgs& a, bool bf16, hipStream_t st) {
    const bool stream = a.ola != nullptr;
    if (a.split && a.glw && a.front_B > 0 && d.n_fft == 1024) {
        // a split hop: the chains of the hops in flight first (this launch is what a flush launch is), then the new hop's front halves
        HopArgs chains = a, fronts = a;
        chains.split = 0; chains.front_B = 0;
        chains.host_done = nullptr; chains.host_copy_src = nullptr;          // (the transport belongs to the launch that ends the hop)
        fronts.split = 0; fronts.front_only = 1; fronts.back_blocks = 0; fronts.gl_split = 0;
        launch_hop(d, c, chains, bf16, st);
        launch_hop(d, c, fronts, bf16, st);
        return;
    }
    if (d.n_fft == 1536) {
#ifdef DN_HOP_SPLIT_TUS
        launch_hop_1536(d, c, a, bf16, st);
#else
        if (stream) launch_hop_n<1536, true, false>(d, c, a, bf16, st);
        else launch_hop_n<1536, false, false>(d, c, a, bf16, st);
#endif
    } else if (a.glw) {
#ifdef DN_HOP_SPLIT_TUS
        launch_hop_glw(d, c, a, bf16, st);
#else
        if (stream) launch_hop_n<1024, true, true>(d, c, a, bf16, st);
        else launch_hop_n<1024, false, true>(d, c, a, bf16, st);
#endif
    } else {
        if (stream) launch_hop_n<1024, true, false>(d, c, a, bf16, st);
        else launch_hop_n<1024, false, false>(d, c, a, bf16, st);
    }
}

__global__ void ctl_set_kernel(PipeCtl* ctl, unsigned long long pushes, unsigned long long frames, unsigned int pending) {
    ctl->pushes = pushes; ctl->frames = frames; ctl->pending = pending; ctl->done = 0;
    ctl->launches = 0; ctl->slot_next = 0;
    for (int i = 0; i < 8; ++i) ctl->front_launch[i] = 0;
}
void launch_ctl_set(PipeCtl* ctl, unsigned long long pushes, unsigned long long frames, unsigned int pending, hipStream_t st) {
    hipLaunchKernelGGL(ctl_set_kernel, dim3(1), dim3(1), 0, st, ctl, pushes, frames, pending);
}
#endif          // the n_fft-1024 per-column translation unit

// ---- the unpipelined hop: P1-P12 of one stream in one workgroup, one launch per hop (zero added latency).  Four wavefronts for the
// front half (the conv phases split four ways), three for the Griffin-Lim chain behind it (the fourth exits).
template <int NFFT, bool STREAM, bool BF16, int CT>
__global__ __launch_bounds__(kHopPipeThreads, NFFT == 1536 ? 2 : 1) void frame_kernel(DspDev d, CellDev cd, FrameArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[hop_smem<NFFT>()];
    const int tid = threadIdx.x;
    const size_t b = blockIdx.x;
    const float* frames_in = a.frames;
    if (STREAM) {
        ring_shift<NFFT, kHopPipeThreads>(a.ring, a.hop_in, 0, b, tid);
        frames_in = a.ring;
    }
    stft_body<NFFT, false, true, kHopPipeThreads>(smem, d, frames_in, nullptr, a.mel, a.peak, DN_PEAK_NORMALIZE | DN_PRE_WINDOW, b, tid);   // P1-P6
    __syncthreads();
    cell_body<kHopPipeThreads / 64, BF16, CT>(smem, cd, a.mel, a.hx, a.diff, a.hx, 3, a.C, b, tid);                                      // P7
    __syncthreads();
    if (tid >= kHopThreads) return;
    // P8-P12: the inverse-mel contraction is the Griffin-Lim prologue (the linear magnitudes stay in LDS)
    if (!STREAM)
        gl_body<NFFT, true, false>(smem, d, a.mel, a.diff, reinterpret_cast<const v2f*>(a.init), a.seed, a.sid0, a.peak, a.out, a.n_iter, a.mom, b, tid);
    else
        gl_body<NFFT, true, true>(smem, d, a.mel, a.diff, reinterpret_cast<const v2f*>(a.init), a.seed, a.sid0, a.peak, nullptr, a.n_iter, a.mom, b, tid,
                                  a.ola, a.hop_out, 0);
}

template <int NFFT, bool STREAM>
static void launch_frame_n(const DspDev& d, const CellDev& c, const FrameArgs& a, int B, bool bf16, hipStream_t st) {
    constexpr int kUsualC = NFFT == 1536 ? 4 : 5;
    if (a.C == kUsualC) {
        if (bf16) hipLaunchKernelGGL((frame_kernel<NFFT, STREAM, true, kUsualC>), dim3(B), dim3(kHopPipeThreads), 0, st, d, c, a);
        else hipLaunchKernelGGL((frame_kernel<NFFT, STREAM, false, kUsualC>), dim3(B), dim3(kHopPipeThreads), 0, st, d, c, a);
    } else {
        if (bf16) hipLaunchKernelGGL((frame_kernel<NFFT, STREAM, true, 0>), dim3(B), dim3(kHopPipeThreads), 0, st, d, c, a);
        else hipLaunchKernelGGL((frame_kernel<NFFT, STREAM, false, 0>), dim3(B), dim3(kHopPipeThreads), 0, st, d, c, a);
    }
}

#if !defined(DN_HOP_TU_1536) && !defined(DN_HOP_TU_GLW)
void launch_frame(const DspDev& d, const CellDev& c, const FrameArgs& a, int B, bool bf16, hipStream_t st) {
    const bool stream = a.ring != nullptr;
    if (d.n_fft == 1536) {
#ifdef DN_HOP_SPLIT_TUS
        launch_frame_1536(d, c, a, B, bf16, st);
#else
        if (stream) launch_frame_n<1536, true>(d, c, a, B, bf16, st);
        else launch_frame_n<1536, false>(d, c, a, B, bf16, st);
#endif
    } else {
        if (stream) launch_frame_n<1024, true>(d, c, a, B, bf16, st);
        else launch_frame_n<1024, false>(d, c, a, B, bf16, st);
    }
}
#elif defined(DN_HOP_TU_GLW)          // the wavefront-per-stream translation unit
void launch_hop_glw(const DspDev& d, const CellDev& c, const HopArgs& a, bool bf16, hipStream_t st) {
    if (a.ola != nullptr) launch_hop_n<1024, true, true>(d, c, a, bf16, st);
    else launch_hop_n<1024, false, true>(d, c, a, bf16, st);
}
#else          // the n_fft-1536 translation unit
void launch_hop_1536(const DspDev& d, const CellDev& c, const HopArgs& a, bool bf16, hipStream_t st) {
    if (a.ola != nullptr) launch_hop_n<1536, true, false>(d, c, a, bf16, st);
    else launch_hop_n<1536, false, false>(d, c, a, bf16, st);
}
void launch_frame_1536(const DspDev& d, const CellDev& c, const FrameArgs& a, int B, bool bf16, hipStream_t st) {
    if (a.ring != nullptr) launch_frame_n<1536, true>(d, c, a, B, bf16, st);
    else launch_frame_n<1536, false>(d, c, a, B, bf16, st);
}
#endif

}  // namespace dn

#if defined(DN_PROBE) && !defined(DN_HOP_TU_1536) && !defined(DN_HOP_TU_GLW)
// diagnostic build only: the stamps of the Griffin-Lim workgroup 0 of hop_kernel / frame_kernel
extern "C" int dn_probe_read_hop(unsigned long long* host48) {
    return (int)hipMemcpyFromSymbol(host48, HIP_SYMBOL(dn::g_gl_probe), sizeof(dn::g_gl_probe));
}
extern "C" int dn_probe_read_blk_t(unsigned long long* host4096) {
    return (int)hipMemcpyFromSymbol(host4096, HIP_SYMBOL(dn::g_hop_blk_t), sizeof(dn::g_hop_blk_t));
}
extern "C" int dn_probe_read_blk_hw(unsigned int* host2048) {
    return (int)hipMemcpyFromSymbol(host2048, HIP_SYMBOL(dn::g_hop_blk_hw), sizeof(dn::g_hop_blk_hw));
}
extern "C" int dn_probe_read_glw(unsigned long long* host32) {
    return (int)hipMemcpyFromSymbol(host32, HIP_SYMBOL(dn::g_glw_probe), sizeof(dn::g_glw_probe));
}
extern "C" int dn_probe_read_hop_wg(unsigned long long* host8) {
    return (int)hipMemcpyFromSymbol(host8, HIP_SYMBOL(dn::g_hop_wg_probe), sizeof(dn::g_hop_wg_probe));
}
extern "C" int dn_probe_read_hop_stft(unsigned long long* host8) {
    return (int)hipMemcpyFromSymbol(host8, HIP_SYMBOL(dn::g_stft_probe), sizeof(dn::g_stft_probe));
}
extern "C" int dn_probe_read_hop_invmel(unsigned long long* host8) {
    return (int)hipMemcpyFromSymbol(host8, HIP_SYMBOL(dn::g_inv_probe), sizeof(dn::g_inv_probe));
}
#endif
