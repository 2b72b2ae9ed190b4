// dn_hop_common.hpp -- what the fused launches (dn_hop.hip: hop_kernel, frame_kernel; dn_group.hip: group_kernel) share: workgroup shape, LDS budget,
// the ring shift of the streaming front end and the layout of a pipe's scratch slots.
#pragma once
#include "dn_cell_body.hpp"
#include "dn_gl_body.hpp"
#include "dn_glw_body.hpp"
#include "dn_invmel_body.hpp"
#include "dn_stft_body.hpp"

namespace dn {

constexpr int kHopThreads = 192;          // the Griffin-Lim chain: one wavefront per STFT column
constexpr int kHopPipeThreads = 256;      // workgroup size of the fused launches: a fourth wavefront for the front half
#ifndef DN_GL_PRIO
#define DN_GL_PRIO 3
#endif
#ifndef DN_HS_PRIO
#define DN_HS_PRIO 1
#endif
constexpr int kFrontPerCu = 4;            // front workgroups a CU when they run as a launch of their own (hop_kernel<.., FRONT>, group_kernel<.., kGroupFronts>)
constexpr int cmax(int a, int b) { return a > b ? a : b; }
template <int NFFT> constexpr int hop_smem() {
    return cmax(cmax(cmax(kCellSmem, gl_smem<NFFT>()), cmax(stft_smem<NFFT>(), kInvSmem)), NFFT == 1024 ? glw_smem<1024>() : 0);
}
template <int NFFT> constexpr int front_smem() { return cmax(cmax(kCellSmemUnstaged, stft_smem<NFFT>()), kInvSmem); }    // 35 KB: four a CU
static_assert(kHopThreads == kGlThreads && kHopThreads == kStftThreads && kHopThreads == kInvThreads, "one block size for all bodies");

// ring <- concat(ring[hop:], hop_in): every thread holds its float4s before anything is overwritten   (app3.py:174,226)
template <int NFFT, int THREADS = kHopThreads>
__device__ __forceinline__ void ring_shift(float* ring, const void* hop_in, int in_s16, size_t b, int tid) {
    constexpr int kNR = NFFT, kLine4 = kNR / 4, kHop4 = kNR / 8;
    static_assert(kLine4 <= 2 * THREADS, "two float4 per thread cover the line");
    float4* r4 = reinterpret_cast<float4*>(ring + b * kNR);
    float4 v[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i4 = tid + THREADS * r;
        v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i4 < kLine4 - kHop4) v[r] = r4[i4 + kHop4];
        else if (i4 < kLine4) {
            const int j4 = i4 - (kLine4 - kHop4);
            if (in_s16) {      // int16 -> float32 / iinfo(int16).max   (app3.py:172)
                const short4 q = reinterpret_cast<const short4*>(static_cast<const short*>(hop_in) + b * (kNR / 2))[j4];
                v[r] = make_float4((float)q.x / 32767.0f, (float)q.y / 32767.0f, (float)q.z / 32767.0f, (float)q.w / 32767.0f);
            } else {
                v[r] = reinterpret_cast<const float4*>(static_cast<const float*>(hop_in) + b * (kNR / 2))[j4];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i4 = tid + THREADS * r;
        if (i4 < kLine4) r4[i4] = v[r];
    }
    __syncthreads();
}

// offsets (in floats) of the parts of a scratch slot.  meta: kSlotMeta u32 per stream, written by the frame's front workgroup and read by its
// Griffin-Lim workgroup in the next launch -- everything the pending hop is finished with is the FRAME's own, not the next call's:
//   [0] has injected phases  [1,2] Griffin-Lim seed  [3,4] stream id of stream 0  [5] head-start iterations already run
//   [6] n_iter  [7] momentum / (1 + momentum) (bits)  [8,9] where the frame goes (frame mode: the `out` of its dn_pipe_submit)
struct SlotLayout {
    size_t diff, peak, meta, lin;
    __host__ __device__ SlotLayout(int B, int M, int K) {
        diff = (size_t)B * 3 * M; peak = 2 * diff; meta = peak + B; lin = meta + kSlotMeta * (size_t)B; (void)K;
    }
};

// A zero the compiler cannot see through.  dn_group.hip adds it, once per hop, to the pointers through which the front half reads the plan's and the
// model's views in device memory: loads through a pointer that changes every iteration are not loop-invariant, so a loop over large inlined stages
// does not hoist every table access of every stage above itself.
#ifndef DN_OPAQUE_ZERO
#define DN_OPAQUE_ZERO(z) asm volatile("s_mov_b32 %0, 0" : "=s"(z))
#endif

// slot of the frame that is `back` frames behind the next one (slot_next is the slot the next front half writes)
__device__ __forceinline__ int slot_behind(unsigned int slot_next, int back, int n_slots) {
    int s = (int)slot_next - back;
    while (s < 0) s += n_slots;
    return s;
}

}  // namespace dn
