// dn_hop.hip -- software-pipelined hop: ONE launch per hop that overlaps hop n's Griffin-Lim with hop n+1's
// analysis + model + inverse mel (app3.py:178-217 for B streams, consecutive loop iterations overlapped).
//
// Consecutive hops depend on each other only through hx (the model); hop n's Griffin-Lim (~3/4 of a hop, a
// strictly serial chain per stream that occupies three wavefronts of a CU) does not depend on hop n+1's
// front half.  A launch therefore carries two kinds of workgroups:
//   blocks [0, back_B)           Griffin-Lim of the PREVIOUS hop, reading scratch slot s^1
//   blocks [back_B, back_B + B)  P1-P10 of THIS hop (stft -> GRUUNet2 -> inverse mel, one stream per workgroup,
//                                stages separated by workgroup barriers), writing scratch slot s
// Both kinds are resident together (192 threads, <= 35 KB LDS, 204 VGPRs: a Griffin-Lim and a front workgroup
// share a CU), so the front half fills issue slots and the fourth SIMD the latency-bound Griffin-Lim leaves
// idle.  Everything is on the caller's stream: launch k+1 is ordered behind launch k, which is all the
// synchronisation the slot hand-over needs -- no events, no second stream, no cross-queue latency
// (a two-stream/event version of this overlap lost ~13 us per hop to cross-queue signalling).
#include "dn_cell_body.hpp"
#include "dn_gl_body.hpp"
#include "dn_invmel_body.hpp"
#include "dn_stft_body.hpp"

namespace dn {

constexpr int kHopThreads = 192;
constexpr int kHopSmem = kCellSmem > kGlSmem ? (kCellSmem > kStftSmem ? kCellSmem : kStftSmem)
                                             : (kGlSmem > kStftSmem ? kGlSmem : kStftSmem);
static_assert(kHopThreads == kGlThreads && kHopThreads == kStftThreads && kHopThreads == kInvThreads, "one block size for all bodies");

__global__ __launch_bounds__(kHopThreads) void hop_kernel(DspDev d, CellDev cd, HopArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[kHopSmem];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < a.back_B) {
        gl_body<false>(smem, d, a.gl_lin, nullptr, reinterpret_cast<const v2f*>(a.gl_init), a.gl_seed, a.gl_sid0, a.gl_peak,
                       a.gl_out, a.n_iter, a.mom, blockIdx.x, tid);
    } else {
        const size_t b = blockIdx.x - a.back_B;
        stft_body<false, true>(smem, d, a.frames, nullptr, a.mel, a.peak, DN_PEAK_NORMALIZE | DN_PRE_WINDOW, b, tid);   // P1-P6
        __syncthreads();
        cell_body<kHopThreads / 64>(smem, cd, a.mel, a.hx, a.diff, a.hx, 3, a.C, b, tid);                               // P7
        __syncthreads();
        invmel_body<true>(smem, d, a.mel, a.diff, a.lin, 3 * a.front_B, b * 3, tid);                                    // P8-P10
    }
}

void launch_hop(const DspDev& d, const CellDev& c, const HopArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(hop_kernel, dim3(a.back_B + a.front_B), dim3(kHopThreads), 0, st, d, c, a);
}

}  // namespace dn
