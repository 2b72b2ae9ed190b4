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
constexpr int cmax(int a, int b) { return a > b ? a : b; }
template <int NFFT> constexpr int hop_smem() { return cmax(cmax(kCellSmem, gl_smem<NFFT>()), cmax(stft_smem<NFFT>(), kInvSmem)); }
static_assert(kHopThreads == kGlThreads && kHopThreads == kStftThreads && kHopThreads == kInvThreads, "one block size for all bodies");

// n_fft 1536: the Griffin-Lim body would take 330 registers and shut the front workgroup out of the CU; capping the
// kernel at two waves per SIMD (256 registers, ~80 values spilled to scratch) keeps both halves resident (+30 % at 1024
// streams).  n_fft 1024 fits in 229 registers without a cap (capping it costs 8 %).
template <int NFFT>
__global__ __launch_bounds__(kHopThreads, NFFT == 1536 ? 2 : 1) void hop_kernel(DspDev d, CellDev cd, HopArgs a) {
    constexpr int kNR = NFFT;
    __shared__ __attribute__((aligned(16))) char smem[hop_smem<NFFT>()];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < a.back_B) {
        // the Griffin-Lim chain is the critical path of the launch: let its waves win issue arbitration against the
        // front-half waves they share SIMDs with
        __builtin_amdgcn_s_setprio(3);
        if (a.ola == nullptr)
            gl_body<NFFT, false, false>(smem, d, a.gl_lin, nullptr, reinterpret_cast<const v2f*>(a.gl_init), a.gl_seed, a.gl_sid0,
                                        a.gl_peak, a.gl_out, a.n_iter, a.mom, blockIdx.x, tid);
        else
            gl_body<NFFT, false, true>(smem, d, a.gl_lin, nullptr, reinterpret_cast<const v2f*>(a.gl_init), a.gl_seed, a.gl_sid0,
                                       a.gl_peak, nullptr, a.n_iter, a.mom, blockIdx.x, tid, a.ola, a.hop_out, a.out_s16);
    } else {
        const size_t b = blockIdx.x - a.back_B;
        const float* frames = a.frames;
        if (a.ring != nullptr) {
            // ring <- concat(ring[hop:], hop_in): every thread holds its float4s before anything is overwritten
            constexpr int kLine4 = kNR / 4, kHop4 = kNR / 8;
            static_assert(kLine4 <= 2 * kHopThreads, "two float4 per thread cover the line");
            float4* r4 = reinterpret_cast<float4*>(a.ring + b * kNR);
            float4 v[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int i4 = tid + kHopThreads * r;
                v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i4 < kLine4 - kHop4) v[r] = r4[i4 + kHop4];
                else if (i4 < kLine4) {
                    const int j4 = i4 - (kLine4 - kHop4);
                    if (a.in_s16) {      // int16 -> float32 / iinfo(int16).max   (app3.py:172)
                        const short4 q = reinterpret_cast<const short4*>(static_cast<const short*>(a.hop_in) + b * (kNR / 2))[j4];
                        v[r] = make_float4((float)q.x / 32767.0f, (float)q.y / 32767.0f, (float)q.z / 32767.0f, (float)q.w / 32767.0f);
                    } else {
                        v[r] = reinterpret_cast<const float4*>(static_cast<const float*>(a.hop_in) + b * (kNR / 2))[j4];
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int i4 = tid + kHopThreads * r;
                if (i4 < kLine4) r4[i4] = v[r];
            }
            __syncthreads();
            if (a.prime_only) return;
            frames = a.ring;
        }
        stft_body<NFFT, false, true>(smem, d, frames, nullptr, a.mel, a.peak, DN_PEAK_NORMALIZE | DN_PRE_WINDOW, b, tid);   // P1-P6
        __syncthreads();
        cell_body<kHopThreads / 64>(smem, cd, a.mel, a.hx, a.diff, a.hx, 3, a.C, b, tid);                                   // P7
        __syncthreads();
        invmel_body<NFFT, true>(smem, d, a.mel, a.diff, a.lin, 3 * a.front_B, b * 3, tid);                                  // P8-P10
    }
}

void launch_hop(const DspDev& d, const CellDev& c, const HopArgs& a, hipStream_t st) {
    if (d.n_fft == 1536) hipLaunchKernelGGL(hop_kernel<1536>, dim3(a.back_B + a.front_B), dim3(kHopThreads), 0, st, d, c, a);
    else hipLaunchKernelGGL(hop_kernel<1024>, dim3(a.back_B + a.front_B), dim3(kHopThreads), 0, st, d, c, a);
}

}  // namespace dn
