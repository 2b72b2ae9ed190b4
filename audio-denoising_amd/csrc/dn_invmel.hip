// dn_invmel.hip -- synthesis-side mel stages.  Replaces app3.py:203-211 (P8,P9,P10):
//   rec  = leaky_relu(model_input - predicted_diff, 0.2)
//   mmag = clamp(expm1(rec), min=0)
//   lin  = clamp(relu(lstsq(fb^T, mmag)), min=0)
// The reference solves an underdetermined full-rank least-squares problem per item
// (torch.linalg.lstsq, driver gels); its solution is the minimum-norm one, pinv(fb^T) @ mmag,
// so here it is ONE dense contraction against a pseudo-inverse computed once per plan.
//
// A workgroup handles kRows = 3 rows (the three columns of one stream): the mel rows sit in
// LDS and are broadcast; each lane owns output bins k, k+192, ... and streams the transposed
// pseudo-inverse ([M][K'], coalesced over k, L2-resident) exactly once for all three rows.
#include "dn_internal.hpp"
#include "dn_wavefft.hpp"

namespace dn {

constexpr int kInvThreads = 192;
constexpr int kInvRows = 3;
constexpr int kMaxMels = 128;

template <bool RESIDUAL>
__global__ __launch_bounds__(kInvThreads) void invmel_kernel(DspDev d, const float* __restrict__ x,
                                                             const float* __restrict__ diff, float* __restrict__ lin,
                                                             int rows) {
    __shared__ float mm[kInvRows][kMaxMels];
    const int tid = threadIdx.x;
    const int M = d.n_mels;
    const size_t r0 = (size_t)blockIdx.x * kInvRows;
    for (int i = tid; i < kInvRows * M; i += kInvThreads) {
        const int r = i / M, m = i - r * M;
        float v = 0.0f;
        if (r0 + r < (size_t)rows) {
            v = x[(r0 + r) * M + m];
            if (RESIDUAL) {
                v = v - diff[(r0 + r) * M + m];
                v = v >= 0.0f ? v : 0.2f * v;          // leaky_relu, app3.py:204
                v = fmaxf(expm1f(v), 0.0f);            // app3.py:207-208
            }
        }
        mm[r][m] = v;
    }
    __syncthreads();
    // thread <-> bins tid, tid+192, tid+384: three independent load streams, 8 rows of pinv_t in flight each
    float acc[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r) acc[r][0] = acc[r][1] = acc[r][2] = 0.0f;
    const float* p = d.pinv_t + tid;
    const bool third = tid + 2 * kInvThreads < kBins;
#pragma unroll 8
    for (int m = 0; m < M; ++m) {
        const float* pm = p + (size_t)m * d.pinv_stride;
        const float p0 = pm[0], p1 = pm[kInvThreads], p2 = third ? pm[2 * kInvThreads] : 0.0f;
        const float m0 = mm[0][m], m1 = mm[1][m], m2 = mm[2][m];
        acc[0][0] = fmaf(p0, m0, acc[0][0]); acc[0][1] = fmaf(p0, m1, acc[0][1]); acc[0][2] = fmaf(p0, m2, acc[0][2]);
        acc[1][0] = fmaf(p1, m0, acc[1][0]); acc[1][1] = fmaf(p1, m1, acc[1][1]); acc[1][2] = fmaf(p1, m2, acc[1][2]);
        acc[2][0] = fmaf(p2, m0, acc[2][0]); acc[2][1] = fmaf(p2, m1, acc[2][1]); acc[2][2] = fmaf(p2, m2, acc[2][2]);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int k = tid + kInvThreads * r;
        if (k < kBins) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (r0 + c < (size_t)rows) lin[(r0 + c) * kBins + k] = fmaxf(acc[r][c], 0.0f);
        }
    }
}

void launch_invmel(const DspDev& d, const float* x, const float* diff, float* lin, int rows, hipStream_t st) {
    dim3 grid((rows + kInvRows - 1) / kInvRows), block(kInvThreads);
    if (diff != nullptr)
        hipLaunchKernelGGL((invmel_kernel<true>), grid, block, 0, st, d, x, diff, lin, rows);
    else
        hipLaunchKernelGGL((invmel_kernel<false>), grid, block, 0, st, d, x, diff, lin, rows);
}

}  // namespace dn
