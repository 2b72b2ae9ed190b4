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
    for (int k = tid; k < kBins; k += kInvThreads) {
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
        const float* p = d.pinv_t + k;
        for (int m = 0; m < M; ++m) {
            const float pv = p[(size_t)m * d.pinv_stride];
            a0 = fmaf(pv, mm[0][m], a0);
            a1 = fmaf(pv, mm[1][m], a1);
            a2 = fmaf(pv, mm[2][m], a2);
        }
        if (r0 + 0 < (size_t)rows) lin[(r0 + 0) * kBins + k] = fmaxf(a0, 0.0f);
        if (r0 + 1 < (size_t)rows) lin[(r0 + 1) * kBins + k] = fmaxf(a1, 0.0f);
        if (r0 + 2 < (size_t)rows) lin[(r0 + 2) * kBins + k] = fmaxf(a2, 0.0f);
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
