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
#include "dn_invmel_body.hpp"

namespace dn {

template <int NFFT, bool RESIDUAL>
__global__ __launch_bounds__(kInvThreads) void invmel_kernel(DspDev d, const float* __restrict__ x,
                                                             const float* __restrict__ diff, float* __restrict__ lin,
                                                             int rows) {
    __shared__ __attribute__((aligned(16))) char smem[kInvSmem];
    invmel_body<NFFT, RESIDUAL>(smem, d, x, diff, lin, rows, (size_t)blockIdx.x * kInvRows, threadIdx.x);
}

template <int NFFT>
static void launch_invmel_n(const DspDev& d, const float* x, const float* diff, float* lin, int rows, hipStream_t st) {
    dim3 grid((rows + kInvRows - 1) / kInvRows), block(kInvThreads);
    if (diff != nullptr)
        hipLaunchKernelGGL((invmel_kernel<NFFT, true>), grid, block, 0, st, d, x, diff, lin, rows);
    else
        hipLaunchKernelGGL((invmel_kernel<NFFT, false>), grid, block, 0, st, d, x, diff, lin, rows);
}

void launch_invmel(const DspDev& d, const float* x, const float* diff, float* lin, int rows, hipStream_t st) {
    if (d.n_fft == 1536) launch_invmel_n<1536>(d, x, diff, lin, rows, st);
    else launch_invmel_n<1024>(d, x, diff, lin, rows, st);
}

}  // namespace dn
