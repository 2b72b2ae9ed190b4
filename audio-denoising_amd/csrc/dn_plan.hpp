// dn_plan.hpp -- device-side views of the immutable handles (passed to kernels by value).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dn {

// STFT / mel / inverse-mel / Griffin-Lim constants, all resident in HBM (L2-hot: ~0.4 MB total).
struct DspDev {
    int n_fft;              // 1024 or 1536; NC = n_fft/2 is the complex FFT length, K = NC + 1 bins
    const float2* twc;      // [NC]     exp(-2 pi i k / NC)
    const float2* twr;      // [NC/2+1] exp(-2 pi i k / n_fft)
    const float* window;    // [n_fft]  analysis == synthesis window (periodic Hann by default)
    const float* inv_env;   // [n_fft]  1 / (w[i]^2 + w[(i + hop) % n_fft]^2): istft envelope over the kept region
    // window products of the wavefront-per-stream Griffin-Lim (dn_glw_body.hpp), n_fft 1024 only (else null): [4][NC] complex = for columns 0, 1, 2
    // the analysis window x 1/envelope of the column's source samples, then the synthesis window / NC -- every workgroup copies them to LDS
    const float2* glw_tables;
    // banded mel filterbank: filter m covers bins [mel_start[m], mel_start[m]+mel_len[m])
    const int* mel_start;   // [M]
    const int* mel_len;     // [M]
    const float* mel_w;     // [mel_maxlen][M]  weight of the i-th bin of filter m
    int mel_maxlen;
    // the same bands as a packed schedule for one wavefront (dn_stft_body.hpp): four lanes share a filter (lane l: filter
    // 16 g + l / 4 of group g, taps 4 u + l % 4), mel_qsteps <= mel_q_steps(n_fft) = 32 / 48 steps of 64 (weight, bin) pairs in lane order; bit t of
    // mel_qlast = step t closes its group of 16 filters.  mel_qsteps = 0: bands too long for the schedule, use the plain one.
    const float2* mel_q;    // [mel_qsteps][64]  weight, bin index as integer bits (weight 0, bin 0 where a lane has no tap)
    int mel_qsteps;
    unsigned long long mel_qlast;
    int n_mels;
    const float* pinv_t;    // [M][kPinvStride]  pseudo-inverse of fb^T, transposed, row-padded
    int pinv_stride;
    // The same operator in factors, pinv = fb G^-1 with G = fb^T fb, when the plan built it itself from a bank with at most two filters a bin
    // (every triangular bank): lin = fb (G^-1 mel).  G is the Gram matrix of overlapping triangles -- nearly tridiagonal -- and its inverse
    // decays by ~30x every three diagonals: beyond +-16 every entry is below 1e-8 of the largest one (checked when the plan is built),
    // i.e. below fp32 rounding of the sum it would enter, and is not stored.  33 diagonals of G^-1 and two weights a bin are 19 KB
    // instead of the dense matrix's 164 KB (80 mels, 513 bins), and ~100 instructions a thread instead of ~900.  nullptr: dense form only.
    const float* ginv_band; // [33][M]  ginv_band[t][a] = G^-1[a][a - 16 + t] (0 outside the matrix)
    const float4* fb2;      // [K]      w0, w1, then the two filter indices as integer bits
};

// Packed GRUUNet2 weights (data channels only; the position-code channels are folded into
// the per-position bias tables `bt_*`, which depend on the number of compressed bins C).
struct CellDev {
    // Encoder / decoder weights are stored as v_mfma_f32_16x16x4_f32 A fragments in lane order (dn_cell.hip):
    //   w_down[l]: [m-tile][k-step][64]          K = taps x channels-in-fours (level 0: the 3 taps)
    //   w_up[l<3]: [m-tile][tap set][k-step][64]  tap sets k=1 (even outputs), k=2 and k=0 (odd outputs)
    //   w_up[3]  : the single-output-channel last level runs on the VALU: [part: a, skip][17][4] = w[k=1], w[k=2], w[k=0], 0
    const float* w_down[4];  // data channels 1,17,17,17 -> 17,17,17,51
    const float* w_gh;       // [m-tile 4][k-step 13][64]  hidden-gate conv fragments, K slot = 3 c + tap (held in VGPRs across time steps)
    const float* w_up[4];    // data channels 17,34,34,34 -> 17,17,17,1
    // bf16 variants of the same fragments for v_mfma_f32_16x16x32_bf16 (BASELINE config 3): 8 bf16 per lane per k-step,
    // K = 32 slots = channels 8 q + j of one tap (level 0: the 3 taps); [m-tile][k-step][64][8] / [m-tile][set][part][64][8]
    const void* wb_down[4];
    const void* wb_up[3];
    const float* bt_down[4]; // [Cout][Lout]    Lout = 8C,4C,2C,C
    const float* bt_gh;      // [51][C]
    const float* bt_up[4];   // [Cout][Lout]    Lout = 2C,4C,8C,16C
};

}  // namespace dn
