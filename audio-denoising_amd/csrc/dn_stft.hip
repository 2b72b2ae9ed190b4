// dn_stft.hip -- analysis side of the hop: peak-normalise, Hann, 3-column STFT, |.|, banded mel,
// log1p.  Replaces app3.py:179-195 (P1,P2,P4,P5,P6 of SURVEY.md section 8a).
//
// One workgroup = one stream's frame, three wavefronts = the three STFT columns a centred,
// reflect-padded n_fft-sample input produces.  The frame is read from HBM once (float4,
// coalesced), lives in LDS, and only the 3 x n_mels log-mel values (or the 3 x 513 complex
// bins for the plain Spectrogram entry point) go back to HBM.
#include "dn_stft_body.hpp"

namespace dn {

template <int NFFT, bool WRITE_SPEC, bool WRITE_MEL>
__global__ __launch_bounds__(kStftThreads) void stft_kernel(DspDev d, const float* __restrict__ frames,
                                                            float2* __restrict__ spec, float* __restrict__ mel,
                                                            float* __restrict__ peak_out, uint32_t flags) {
    __shared__ __attribute__((aligned(16))) char smem[stft_smem<NFFT>()];
    stft_body<NFFT, WRITE_SPEC, WRITE_MEL>(smem, d, frames, spec, mel, peak_out, flags, blockIdx.x, threadIdx.x);
}

// MelScale alone (app3.py:193 without the log): one wavefront per (b,t) row.
__global__ __launch_bounds__(64) void mel_kernel(DspDev d, const float* __restrict__ mag, float* __restrict__ mel) {
    __shared__ float row[Geo<1536>::kBins + 7];
    const int lane = threadIdx.x;
    const size_t r = blockIdx.x;
    const int bins = d.n_fft / 2 + 1;
    for (int k = lane; k < bins; k += 64) row[k] = mag[r * bins + k];
    wave_sync();
    for (int m = lane; m < d.n_mels; m += 64) {
        const int s = d.mel_start[m], len = d.mel_len[m];
        float acc = 0.0f;
        for (int i = 0; i < len; ++i) acc = fmaf(d.mel_w[i * d.n_mels + m], row[s + i], acc);
        mel[r * d.n_mels + m] = acc;
    }
}

template <int NFFT>
static void launch_stft_n(const DspDev& d, const float* frames, float* spec, float* mel, float* peak, int B, uint32_t flags,
                          hipStream_t st) {
    dim3 grid(B), block(kStftThreads);
    float2* sp = reinterpret_cast<float2*>(spec);
    if (spec != nullptr && mel != nullptr)
        hipLaunchKernelGGL((stft_kernel<NFFT, true, true>), grid, block, 0, st, d, frames, sp, mel, peak, flags);
    else if (spec != nullptr)
        hipLaunchKernelGGL((stft_kernel<NFFT, true, false>), grid, block, 0, st, d, frames, sp, mel, peak, flags);
    else
        hipLaunchKernelGGL((stft_kernel<NFFT, false, true>), grid, block, 0, st, d, frames, sp, mel, peak, flags);
}

void launch_stft(const DspDev& d, const float* frames, float* spec, float* mel, float* peak, int B, uint32_t flags,
                 hipStream_t st) {
    if (d.n_fft == 1536) launch_stft_n<1536>(d, frames, spec, mel, peak, B, flags, st);
    else launch_stft_n<1024>(d, frames, spec, mel, peak, B, flags, st);
}

void launch_mel(const DspDev& d, const float* mag, float* mel, int rows, hipStream_t st) {
    hipLaunchKernelGGL(mel_kernel, dim3(rows), dim3(64), 0, st, d, mag, mel);
}

}  // namespace dn
