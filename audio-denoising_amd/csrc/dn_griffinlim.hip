// dn_griffinlim.hip -- 32-iteration fast Griffin-Lim on a 3-column spectrogram, one persistent
// workgroup per stream.  Replaces torchaudio.transforms.GriffinLim as called at app3.py:149-153,213
// (algorithm: SURVEY.md Appendix B.5) and, with n_iter = 0, torch.istft (server.py:174,216).
//
// The reference runs 32 x {istft, stft, phase update} as ~65 separate passes over
// (B,513,3) complex tensors.  Here a workgroup of three wavefronts (one per STFT column) keeps
// the whole problem on chip for all iterations:
//   registers : per lane 4 bin PAIRS (k, 512-k) of magnitude, current phase estimate and previous
//               rebuilt spectrum (the Hermitian split/merge of the real FFT is then lane-local),
//               the FFT twiddles and the folded window constants
//   LDS       : one 576-entry complex exchange tile per wave, and a ping-pong pair of
//               overlap-add lines (the 2*n_fft padded signal collapses to two n_fft lines
//               because only the centre n_fft samples survive the istft trim)
//   HBM       : magnitudes in (3*513 floats), waveform out (1024 floats).  That is all.
// One __syncthreads per iteration (overlap-add hand-off between the three columns); every
// FFT-internal exchange is wave-private.
#include "dn_gl_body.hpp"

namespace dn {

template <int NFFT, bool FROM_MEL>
__global__ __launch_bounds__(kGlThreads) void griffinlim_kernel(DspDev d, const float* __restrict__ mag,
                                                                const float* __restrict__ diff,
                                                                const v2f* __restrict__ init, uint64_t seed,
                                                                uint64_t sid0, const float* __restrict__ scale,
                                                                float* __restrict__ wave, int n_iter, float mom) {
    __shared__ __attribute__((aligned(16))) char smem[gl_smem<NFFT>()];
    gl_body<NFFT, FROM_MEL>(smem, d, mag, diff, init, seed, sid0, scale, wave, n_iter, mom, blockIdx.x, threadIdx.x);
}

void launch_griffinlim(const DspDev& d, const float* mag, const float* init, uint64_t seed, uint64_t sid0,
                       const float* scale, float* wave, int B, int n_iter, float momentum, hipStream_t st) {
    const float mom = momentum / (1.0f + momentum);
    const v2f* ia = reinterpret_cast<const v2f*>(init);
    if (d.n_fft == 1536)
        hipLaunchKernelGGL((griffinlim_kernel<1536, false>), dim3(B), dim3(kGlThreads), 0, st, d, mag, (const float*)nullptr, ia, seed, sid0, scale, wave, n_iter, mom);
    else
        hipLaunchKernelGGL((griffinlim_kernel<1024, false>), dim3(B), dim3(kGlThreads), 0, st, d, mag, (const float*)nullptr, ia, seed, sid0, scale, wave, n_iter, mom);
}

// The initial phases a Griffin-Lim launch with init_angles == NULL draws for (seed, stream_id0 + stream): [B][3][K] complex, real and imaginary
// part ~ U[0,1) independently (torchaudio's rand_init=True: torch.rand(complex64), app3.py:149-153), Philox4x32-10 keyed by
// (seed; bin pair, column, stream id) -- the same rand_angle_pair() the kernels call, so handing the result back as init_angles reproduces the launch.
__global__ void draw_phases_kernel(float2* __restrict__ out, uint64_t seed, uint64_t sid0, int bins) {
    const size_t b = blockIdx.x / 3;
    const int col = blockIdx.x % 3;
    const int nc = bins - 1;
    for (int m = threadIdx.x; m <= nc / 2; m += blockDim.x) {          // one block per bin pair (m, nc - m)
        v2f lo, hi;
        rand_angle_pair(seed, sid0 + b, col, m, lo, hi);
        out[(b * 3 + col) * bins + m] = make_float2(lo[0], lo[1]);
        if (2 * m != nc) out[(b * 3 + col) * bins + nc - m] = make_float2(hi[0], hi[1]);
    }
}
void launch_draw_phases(const DspDev& d, float* out, uint64_t seed, uint64_t sid0, int B, hipStream_t st) {
    hipLaunchKernelGGL(draw_phases_kernel, dim3(3 * B), dim3(256), 0, st, reinterpret_cast<float2*>(out), seed, sid0, d.n_fft / 2 + 1);
}

// P8..P12 in one launch: residual -> mel magnitude -> inverse mel -> Griffin-Lim -> * peak.
void launch_synthesis(const DspDev& d, const float* x, const float* diff, const float* init, uint64_t seed, uint64_t sid0,
                      const float* scale, float* wave, int B, int n_iter, float momentum, hipStream_t st) {
    const float mom = momentum / (1.0f + momentum);
    const v2f* ia = reinterpret_cast<const v2f*>(init);
    if (d.n_fft == 1536)
        hipLaunchKernelGGL((griffinlim_kernel<1536, true>), dim3(B), dim3(kGlThreads), 0, st, d, x, diff, ia, seed, sid0, scale, wave, n_iter, mom);
    else
        hipLaunchKernelGGL((griffinlim_kernel<1024, true>), dim3(B), dim3(kGlThreads), 0, st, d, x, diff, ia, seed, sid0, scale, wave, n_iter, mom);
}

}  // namespace dn

#ifdef DN_PROBE
// diagnostic build only: the stamps of the stand-alone Griffin-Lim kernels of this translation unit
extern "C" int dn_probe_read_gl(unsigned long long* host48) {
    return (int)hipMemcpyFromSymbol(host48, HIP_SYMBOL(dn::g_gl_probe), sizeof(dn::g_gl_probe));
}
#endif
