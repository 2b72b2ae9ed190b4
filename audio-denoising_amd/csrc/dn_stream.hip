// dn_stream.hip -- per-stream streaming state on the device (P12 of SURVEY.md section 8a):
// the input ring (app3.py:178,226) and the output overlap-add buffer (app3.py:219-224).
// One workgroup per stream; every thread owns one float4 of the n_fft-sample line, so the
// in-place shifts are a load, a barrier and a store.  (The pipelined path does this inside hop_kernel.)
#include "dn_internal.hpp"

namespace dn {

// ring <- concat(ring[hop:], hop_in)           (input_buffer = input_buffer[hop:], then the new hop arrives)
template <int NFFT>
__global__ __launch_bounds__(NFFT / 4) void stream_shift_kernel(const float* __restrict__ hop_in, float* ring) {
    constexpr int kLineThreads = NFFT / 4, kHop4 = NFFT / 8;
    const int tid = threadIdx.x;
    const size_t b = blockIdx.x;
    float4* r4 = reinterpret_cast<float4*>(ring + b * NFFT);
    const float4* h4 = reinterpret_cast<const float4*>(hop_in + b * (NFFT / 2));
    const float4 v = tid < kLineThreads - kHop4 ? r4[tid + kHop4] : h4[tid - (kLineThreads - kHop4)];
    __syncthreads();
    r4[tid] = v;
}

// hop_out <- ola[:hop]; ola <- concat(ola[hop:], 0) + y        (app3.py:219-224)
template <int NFFT>
__global__ __launch_bounds__(NFFT / 4) void stream_ola_kernel(const float* __restrict__ y, float* ola,
                                                              float* __restrict__ hop_out) {
    constexpr int kLineThreads = NFFT / 4, kHop4 = NFFT / 8;
    const int tid = threadIdx.x;
    const size_t b = blockIdx.x;
    float4* o4 = reinterpret_cast<float4*>(ola + b * NFFT);
    const float4* y4 = reinterpret_cast<const float4*>(y + b * NFFT);
    float4* out4 = reinterpret_cast<float4*>(hop_out + b * (NFFT / 2));
    const float4 cur = o4[tid];
    float4 nxt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < kLineThreads - kHop4) nxt = o4[tid + kHop4];
    const float4 add = y4[tid];
    __syncthreads();
    if (tid < kHop4) out4[tid] = cur;
    o4[tid] = make_float4(nxt.x + add.x, nxt.y + add.y, nxt.z + add.z, nxt.w + add.w);
}

void launch_stream_shift(int n_fft, const float* hop_in, float* ring, int B, hipStream_t st) {
    if (n_fft == 1536) hipLaunchKernelGGL(stream_shift_kernel<1536>, dim3(B), dim3(1536 / 4), 0, st, hop_in, ring);
    else hipLaunchKernelGGL(stream_shift_kernel<1024>, dim3(B), dim3(1024 / 4), 0, st, hop_in, ring);
}
void launch_stream_ola(int n_fft, const float* y, float* ola, float* hop_out, int B, hipStream_t st) {
    if (n_fft == 1536) hipLaunchKernelGGL(stream_ola_kernel<1536>, dim3(B), dim3(1536 / 4), 0, st, y, ola, hop_out);
    else hipLaunchKernelGGL(stream_ola_kernel<1024>, dim3(B), dim3(1024 / 4), 0, st, y, ola, hop_out);
}

}  // namespace dn
