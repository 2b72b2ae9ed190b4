// dn_general.hip -- arbitrary-length chunks: the request loop of the reference's socket server (server.py:199-217).
//
// The per-hop path is specialised for one n_fft-sample frame (3 STFT columns).  The server variant receives
// chunks of any length L, runs GRUUNet2 over all T = 1 + L/hop columns, and resynthesises with the NOISY phase
// (torch.polar + InverseSpectrogram) instead of Griffin-Lim.  Three small kernels built from the same one-wave FFT:
//   stft_general_kernel   Spectrogram(power=None) [+ MelScale, log1p]      server.py:207-210   one wave per column
//   server_rows_kernel    relu(out)*3, exp(log_mel - out) - 1, InverseMelScale, torch.polar(., phase)   server.py:213-216
//   istft_general_kernel  InverseSpectrogram (length=None)                  server.py:216       two waves per output hop
// GRUUNet2 itself is the ordinary dn_cell_forward (any T), with the `hx = hx * 0.9` of server.py:214 folded into
// the state write-back (dn_cell_forward_ex).
#include "dn_invmel_body.hpp"
#include "dn_stft_body.hpp"

namespace dn {

template <int NFFT>
__global__ __launch_bounds__(64) void stft_general_kernel(DspDev d, const float* __restrict__ x, int L, int T,
                                                          float2* __restrict__ spec, float* __restrict__ logmel) {
    using G = Geo<NFFT>;
    constexpr int kNV = G::kNV, kNP = G::kNP, kHop = G::kHop, kNC = G::kNC, kBins = G::kBins;
    __shared__ v2f tile[G::kTile];
    __shared__ float magrow[kBins + 7];
    const int lane = threadIdx.x;
    const size_t blk = blockIdx.x;
    const size_t b = blk / T;
    const int t = (int)(blk - b * T);
    const float* xb = x + b * (size_t)L;
    typename G::Fft::Tw tw;
    G::Fft::load(tw, reinterpret_cast<const v2f*>(d.twc), lane);
    v2f wkh[kNP], v[kNV];
#pragma unroll
    for (int s = 0; s < kNP; ++s) wkh[s] = cscale(reinterpret_cast<const v2f*>(d.twr)[lane + 64 * s], 0.5f);
#pragma unroll
    for (int s = 0; s < kNV; ++s) {
        const int m = lane + 64 * s;
        int i0 = t * kHop + 2 * m - kHop, i1 = i0 + 1;            // centred: pad n_fft/2 by reflection
        i0 = i0 < 0 ? -i0 : (i0 >= L ? 2 * L - 2 - i0 : i0);
        i1 = i1 < 0 ? -i1 : (i1 >= L ? 2 * L - 2 - i1 : i1);
        const v2f ww = reinterpret_cast<const v2f*>(d.window)[m];
        v[s] = mk2(xb[i0] * ww[0], xb[i1] * ww[1]);
    }
    G::Fft::template run<false>(v, tw, tile, lane);
    v2f lo[kNP], hi[kNP], mid;
    rfft_split_pairs<kNV>(v, wkh, lane, lo, hi, mid);
    if (spec != nullptr) {
        v2f* srow = reinterpret_cast<v2f*>(spec) + blk * kBins;
#pragma unroll
        for (int s = 0; s < kNP; ++s) {
            srow[lane + 64 * s] = lo[s];
            srow[kNC - (lane + 64 * s)] = hi[s];
        }
        if (lane == 0) srow[kNC / 2] = mid;
    }
    if (logmel != nullptr) {
#pragma unroll
        for (int s = 0; s < kNP; ++s) {
            magrow[lane + 64 * s] = hypotf(lo[s][0], lo[s][1]);
            magrow[kNC - (lane + 64 * s)] = hypotf(hi[s][0], hi[s][1]);
        }
        if (lane == 0) magrow[kNC / 2] = hypotf(mid[0], mid[1]);
        wave_sync();
        for (int m = lane; m < d.n_mels; m += 64) {
            const int st = d.mel_start[m], len = d.mel_len[m];
            float acc = 0.0f;
            for (int i = 0; i < len; ++i) acc = fmaf(d.mel_w[i * d.n_mels + m], magrow[st + i], acc);
            logmel[blk * d.n_mels + m] = log1pf(acc);
        }
    }
}

// One workgroup (192 threads) per (b,t) row: mel magnitude exp(log_mel - 3 relu(out)) - 1  ->  pinv contraction, relu
// -> times the unit phasor of the noisy bin (torch.polar(O, spec.angle()); angle(0) = 0).
template <int NFFT>
__global__ __launch_bounds__(kInvThreads) void server_rows_kernel(DspDev d, const float* __restrict__ logmel,
                                                                  const float* __restrict__ model_out,
                                                                  const float2* spec_in, float2* spec_out) {      // may alias (in-place): no __restrict__
    constexpr int kBins = Geo<NFFT>::kBins;
    constexpr int kRounds = (kBins + kInvThreads - 1) / kInvThreads;
    __shared__ float mm[kMaxMels];
    const int tid = threadIdx.x;
    const size_t row = blockIdx.x;
    const int M = d.n_mels;
    for (int m = tid; m < M; m += kInvThreads) {
        const float o = fmaxf(model_out[row * M + m], 0.0f) * 3.0f;            // leaky_relu(., 0) * 3, server.py:213
        mm[m] = expf(logmel[row * M + m] - o) - 1.0f;                           // server.py:215
    }
    __syncthreads();
    float acc[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; ++r) acc[r] = 0.0f;
    const float* p = d.pinv_t + tid;
#pragma unroll 4
    for (int m = 0; m < M; ++m) {
        const float* pm = p + (size_t)m * d.pinv_stride;
        const float mv = mm[m];
#pragma unroll
        for (int r = 0; r < kRounds; ++r) acc[r] = fmaf(pm[kInvThreads * r], mv, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        const int k = tid + kInvThreads * r;
        if (k < kBins) {
            const float mag = fmaxf(acc[r], 0.0f);                              // InverseMelScale's relu
            const float2 z = spec_in[row * kBins + k];
            const float h = hypotf(z.x, z.y);
            const float2 u = h > 0.0f ? make_float2(z.x / h, z.y / h) : make_float2(1.0f, 0.0f);
            spec_out[row * kBins + k] = make_float2(mag * u.x, mag * u.y);      // torch.polar(O, phase), server.py:216
        }
    }
}

// torch.istft(center=True, length=None) for any number of columns T >= 2.  One workgroup = one output hop j: wave 0
// inverts column j (its second half lands in the hop), wave 1 column j+1 (first half); sum / window envelope.
template <int NFFT>
__global__ __launch_bounds__(128) void istft_general_kernel(DspDev d, const float2* __restrict__ spec, int T, float* __restrict__ wave) {
    using G = Geo<NFFT>;
    constexpr int kNV = G::kNV, kNP = G::kNP, kHop = G::kHop, kNC = G::kNC, kBins = G::kBins;
    __shared__ v2f tile[2][G::kTile];
    __shared__ float seg[2][kHop];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t blk = blockIdx.x;
    const size_t b = blk / (T - 1);
    const int j = (int)(blk - b * (T - 1));
    const v2f* srow = reinterpret_cast<const v2f*>(spec) + (b * T + j + w) * kBins;
    typename G::Fft::Tw tw;
    G::Fft::load(tw, reinterpret_cast<const v2f*>(d.twc), lane);
    v2f wkh[kNP], lo[kNP], hi[kNP], v[kNV];
#pragma unroll
    for (int s = 0; s < kNP; ++s) {
        wkh[s] = cscale(reinterpret_cast<const v2f*>(d.twr)[lane + 64 * s], 0.5f);
        lo[s] = srow[lane + 64 * s];
        hi[s] = srow[kNC - (lane + 64 * s)];
    }
    const v2f mid = srow[kNC / 2];
    irfft_merge_pairs<kNV>(lo, hi, mid, wkh, lane, v);
    G::Fft::template run<true>(v, tw, tile[w], lane);
    // column j keeps samples n >= hop (registers s >= NP), column j+1 samples n < hop (s < NP)
    if (w == 0) {
#pragma unroll
        for (int s = kNP; s < kNV; ++s) {
            const int m = lane + 64 * s;
            const v2f ww = reinterpret_cast<const v2f*>(d.window)[m];
            *reinterpret_cast<v2f*>(&seg[0][2 * m - kHop]) = v[s] * ww * (1.0f / (float)kNC);
        }
    } else {
#pragma unroll
        for (int s = 0; s < kNP; ++s) {
            const int m = lane + 64 * s;
            const v2f ww = reinterpret_cast<const v2f*>(d.window)[m];
            *reinterpret_cast<v2f*>(&seg[1][2 * m]) = v[s] * ww * (1.0f / (float)kNC);
        }
    }
    __syncthreads();
    float* out = wave + (b * (size_t)(T - 1) + j) * kHop;
    for (int n = tid; n < kHop; n += 128) out[n] = (seg[0][n] + seg[1][n]) * d.inv_env[n];
}

void launch_stft_general(const DspDev& d, const float* x, float* spec, float* logmel, int B, int L, hipStream_t st) {
    const int T = 1 + L / (d.n_fft / 2);
    float2* sp = reinterpret_cast<float2*>(spec);
    if (d.n_fft == 1536) hipLaunchKernelGGL(stft_general_kernel<1536>, dim3(B * T), dim3(64), 0, st, d, x, L, T, sp, logmel);
    else hipLaunchKernelGGL(stft_general_kernel<1024>, dim3(B * T), dim3(64), 0, st, d, x, L, T, sp, logmel);
}

void launch_server_rows(const DspDev& d, const float* logmel, const float* model_out, const float* spec_in, float* spec_out, int rows,
                        hipStream_t st) {
    const float2* si = reinterpret_cast<const float2*>(spec_in);
    float2* so = reinterpret_cast<float2*>(spec_out);
    if (d.n_fft == 1536) hipLaunchKernelGGL(server_rows_kernel<1536>, dim3(rows), dim3(kInvThreads), 0, st, d, logmel, model_out, si, so);
    else hipLaunchKernelGGL(server_rows_kernel<1024>, dim3(rows), dim3(kInvThreads), 0, st, d, logmel, model_out, si, so);
}

void launch_istft_general(const DspDev& d, const float* spec, float* wave, int B, int T, hipStream_t st) {
    const float2* sp = reinterpret_cast<const float2*>(spec);
    if (d.n_fft == 1536) hipLaunchKernelGGL(istft_general_kernel<1536>, dim3(B * (T - 1)), dim3(128), 0, st, d, sp, T, wave);
    else hipLaunchKernelGGL(istft_general_kernel<1024>, dim3(B * (T - 1)), dim3(128), 0, st, d, sp, T, wave);
}

}  // namespace dn
