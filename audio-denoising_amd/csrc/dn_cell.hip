// dn_cell.hip -- GRUUNet2.forward: Conv1d U-Net encoder / convolutional-GRU bottleneck /
// ConvTranspose1d decoder, T sequential time steps, persistent hidden state.
// Replaces gruunet2.py:127-156 (DownBlocks), 184-199 (UpBlocks), 228-244 (GRUUNetCell.forward),
// 266-306 (GRUUNet2._gruunet / forward); math restated in SURVEY.md Appendix A.
//
// The reference issues ~60 small tensor ops per time step.  Here ONE workgroup owns ONE stream
// and runs the whole forward out of LDS (activations of kCellChunk = 3 steps: ~34 KB at F = 80):
//   * the encoder does not depend on hx (gruunet2.py:231), so it runs batched over the chunk's
//     time steps; likewise the decoder after the recurrent part;
//   * only  gh = relu(conv(hx)), the GRU gates and h' are sequential in t;
//   * the 6 Gaussian position-code channels are input independent: their convolution is folded
//     into per-position bias tables at plan time (CellDev::bt_*), so kernels convolve data
//     channels only (-26 % MACs);
//   * encoder/decoder: lane = (t, position), wavefront = group of output channels, so every
//     weight is wave-uniform and comes through the scalar cache while activations are
//     stride-1 LDS reads;  bottleneck (51 gate channels x C positions): lane = (position,
//     gate channel) with that lane's 51 recurrent weights pinned in VGPRs across time steps.
#include "dn_internal.hpp"

namespace dn {

constexpr int kCellThreads = 256;

// ---- Conv1d k3 s2 p1 + folded position bias + relu.  in [TT][CIN][2*LOUT] -> out [TT][COUT][LOUT]
// OG = output channels per wavefront.
template <int CIN, int COUT, int OG>
__device__ __forceinline__ void conv_down(cfloat_ptr wgt, const float* __restrict__ bt,
                                          const float* in, float* out, int lout, int tt, int wv, int lane) {
    const int lin = 2 * lout;
    const int o0 = wv * OG;
    if (o0 >= COUT) return;
    const int items = tt * lout;
    for (int it = lane; it < items; it += 64) {
        const int t = it / lout, j = it - t * lout;
        float acc[OG];
#pragma unroll
        for (int oo = 0; oo < OG; ++oo) acc[oo] = (o0 + oo < COUT) ? bt[(o0 + oo) * lout + j] : 0.0f;
        const float* xin = in + (size_t)t * CIN * lin + 2 * j;
        for (int c = 0; c < CIN; ++c) {
            const float x0 = j > 0 ? xin[c * lin - 1] : 0.0f;
            const float x1 = xin[c * lin];
            const float x2 = xin[c * lin + 1];
            cfloat_ptr wc = wgt + c * 3 * COUT + o0;
#pragma unroll
            for (int oo = 0; oo < OG; ++oo) {
                if (o0 + oo < COUT) {
                    acc[oo] = fmaf(wc[oo], x0, acc[oo]);
                    acc[oo] = fmaf(wc[COUT + oo], x1, acc[oo]);
                    acc[oo] = fmaf(wc[2 * COUT + oo], x2, acc[oo]);
                }
            }
        }
#pragma unroll
        for (int oo = 0; oo < OG; ++oo)
            if (o0 + oo < COUT) out[((size_t)t * COUT + o0 + oo) * lout + j] = fmaxf(acc[oo], 0.0f);
    }
}

// ---- ConvTranspose1d k3 s2 p1 output_padding 1 (L -> 2L) on cat(a, skip) data channels + folded
// position bias.  a [TT][CA][L], skip [TT][CS][L] (CS may be 0) -> out [TT][COUT][2L], relu unless LAST.
// Lane = (t, input position i) produces outputs 2i (tap k=1 of x[i]) and 2i+1 (k=2 of x[i], k=0 of x[i+1]).
template <int CA, int CS, int COUT, int OG, bool LAST>
__device__ __forceinline__ void conv_up(cfloat_ptr wgt, const float* __restrict__ bt, const float* a,
                                        const float* skip, float* out, int l, int tt, int wv, int lane,
                                        size_t out_t_stride) {
    const int o0 = wv * OG;
    if (o0 >= COUT) return;
    const int items = tt * l;
    const int lo = 2 * l;
    for (int it = lane; it < items; it += 64) {
        const int t = it / l, i = it - t * l;
        float ev[OG], od[OG];
#pragma unroll
        for (int oo = 0; oo < OG; ++oo) {
            const bool ok = o0 + oo < COUT;
            ev[oo] = ok ? bt[(o0 + oo) * lo + 2 * i] : 0.0f;
            od[oo] = ok ? bt[(o0 + oo) * lo + 2 * i + 1] : 0.0f;
        }
        const bool has_next = i + 1 < l;
        for (int c = 0; c < CA + CS; ++c) {
            const float* src = c < CA ? a + ((size_t)t * CA + c) * l : skip + ((size_t)t * CS + (c - CA)) * l;
            const float x0 = src[i];
            const float x1 = has_next ? src[i + 1] : 0.0f;
            cfloat_ptr wc = wgt + c * 3 * COUT + o0;
#pragma unroll
            for (int oo = 0; oo < OG; ++oo) {
                if (o0 + oo < COUT) {
                    ev[oo] = fmaf(wc[COUT + oo], x0, ev[oo]);
                    od[oo] = fmaf(wc[2 * COUT + oo], x0, od[oo]);
                    od[oo] = fmaf(wc[oo], x1, od[oo]);
                }
            }
        }
#pragma unroll
        for (int oo = 0; oo < OG; ++oo) {
            if (o0 + oo < COUT) {
                float e = ev[oo], o = od[oo];
                if (!LAST) { e = fmaxf(e, 0.0f); o = fmaxf(o, 0.0f); }
                float* dst = out + (size_t)t * out_t_stride + (size_t)(o0 + oo) * lo + 2 * i;
                dst[0] = e;
                dst[1] = o;
            }
        }
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// LDS plan (floats), C = compressed bins, per chunk of TT <= 3 steps.
struct CellLds {
    int x, d0, d1, d2, d3, h, gh, hi, u0, u1, u2, total;
    __host__ __device__ explicit CellLds(int C) {
        const int T = kCellChunk, F = 16 * C;
        int o = 0;
        x = o;  o += T * F;
        d0 = o; o += T * kHidden * 8 * C;
        d1 = o; o += T * kHidden * 4 * C;
        d2 = o; o += T * kHidden * 2 * C;
        d3 = o; o += T * kGates * C;
        h = o;  o += kHidden * C;
        gh = o; o += kGates * C;
        hi = o; o += T * kHidden * C;
        u0 = o; o += T * kHidden * 2 * C;
        u1 = o; o += T * kHidden * 4 * C;
        u2 = o; o += T * kHidden * 8 * C;
        total = o;
    }
};

constexpr int kCellLdsFloats = 3 * 16 * kMaxC + 3 * 17 * 14 * kMaxC * 2 + 3 * 51 * kMaxC + 17 * kMaxC + 51 * kMaxC + 3 * 17 * kMaxC;

__global__ __launch_bounds__(kCellThreads) void cell_kernel(CellDev cd, const float* __restrict__ x,
                                                            const float* __restrict__ hx_in, float* __restrict__ out,
                                                            float* __restrict__ hx_out, int T, int C) {
    __shared__ float lds[kCellLdsFloats];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t b = blockIdx.x;
    const int F = 16 * C;
    const CellLds L(C);
    float* sx = lds + L.x;   float* sd0 = lds + L.d0; float* sd1 = lds + L.d1; float* sd2 = lds + L.d2;
    float* sd3 = lds + L.d3; float* sh = lds + L.h;   float* sgh = lds + L.gh; float* shi = lds + L.hi;
    float* su0 = lds + L.u0; float* su1 = lds + L.u1; float* su2 = lds + L.u2;

    // hidden state -> LDS (gruunet2.py:294-301: zeros when the caller passes none)
    for (int i = tid; i < kHidden * C; i += kCellThreads) sh[i] = hx_in != nullptr ? hx_in[b * kHidden * C + i] : 0.0f;

    // bottleneck mapping: thread <-> (gate channel o, position p); its 51 recurrent weights stay in VGPRs
    const int gate_items = kGates * C;
    const int g_o = tid / C, g_p = tid - g_o * C;
    const bool g_on = tid < gate_items;
    float wgh[kHidden * 3];
#pragma unroll
    for (int i = 0; i < kHidden * 3; ++i) wgh[i] = g_on ? cd.w_gh[i * kGates + g_o] : 0.0f;
    const float bgh = g_on ? cd.bt_gh[g_o * C + g_p] : 0.0f;

    for (int t0 = 0; t0 < T; t0 += kCellChunk) {
        const int tt = min(kCellChunk, T - t0);
        __syncthreads();
        for (int i = tid; i < tt * F; i += kCellThreads) sx[i] = x[(b * T + t0) * F + i];
        __syncthreads();
        // ---- encoder, batched over the chunk (gruunet2.py:136-144)
        conv_down<1, kHidden, 5>((cfloat_ptr)cd.w_down[0], cd.bt_down[0], sx, sd0, 8 * C, tt, wv, lane);
        __syncthreads();
        conv_down<kHidden, kHidden, 5>((cfloat_ptr)cd.w_down[1], cd.bt_down[1], sd0, sd1, 4 * C, tt, wv, lane);
        __syncthreads();
        conv_down<kHidden, kHidden, 5>((cfloat_ptr)cd.w_down[2], cd.bt_down[2], sd1, sd2, 2 * C, tt, wv, lane);
        __syncthreads();
        conv_down<kHidden, kGates, 13>((cfloat_ptr)cd.w_down[3], cd.bt_down[3], sd2, sd3, C, tt, wv, lane);
        __syncthreads();
        // ---- recurrent part, sequential in t (gruunet2.py:232-240)
        for (int t = 0; t < tt; ++t) {
            if (g_on) {            // gh = relu(conv k3 s1 p1 (hx) + position bias)
                float acc = bgh;
#pragma unroll
                for (int c = 0; c < kHidden; ++c) {
                    const float* hc = sh + c * C + g_p;
                    const float x0 = g_p > 0 ? hc[-1] : 0.0f;
                    const float x1 = hc[0];
                    const float x2 = g_p + 1 < C ? hc[1] : 0.0f;
                    acc = fmaf(wgh[c * 3 + 0], x0, acc);
                    acc = fmaf(wgh[c * 3 + 1], x1, acc);
                    acc = fmaf(wgh[c * 3 + 2], x2, acc);
                }
                sgh[tid] = fmaxf(acc, 0.0f);
            }
            __syncthreads();
            if (tid < kHidden * C) {   // chunk order r, i, n (gruunet2.py:234-240)
                const float* gx = sd3 + (size_t)t * kGates * C;
                const float r = sigmoidf_(gx[tid] + sgh[tid]);
                const float z = sigmoidf_(gx[kHidden * C + tid] + sgh[kHidden * C + tid]);
                const float n = tanhf(gx[2 * kHidden * C + tid] + r * sgh[2 * kHidden * C + tid]);
                const float hn = n + z * (sh[tid] - n);
                sh[tid] = hn;
                shi[t * kHidden * C + tid] = hn;
            }
            __syncthreads();
        }
        // ---- decoder, batched over the chunk (gruunet2.py:184-199); skips are d2, d1, d0, (x unused: last has no cat)
        conv_up<kHidden, 0, kHidden, 5, false>((cfloat_ptr)cd.w_up[0], cd.bt_up[0], shi, nullptr, su0, C, tt, wv, lane, (size_t)kHidden * 2 * C);
        __syncthreads();
        conv_up<kHidden, kHidden, kHidden, 5, false>((cfloat_ptr)cd.w_up[1], cd.bt_up[1], su0, sd2, su1, 2 * C, tt, wv, lane, (size_t)kHidden * 4 * C);
        __syncthreads();
        conv_up<kHidden, kHidden, kHidden, 5, false>((cfloat_ptr)cd.w_up[2], cd.bt_up[2], su1, sd1, su2, 4 * C, tt, wv, lane, (size_t)kHidden * 8 * C);
        __syncthreads();
        // last level: one output channel; spread (t, position) over all four waves
        {
            const int l = 8 * C, items = tt * l;
            cfloat_ptr wgt = (cfloat_ptr)cd.w_up[3];
            for (int it = tid; it < items; it += kCellThreads) {
                const int t = it / l, i = it - t * l;
                float ev = cd.bt_up[3][2 * i], od = cd.bt_up[3][2 * i + 1];
                const bool has_next = i + 1 < l;
                for (int c = 0; c < 2 * kHidden; ++c) {
                    const float* src = c < kHidden ? su2 + ((size_t)t * kHidden + c) * l
                                                   : sd0 + ((size_t)t * kHidden + (c - kHidden)) * l;
                    const float x0 = src[i];
                    const float x1 = has_next ? src[i + 1] : 0.0f;
                    ev = fmaf(wgt[c * 3 + 1], x0, ev);
                    od = fmaf(wgt[c * 3 + 2], x0, od);
                    od = fmaf(wgt[c * 3 + 0], x1, od);
                }
                float* dst = out + (b * T + t0 + t) * F + 2 * i;
                *reinterpret_cast<float2*>(dst) = make_float2(ev, od);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < kHidden * C; i += kCellThreads) hx_out[b * kHidden * C + i] = sh[i];
}

void launch_cell(const CellDev& c, const float* x, const float* hx_in, float* out, float* hx_out, int B, int T,
                 int C, hipStream_t st) {
    hipLaunchKernelGGL(cell_kernel, dim3(B), dim3(kCellThreads), 0, st, c, x, hx_in, out, hx_out, T, C);
}

}  // namespace dn
