// dn_momo.hip -- MOMO3.forward, the sibling model of GRUUNet2, on the same fp32 MFMA conv tiles.
// Replaces momo3.py:103-157 (DownBlocks), 159-189 (UpBlocks), 191-245 (MOMOCell.forward), 266-324 (MOMO3._momo / forward)
// for the architecture of the reference's checkpoint saves/MOMO3-4d4ea0: 3 levels, hidden 16, kernel 3, stride 2,
// paddings (1, 0, 1), 6 gaussians, in_size 1 (22 bins -> 11 -> 5 -> 3 compressed bins).
//
// What differs from GRUUNet2 (dn_cell_body.hpp):
//   * a second data channel, the frame delta x_t - prev (momo3.py:285-289; prev starts as x_t, so the first delta is 0,
//     and is carried by the CALLER across forward() calls: momo3.py:300,318);
//   * the Gaussian position code enters once, at the encoder input and at the hidden-gate conv (momo3.py:138-145); the other
//     encoder levels and the whole decoder convolve data channels only -- so only two levels have position-dependent bias
//     tables, the rest a plain bias;
//   * per-level paddings, so lengths are not powers of two: every conv is the GENERAL tile below (any length, stride 1|2,
//     padding 0|1, ConvTranspose with output_size = length of the skip, momo3.py:185-187).
// One workgroup (4 wavefronts) per stream runs all T steps out of ~10 KB of LDS.  Every conv level is
//     out[o][j] = bias[o][j] + sum_(c,k) W[o][c][k] * gather(x, c, j, k)
// on v_mfma_f32_16x16x4_f32 (exact fp32: the 1e-4 parity bar holds): rows = 16 output channels, columns = 16 output
// positions, K slots = (channel, tap) pairs in fours; the weight (A) fragments are packed on the host in lane order, the
// B fragments are the gather view of the LDS activations.  hidden = 16 is exactly one row tile, the gates three.
#include <math.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "dn_internal.hpp"

namespace dn {

constexpr int kMomoH = 16, kMomoGates = 48, kMomoMaxF = 64, kMomoMaxC = 8, kMomoThreads = 256;

struct MomoLevel {
    const float* af;     // [m-tile][k-step][64] weight fragments (data channels only)
    const float* bt;     // [cout][lout] bias (+ folded position code where the level has one)
    int cin, cout, lin, lout, stride, pad;
};

struct MomoDev {
    MomoLevel enc[3], gh, dec[3];
    int F, L1, L2, C;
};

__device__ __forceinline__ f32x4 mfma16m(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// One conv level.  in: [cin][in_ld] (LDS), out: [cout][out_ld] (LDS or global).  UP = ConvTranspose1d (stride 2):
// out[j] += w[c][o][k] x[c][i] for j = 2 i - pad + k.  Down: out[j] += w[o][c][k] x[c][stride j - pad + k].
template <bool UP, bool RELU>
__device__ __forceinline__ void gconv(const MomoLevel& lv, const float* in, int in_ld, float* out, int out_ld, int wv, int lane) {
    const int ks_n = (lv.cin * 3 + 3) >> 2, mtiles = (lv.cout + 15) >> 4, ntiles = (lv.lout + 15) >> 4;
    const int q = lane >> 4, jl = lane & 15;
    for (int tile = wv; tile < mtiles * ntiles; tile += kMomoThreads / 64) {
        const int mt = tile / ntiles, nt = tile - mt * ntiles;
        const int j = nt * 16 + jl;
        const bool valid = j < lv.lout;
        f32x4 acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = mt * 16 + q * 4 + r;
            acc[r] = (valid && o < lv.cout) ? lv.bt[o * lv.lout + j] : 0.0f;
        }
        const float* af = lv.af + (size_t)mt * ks_n * 64 + lane;
        for (int ks = 0; ks < ks_n; ++ks) {
            const int kk = 4 * ks + q, c = kk / 3, k = kk - 3 * c;
            float b = 0.0f;
            if (valid && c < lv.cin) {
                if (UP) {
                    const int num = j + lv.pad - k;
                    if (num >= 0 && !(num & 1) && (num >> 1) < lv.lin) b = in[c * in_ld + (num >> 1)];
                } else {
                    const int p = lv.stride * j - lv.pad + k;
                    if (p >= 0 && p < lv.lin) b = in[c * in_ld + p];
                }
            }
            acc = mfma16m(af[ks * 64], b, acc);
        }
        if (valid) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = mt * 16 + q * 4 + r;
                if (o < lv.cout) out[o * out_ld + j] = RELU ? fmaxf(acc[r], 0.0f) : acc[r];
            }
        }
    }
}

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(kMomoThreads) void momo_kernel(MomoDev md, const float* __restrict__ x, const float* __restrict__ hx_in,
                                                          const float* __restrict__ prev_in, float* __restrict__ out,
                                                          float* __restrict__ hx_out, float* __restrict__ prev_out, int T) {
    __shared__ float x2[2 * kMomoMaxF];                       // [x_t ; x_t - prev]
    __shared__ float prev[kMomoMaxF];
    __shared__ float u2in[32 * (kMomoMaxF / 2)];              // rows 0..15 relu(ups.1 out), rows 16..31 d0      (length L1)
    __shared__ float u1in[32 * (kMomoMaxF / 4)];              // rows 0..15 relu(ups.0 out), rows 16..31 d1      (length L2)
    __shared__ float d2[kMomoGates * kMomoMaxC], gh[kMomoGates * kMomoMaxC], hs[kMomoH * kMomoMaxC];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t b = blockIdx.x;
    const int F = md.F, L1 = md.L1, L2 = md.L2, C = md.C;
    for (int i = tid; i < kMomoH * C; i += kMomoThreads) hs[i] = hx_in != nullptr ? hx_in[b * kMomoH * C + i] : 0.0f;    // momo3.py:309-316
    if (prev_in != nullptr)
        for (int i = tid; i < F; i += kMomoThreads) prev[i] = prev_in[b * F + i];
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        for (int i = tid; i < F; i += kMomoThreads) {
            const float v = x[(b * T + t) * F + i];
            const float pv = (t == 0 && prev_in == nullptr) ? v : prev[i];        // prev = x_t.clone() when none is given (momo3.py:277-278)
            x2[i] = v;
            x2[F + i] = v - pv;                                                   // momo3.py:285
            prev[i] = v;                                                          // momo3.py:289
        }
        __syncthreads();
        gconv<false, true>(md.enc[0], x2, F, u2in + 16 * L1, L1, wv, lane);                 // d0
        __syncthreads();
        gconv<false, true>(md.enc[1], u2in + 16 * L1, L1, u1in + 16 * L2, L2, wv, lane);    // d1
        __syncthreads();
        gconv<false, true>(md.enc[2], u1in + 16 * L2, L2, d2, C, wv, lane);                 // input gates (48 channels)
        gconv<false, true>(md.gh, hs, C, gh, C, wv, lane);                                  // hidden gates relu(conv(hx ; position code))
        __syncthreads();
        if (tid < kMomoH * C) {                                                             // chunk order r, i, n (momo3.py:229-240)
            const float r = sigm(d2[tid] + gh[tid]);
            const float z = sigm(d2[kMomoH * C + tid] + gh[kMomoH * C + tid]);
            const float n = tanhf(d2[2 * kMomoH * C + tid] + r * gh[2 * kMomoH * C + tid]);
            hs[tid] = n + z * (hs[tid] - n);
        }
        __syncthreads();
        gconv<true, true>(md.dec[0], hs, C, u1in, L2, wv, lane);                            // cat(relu(.), d1) is the layout of u1in
        __syncthreads();
        gconv<true, true>(md.dec[1], u1in, L2, u2in, L1, wv, lane);                         // cat(relu(.), d0)
        __syncthreads();
        gconv<true, false>(md.dec[2], u2in, L1, out + (b * T + t) * F, F, wv, lane);        // last level: linear, no cat (momo3.py:94-98)
        __syncthreads();
    }
    for (int i = tid; i < kMomoH * C; i += kMomoThreads) hx_out[b * kMomoH * C + i] = hs[i];
    if (prev_out != nullptr)
        for (int i = tid; i < F; i += kMomoThreads) prev_out[b * F + i] = prev[i];
}

}  // namespace dn

// ------------------------------------------------------------------ host side: packing + C ABI
namespace {

thread_local std::string g_momo_err;
int mfail(int code, const std::string& msg);

struct MomoLayout {     // state_dict order of momo3.MOMO3 (offsets in floats)
    size_t dw[3], db[3], off_in, gw, gb, off_rs, uw[3], ub[3], total;
    MomoLayout() {
        size_t o = 0;
        const int dco[3] = {16, 16, 48}, dci[3] = {8, 16, 16};
        for (int l = 0; l < 3; ++l) { dw[l] = o; o += (size_t)dco[l] * dci[l] * 3; db[l] = o; o += dco[l]; }
        off_in = o; o += 6;
        gw = o; o += 48 * 22 * 3; gb = o; o += 48;
        off_rs = o; o += 6;
        const int uci[3] = {16, 32, 32}, uco[3] = {16, 16, 1};
        for (int l = 0; l < 3; ++l) { uw[l] = o; o += (size_t)uci[l] * uco[l] * 3; ub[l] = o; o += uco[l]; }
        total = o;
    }
};

struct MomoTables {
    float* dev = nullptr;
    dn::MomoDev view;
};

}  // namespace

struct dn_momo {
    dn_momo_cfg cfg;
    std::vector<float> w;
    float* packed = nullptr;               // weight fragments of the seven conv levels
    size_t off_af[7];
    std::mutex mu;
    std::map<int, MomoTables*> tables;     // per input length F (bias tables depend on the lengths)
};

const char* dn_last_error_set(const char* msg);     // dn_api.hip: sets the calling thread's error string

namespace {

int mfail(int code, const std::string& msg) {
    dn_last_error_set(msg.c_str());
    return code;
}

// torch.linspace(0, 1, n) in fp32 (the CPU kernel's symmetric formula) and the GaussianSmearing table (momo3.py:54-68)
std::vector<float> lin01(int n) {
    std::vector<float> v(n);
    if (n == 1) { v[0] = 0.0f; return v; }
    const float step = 1.0f / (float)(n - 1);
    for (int i = 0; i < n; ++i) v[i] = i < n / 2 ? step * (float)i : 1.0f - step * (float)(n - 1 - i);
    return v;
}
std::vector<float> smear(const float* offset, int L) {
    // the reference fixes coeff at construction from linspace(0, 1, 6) (momo3.py:62-63); the loaded offsets enter only in `dist - offset`
    const std::vector<float> o6 = lin01(6);
    const float diff = o6[1] - o6[0];
    const float cf = (float)(-0.5 / ((double)diff * (double)diff));
    std::vector<float> pos = lin01(L), s((size_t)6 * L);
    for (int g = 0; g < 6; ++g)
        for (int l = 0; l < L; ++l) {
            const float d = pos[l] - offset[g];
            s[(size_t)g * L + l] = expf(cf * (d * d));
        }
    return s;
}

// A fragments of one level: lane l of k-step ks supplies W[o = 16 mt + (l & 15)][K slot 4 ks + (l >> 4)], slot = 3 c + tap
std::vector<float> pack_level(const float* W, bool transposed, int cin_data, int cin_total, int cout) {
    const int ks_n = (cin_data * 3 + 3) / 4, mtiles = (cout + 15) / 16;
    std::vector<float> p((size_t)mtiles * ks_n * 64, 0.0f);
    for (int mt = 0; mt < mtiles; ++mt)
        for (int ks = 0; ks < ks_n; ++ks)
            for (int ln = 0; ln < 64; ++ln) {
                const int o = mt * 16 + (ln & 15), kk = 4 * ks + (ln >> 4), c = kk / 3, k = kk % 3;
                if (o < cout && c < cin_data)
                    p[((size_t)mt * ks_n + ks) * 64 + ln] = transposed ? W[((size_t)c * cout + o) * 3 + k] : W[((size_t)o * cin_total + c) * 3 + k];
            }
    return p;
}

int conv_len(int L, int pad, int stride) { return (L + 2 * pad - 3) / stride + 1; }

int build_tables(dn_momo* m, int F, MomoTables** out) {
    std::lock_guard<std::mutex> lock(m->mu);
    auto it = m->tables.find(F);
    if (it != m->tables.end()) { *out = it->second; return DN_OK; }
    const MomoLayout ml;
    const float* W = m->w.data();
    const int* pd = m->cfg.paddings;
    const int L1 = conv_len(F, pd[0], 2), L2 = conv_len(L1, pd[1], 2), C = conv_len(L2, pd[2], 2);
    std::vector<float> host;
    size_t off_bt[7];
    auto add = [&](const std::vector<float>& v) { size_t o = host.size(); host.insert(host.end(), v.begin(), v.end()); return o; };
    // encoder level 0: 2 data channels + 6 position-code channels folded into the bias (stride 2)
    {
        std::vector<float> S = smear(W + ml.off_in, F), bt((size_t)16 * L1);
        for (int o = 0; o < 16; ++o)
            for (int j = 0; j < L1; ++j) {
                double acc = W[ml.db[0] + o];
                for (int g = 0; g < 6; ++g)
                    for (int k = 0; k < 3; ++k) {
                        const int p = 2 * j - pd[0] + k;
                        if (p >= 0 && p < F) acc += (double)W[ml.dw[0] + ((size_t)o * 8 + 2 + g) * 3 + k] * S[(size_t)g * F + p];
                    }
                bt[(size_t)o * L1 + j] = (float)acc;
            }
        off_bt[0] = add(bt);
    }
    auto plain = [&](size_t boff, int cout, int lout) {
        std::vector<float> bt((size_t)cout * lout);
        for (int o = 0; o < cout; ++o)
            for (int j = 0; j < lout; ++j) bt[(size_t)o * lout + j] = W[boff + o];
        return add(bt);
    };
    off_bt[1] = plain(ml.db[1], 16, L2);
    off_bt[2] = plain(ml.db[2], 48, C);
    {   // hidden gates: stride 1, pad 1, 16 data + 6 position-code channels
        std::vector<float> S = smear(W + ml.off_rs, C), bt((size_t)48 * C);
        for (int o = 0; o < 48; ++o)
            for (int p = 0; p < C; ++p) {
                double acc = W[ml.gb + o];
                for (int g = 0; g < 6; ++g)
                    for (int k = 0; k < 3; ++k) {
                        const int q = p - 1 + k;
                        if (q >= 0 && q < C) acc += (double)W[ml.gw + ((size_t)o * 22 + 16 + g) * 3 + k] * S[(size_t)g * C + q];
                    }
                bt[(size_t)o * C + p] = (float)acc;
            }
        off_bt[3] = add(bt);
    }
    off_bt[4] = plain(ml.ub[0], 16, L2);
    off_bt[5] = plain(ml.ub[1], 16, L1);
    off_bt[6] = plain(ml.ub[2], 1, F);
    MomoTables* t = new MomoTables();
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&t->dev), host.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(t->dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { if (t->dev) (void)hipFree(t->dev); delete t; return mfail(DN_ERR_HIP, std::string("MOMO3 bias tables: ") + hipGetErrorString(e)); }
    dn::MomoDev& v = t->view;
    v.F = F; v.L1 = L1; v.L2 = L2; v.C = C;
    auto lvl = [&](int idx, int cin, int cout, int lin, int lout, int stride, int pad) {
        dn::MomoLevel l;
        l.af = m->packed + m->off_af[idx]; l.bt = t->dev + off_bt[idx];
        l.cin = cin; l.cout = cout; l.lin = lin; l.lout = lout; l.stride = stride; l.pad = pad;
        return l;
    };
    v.enc[0] = lvl(0, 2, 16, F, L1, 2, pd[0]);
    v.enc[1] = lvl(1, 16, 16, L1, L2, 2, pd[1]);
    v.enc[2] = lvl(2, 16, 48, L2, C, 2, pd[2]);
    v.gh = lvl(3, 16, 48, C, C, 1, 1);
    v.dec[0] = lvl(4, 16, 16, C, L2, 2, pd[2]);       // UpBlocks walks the levels backwards (momo3.py:176-178)
    v.dec[1] = lvl(5, 32, 16, L2, L1, 2, pd[1]);
    v.dec[2] = lvl(6, 32, 1, L1, F, 2, pd[0]);
    m->tables[F] = t;
    *out = t;
    return DN_OK;
}

}  // namespace

extern "C" {

int dn_momo_create(const float* weights, size_t n_floats, const dn_momo_cfg* cfg, dn_momo** out) {
    if (!weights || !cfg || !out) return mfail(DN_ERR_INVALID, "dn_momo_create: null argument");
    const MomoLayout ml;
    if (cfg->in_size != 1 || cfg->n_levels != 3 || cfg->hidden_size != 16 || cfg->kernel_size != 3 || cfg->stride != 2 || cfg->num_gaussians != 6)
        return mfail(DN_ERR_UNSUPPORTED, "MOMO3 kernels are built for in_size 1, 3 levels, hidden 16, k3 s2, 6 gaussians (saves/MOMO3-4d4ea0)");
    for (int l = 0; l < 3; ++l)
        if (cfg->paddings[l] < 0 || cfg->paddings[l] > 1) return mfail(DN_ERR_UNSUPPORTED, "MOMO3 paddings must be 0 or 1");
    if (n_floats != ml.total) return mfail(DN_ERR_INVALID, "expected " + std::to_string(ml.total) + " weight floats, got " + std::to_string(n_floats));
    dn_momo* m = new dn_momo();
    m->cfg = *cfg;
    m->w.assign(weights, weights + n_floats);
    const float* W = m->w.data();
    std::vector<float> host;
    auto add = [&](int idx, const std::vector<float>& v) { m->off_af[idx] = host.size(); host.insert(host.end(), v.begin(), v.end()); };
    add(0, pack_level(W + ml.dw[0], false, 2, 8, 16));
    add(1, pack_level(W + ml.dw[1], false, 16, 16, 16));
    add(2, pack_level(W + ml.dw[2], false, 16, 16, 48));
    add(3, pack_level(W + ml.gw, false, 16, 22, 48));
    add(4, pack_level(W + ml.uw[0], true, 16, 16, 16));
    add(5, pack_level(W + ml.uw[1], true, 32, 32, 16));
    add(6, pack_level(W + ml.uw[2], true, 32, 32, 1));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&m->packed), host.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(m->packed, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { if (m->packed) (void)hipFree(m->packed); delete m; return mfail(DN_ERR_HIP, std::string("MOMO3 weight upload: ") + hipGetErrorString(e)); }
    *out = m;
    return DN_OK;
}

void dn_momo_destroy(dn_momo* m) {
    if (!m) return;
    for (auto& kv : m->tables) { (void)hipFree(kv.second->dev); delete kv.second; }
    if (m->packed) (void)hipFree(m->packed);
    delete m;
}

int dn_momo_forward(const dn_momo* m, const float* x, const float* hx_in, const float* prev_in, float* out, float* hx_out,
                    float* prev_out, int32_t B, int32_t T, int32_t F, int32_t C, void* stream) {
    if (B == 0) return DN_OK;
    if (!m || !x || !out || !hx_out) return mfail(DN_ERR_INVALID, "dn_momo_forward: null argument");
    if (B < 0 || T < 0) return mfail(DN_ERR_INVALID, "dn_momo_forward: negative size");
    if (F < 3 || F > dn::kMomoMaxF) return mfail(DN_ERR_UNSUPPORTED, "MOMO3 kernel supports 3..64 input bins");
    const int* pd = m->cfg.paddings;
    const int L1 = conv_len(F, pd[0], 2), L2 = L1 >= 1 ? conv_len(L1, pd[1], 2) : 0, Cx = L2 >= 1 ? conv_len(L2, pd[2], 2) : 0;
    if (L1 < 1 || L2 < 1 || Cx < 1) return mfail(DN_ERR_INVALID, "input is too short for three stride-2 levels");
    if (Cx != C)
        return mfail(DN_ERR_INVALID, "input of " + std::to_string(F) + " bins compresses to " + std::to_string(Cx) + ", but hx has " +
                                         std::to_string(C) + " compressed bins");
    if (C > dn::kMomoMaxC) return mfail(DN_ERR_UNSUPPORTED, "too many compressed bins");
    // the decoder must be able to reach the skip lengths with output_padding 0 or 1 (ConvTranspose1d output_size, momo3.py:185-187)
    const int up0 = (C - 1) * 2 - 2 * pd[2] + 3, up1 = (L2 - 1) * 2 - 2 * pd[1] + 3, up2 = (L1 - 1) * 2 - 2 * pd[0] + 3;
    if (L2 - up0 < 0 || L2 - up0 > 1 || L1 - up1 < 0 || L1 - up1 > 1 || F - up2 < 0 || F - up2 > 1)
        return mfail(DN_ERR_INVALID, "requested output size is not reachable by the transposed convs (as the reference raises)");
    MomoTables* t = nullptr;
    int rc = build_tables(const_cast<dn_momo*>(m), F, &t);
    if (rc != DN_OK) return rc;
    hipLaunchKernelGGL(dn::momo_kernel, dim3(B), dim3(dn::kMomoThreads), 0, reinterpret_cast<hipStream_t>(stream), t->view, x, hx_in, prev_in, out,
                       hx_out, prev_out, T);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mfail(DN_ERR_HIP, std::string("momo_kernel launch: ") + hipGetErrorString(e));
    return DN_OK;
}

}  // extern "C"
