// dn_api.hip -- the C ABI of include/dn_denoise.h: handle construction (host-side packing of
// the reference's tensors into kernel-friendly layouts) and launch wrappers.  No kernel code here.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "dn_internal.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define DN_HIP(call)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) return fail(DN_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DN_ERR_HIP, std::string(what) + " launch: " + hipGetErrorString(e));
    return DN_OK;
}

// One device allocation holding several arrays, 256-byte aligned slots.
struct DevArena {
    char* base = nullptr;
    size_t size = 0;
    std::vector<char> host;
    size_t add(const void* src, size_t bytes) {
        size_t off = (host.size() + 255) & ~size_t(255);
        host.resize(off + bytes);
        memcpy(host.data() + off, src, bytes);
        return off;
    }
    hipError_t upload() {
        host.resize(host.size() + dn::kArenaSlack);     // zeros: kernels that move whole rounds of an array read past its end (dn_cell_body.hpp)
        size = host.size();
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&base), size ? size : 256);
        if (e != hipSuccess) return e;
        e = hipMemcpy(base, host.data(), size, hipMemcpyHostToDevice);
        host.clear();
        host.shrink_to_fit();
        return e;
    }
    template <typename T>
    const T* ptr(size_t off) const { return reinterpret_cast<const T*>(base + off); }
    void release() {
        if (base) (void)hipFree(base);
        base = nullptr;
    }
};

// ------------------------------------------------------------------ model
// state_dict order, SURVEY.md Appendix A.4 (offsets in floats)
struct StateLayout {
    size_t dw[4], db[4], off_in, gw, gb, off_rs, uw[4], ub[4], off_out, total;
    StateLayout() {
        size_t o = 0;
        const int dco[4] = {17, 17, 17, 51}, dci[4] = {7, 23, 23, 23};
        for (int l = 0; l < 4; ++l) { dw[l] = o; o += (size_t)dco[l] * dci[l] * 3; db[l] = o; o += dco[l]; }
        off_in = o; o += 6;
        gw = o; o += 51 * 23 * 3; gb = o; o += 51;
        off_rs = o; o += 6;
        const int uci[4] = {23, 40, 40, 40}, uco[4] = {17, 17, 17, 1};
        for (int l = 0; l < 4; ++l) { uw[l] = o; o += (size_t)uci[l] * uco[l] * 3; ub[l] = o; o += uco[l]; }
        off_out = o; o += 6;
        total = o;
    }
};

// torch.linspace(0, 1, n) in fp32 (symmetric two-sided formula of the CPU kernel)
std::vector<float> linspace01(int n) {
    std::vector<float> v(n);
    if (n == 1) { v[0] = 0.0f; return v; }
    const float step = 1.0f / (float)(n - 1);
    const int half = n / 2;
    for (int i = 0; i < n; ++i) v[i] = i < half ? 0.0f + step * (float)i : 1.0f - step * (float)(n - 1 - i);
    return v;
}

// GaussianSmearing table S[g][l], gruunet2.py:54-68 on linspace(0,1,L).  The reference fixes `coeff` at construction from its own
// linspace(0, 1, num_gaussians) (gruunet2.py:62-63) and never recomputes it after load_state_dict: only `dist - offset` uses the
// loaded buffer.
std::vector<float> smear_table(const float* offset, int L) {
    const std::vector<float> o6 = linspace01(dn::kGauss);
    const float diff = o6[1] - o6[0];
    const double coeff = -0.5 / ((double)diff * (double)diff);
    const float cf = (float)coeff;
    std::vector<float> pos = linspace01(L), s((size_t)dn::kGauss * L);
    for (int g = 0; g < dn::kGauss; ++g)
        for (int l = 0; l < L; ++l) {
            const float d = pos[l] - offset[g];
            s[(size_t)g * L + l] = expf(cf * (d * d));
        }
    return s;
}

struct BiasSet {
    DevArena arena;
    dn::CellDev view;
    dn::CellDev* view_dev = nullptr;      // the same view in device memory (group_kernel reads its pointers where it needs them: no kernel-argument copy)
};

}  // namespace

struct dn_model {
    std::atomic<int> refs{1};  // the creator's reference + one per dn_pipe bound to it (dn_model_destroy only drops the creator's)
    dn_model_cfg cfg;
    std::vector<float> w;      // the caller's flat state_dict
    DevArena packed;           // MFMA weight fragments (fp32 and bf16), recurrent and last-level weights
    size_t off_down[4], off_gh, off_up[4];
    size_t offb_down[4], offb_up[3];
    std::mutex mu;
    std::map<int, BiasSet*> bias;   // per number of compressed bins C
};

struct dn_dsp {
    std::atomic<int> refs{1};
    dn_dsp_cfg cfg;
    DevArena arena;
    dn::DspDev view;
    dn::DspDev* view_dev = nullptr;        // the same view in device memory (group_kernel)
    std::vector<float> fb, pinv, window;   // host copies [K][M], [K][M], [N]
};

// host-buffer transport of a streaming pipe (dn_pipe_stream_push_host): two copy queues beside the caller's compute stream, device staging
// double-buffered, everything ordered by events on the device -- the host never blocks inside a push
struct HostIo {
    hipStream_t h2d = nullptr, d2h = nullptr;
    static constexpr int kRing = 4;           // pushes whose events are kept (the host may run this far ahead of the result it waits for)
    hipEvent_t ev_h2d[kRing] = {}, ev_k[kRing] = {}, ev_d2h[kRing] = {};
    void* d_in[2] = {nullptr, nullptr};
    void* d_out[2] = {nullptr, nullptr};
    unsigned long long pushes = 0;
    // zero copy: the launch itself publishes "push n is in host memory" in this page-locked word (no HIP event, no barrier packet between hops)
    unsigned long long* done_host = nullptr;
    unsigned long long* done_dev = nullptr;
    unsigned long long zero_copy_pushes = 0;       // pushes whose completion is published there (the others: ev_d2h)
    bool last_zero_copy[kRing] = {};
    // DN_HOST_DEFER: the emitted hop of the newest push waits in z_out[ticket & 1] until the next launch (or dn_pipe_stream_host_wait) moves it out
    void* z_out[2] = {nullptr, nullptr};
    bool defer_pending = false;
    unsigned long long defer_ticket = 0;
    void* defer_dst = nullptr;                     // device view of that push's hop_out_host
    size_t defer_bytes = 0;
    hipStream_t defer_stream = nullptr;
};

struct dn_pipe {
    HostIo* hio = nullptr;
    dn_model* m = nullptr;
    dn_dsp* d = nullptr;
    int B = 0, C = 0;
    bool bf16 = false;                        // DN_CONV_BF16: bf16 MFMA conv tiles in the front half
    dn::PipeCtl* ctl = nullptr;               // device-resident hop counter / pending flag (what makes a captured launch replayable)
    // scratch slots, used round robin by consecutive frames: depth + 1 of them (a frame's slot is read by every segment of its chain)
    int depth = 1, n_slots = 2;                     // hops of one stream in flight (dn_pipe_set_depth)
    float* scratch = nullptr;                       // [n_slots] x { mel [B][3][M], residual [B][3][M], peak [B], meta [B][kSlotMeta], lin [B][3][K] }
    float2* scratch_init = nullptr;                 // [n_slots] x the frame's initial phases [B][3][K] complex (allocated on first parity-mode use)
    float2* gl_state = nullptr;                     // [n_slots] x a parked Griffin-Lim chain ([B][3][2 NV + 2][64] complex: head start, chain segments)
    size_t slot_floats = 0, init_elems = 0, state_elems = 0;
    int group = 0;                                  // hop groups (dn_pipe_set_group): hops a launch carries, 0 = single hops
    unsigned long long group_pushes = 0;            // hops pushed into a streaming group pipe by this host (priming bookkeeping of *hops_valid)
    int group_pending = 0;                          // frames the last group push fronted, as far as this host's own calls tell
    int gl_split = 0;                               // iterations of head start (0 = none)
    int gl_schedule = DN_GL_AUTO;                   // Griffin-Lim schedule of the back half (dn_pipe_set_gl_schedule)
    int split = DN_SPLIT_AUTO;                      // a hop as two launches, chains then front halves (dn_pipe_set_split)
    BiasSet* bs = nullptr;
    bool submitted = false;                   // frame mode: a hop may be pending (its destination travels in the slot, not here)
    // streaming mode: per-stream state owned by the pipe
    unsigned long long* host_done = nullptr;  // set around a zero-copy host push (dn_pipe_stream_push_host)
    unsigned long long host_done_value = 0;
    const void* host_copy_src = nullptr;      // (likewise: the deferred output the launch carries to the host)
    void* host_copy_dst = nullptr;
    size_t host_copy_bytes = 0;
    float* ring = nullptr;                    // [B][n_fft] last n_fft input samples
    float* ola = nullptr;                     // [B][n_fft] output overlap-add line
    float* hx = nullptr;                      // [B][17][C]
};

namespace {

int build_bias(dn_model* m, int C, BiasSet** out) {
    std::lock_guard<std::mutex> lock(m->mu);
    auto it = m->bias.find(C);
    if (it != m->bias.end()) { *out = it->second; return DN_OK; }
    const StateLayout sl;
    const float* W = m->w.data();
    const int F = 16 * C;
    BiasSet* bs = new BiasSet();
    size_t o_down[4], o_gh, o_up[4];
    // encoder levels: Lin = F >> l, Lout = Lin/2, stride 2, left pad 1
    for (int l = 0; l < 4; ++l) {
        const int cd = l == 0 ? 1 : 17, ct = cd + 6, co = l == 3 ? 51 : 17;
        const int lin = F >> l, lout = lin / 2;
        std::vector<float> S = smear_table(W + sl.off_in, lin), bt((size_t)co * lout);
        for (int o = 0; o < co; ++o)
            for (int j = 0; j < lout; ++j) {
                double acc = W[sl.db[l] + o];
                for (int g = 0; g < 6; ++g)
                    for (int k = 0; k < 3; ++k) {
                        const int p = 2 * j - 1 + k;
                        if (p >= 0 && p < lin) acc += (double)W[sl.dw[l] + ((size_t)o * ct + cd + g) * 3 + k] * S[(size_t)g * lin + p];
                    }
                bt[(size_t)o * lout + j] = (float)acc;
            }
        o_down[l] = bs->arena.add(bt.data(), bt.size() * sizeof(float));
    }
    {   // hidden gates: stride 1, pad 1 on both sides, length C
        std::vector<float> S = smear_table(W + sl.off_rs, C), bt((size_t)51 * C);
        for (int o = 0; o < 51; ++o)
            for (int p = 0; p < C; ++p) {
                double acc = W[sl.gb + o];
                for (int g = 0; g < 6; ++g)
                    for (int k = 0; k < 3; ++k) {
                        const int q = p - 1 + k;
                        if (q >= 0 && q < C) acc += (double)W[sl.gw + ((size_t)o * 23 + 17 + g) * 3 + k] * S[(size_t)g * C + q];
                    }
                bt[(size_t)o * C + p] = (float)acc;
            }
        o_gh = bs->arena.add(bt.data(), bt.size() * sizeof(float));
    }
    // decoder levels: Lin = C << l, Lout = 2 Lin; output j = 2 i - 1 + k
    for (int l = 0; l < 4; ++l) {
        const int cd = l == 0 ? 17 : 34, co = l == 3 ? 1 : 17;
        const int lin = C << l, lout = 2 * lin;
        std::vector<float> S = smear_table(W + sl.off_out, lin), bt((size_t)co * lout);
        for (int o = 0; o < co; ++o)
            for (int j = 0; j < lout; ++j) {
                double acc = W[sl.ub[l] + o];
                for (int g = 0; g < 6; ++g)
                    for (int k = 0; k < 3; ++k) {
                        const int num = j + 1 - k;          // 2 i
                        if (num < 0 || (num & 1)) continue;
                        const int i = num >> 1;
                        if (i < lin) acc += (double)W[sl.uw[l] + ((size_t)(cd + g) * co + o) * 3 + k] * S[(size_t)g * lin + i];
                    }
                bt[(size_t)o * lout + j] = (float)acc;
            }
        o_up[l] = bs->arena.add(bt.data(), bt.size() * sizeof(float));
    }
    hipError_t e = bs->arena.upload();
    if (e != hipSuccess) { bs->arena.release(); delete bs; return fail(DN_ERR_HIP, std::string("bias table upload: ") + hipGetErrorString(e)); }
    for (int l = 0; l < 4; ++l) {
        bs->view.w_down[l] = m->packed.ptr<float>(m->off_down[l]);
        bs->view.w_up[l] = m->packed.ptr<float>(m->off_up[l]);
        bs->view.wb_down[l] = m->packed.ptr<char>(m->offb_down[l]);
        if (l < 3) bs->view.wb_up[l] = m->packed.ptr<char>(m->offb_up[l]);
        bs->view.bt_down[l] = bs->arena.ptr<float>(o_down[l]);
        bs->view.bt_up[l] = bs->arena.ptr<float>(o_up[l]);
    }
    bs->view.w_gh = m->packed.ptr<float>(m->off_gh);
    bs->view.bt_gh = bs->arena.ptr<float>(o_gh);
    e = hipMalloc(reinterpret_cast<void**>(&bs->view_dev), sizeof(dn::CellDev));
    if (e == hipSuccess) e = hipMemcpy(bs->view_dev, &bs->view, sizeof(dn::CellDev), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (bs->view_dev) (void)hipFree(bs->view_dev);
        bs->arena.release(); delete bs;
        return fail(DN_ERR_HIP, std::string("bias table view: ") + hipGetErrorString(e));
    }
    m->bias[C] = bs;
    *out = bs;
    return DN_OK;
}

// Cholesky solve of the SPD system G X = R in double; G is n x n (row-major), R is n x m. In place.
bool chol_solve(std::vector<double>& G, std::vector<double>& R, int n, int m) {
    for (int j = 0; j < n; ++j) {
        double d = G[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= G[(size_t)j * n + k] * G[(size_t)j * n + k];
        if (!(d > 0.0)) return false;
        d = sqrt(d);
        G[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = G[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= G[(size_t)i * n + k] * G[(size_t)j * n + k];
            G[(size_t)i * n + j] = s / d;
        }
    }
    for (int c = 0; c < m; ++c) {
        for (int i = 0; i < n; ++i) {           // L y = r
            double s = R[(size_t)i * m + c];
            for (int k = 0; k < i; ++k) s -= G[(size_t)i * n + k] * R[(size_t)k * m + c];
            R[(size_t)i * m + c] = s / G[(size_t)i * n + i];
        }
        for (int i = n - 1; i >= 0; --i) {      // L^T x = y
            double s = R[(size_t)i * m + c];
            for (int k = i + 1; k < n; ++k) s -= G[(size_t)k * n + i] * R[(size_t)k * m + c];
            R[(size_t)i * m + c] = s / G[(size_t)i * n + i];
        }
    }
    return true;
}

hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace

// for the other translation units of the library (dn_momo.hip): set the calling thread's error message
const char* dn_last_error_set(const char* msg) {
    g_err = msg;
    return g_err.c_str();
}

extern "C" {

const char* dn_last_error(void) { return g_err.c_str(); }
int dn_abi_version(void) { return DN_ABI_VERSION; }

int dn_model_create(const float* weights, size_t n_floats, const dn_model_cfg* cfg, dn_model** out) {
    if (!weights || !cfg || !out) return fail(DN_ERR_INVALID, "dn_model_create: null argument");
    const StateLayout sl;
    if (cfg->in_size != 1) return fail(DN_ERR_INVALID, "in_size must be 1 (gruunet2.py:257)");
    if (cfg->n_levels != 4 || cfg->hidden_size != 17 || cfg->kernel_size != 3 || cfg->stride != 2 ||
        cfg->padding != 1 || cfg->num_gaussians != 6)
        return fail(DN_ERR_UNSUPPORTED, "kernels are built for 4 levels, hidden 17, k3 s2 p1, 6 gaussians");
    if (cfg->num_compressed_bins < 1 || cfg->num_compressed_bins > dn::kMaxC)
        return fail(DN_ERR_UNSUPPORTED, "num_compressed_bins must be in 1.." + std::to_string(dn::kMaxC));
    if (n_floats != sl.total) return fail(DN_ERR_INVALID, "expected " + std::to_string(sl.total) + " weight floats, got " + std::to_string(n_floats));
    dn_model* m = new dn_model();
    m->cfg = *cfg;
    m->w.assign(weights, weights + n_floats);
    const float* W = m->w.data();
    // MFMA A fragments (v_mfma_f32_16x16x4_f32): lane l of k-step ks supplies W[o = 16 mt + (l & 15)][K slot 4 ks + (l >> 4)].
    for (int l = 0; l < 4; ++l) {   // Conv1d weight (Cout, Ct, 3): K order tap-major, channels in fours -> [mt][ks][64]
        const int cd = l == 0 ? 1 : 17, ct = cd + 6, co = l == 3 ? 51 : 17;
        const int cs = (cd + 3) / 4, ks_n = cd == 1 ? 1 : 3 * cs, mtiles = (co + 15) / 16;
        std::vector<float> p((size_t)mtiles * ks_n * 64, 0.0f);
        for (int mt = 0; mt < mtiles; ++mt)
            for (int ks = 0; ks < ks_n; ++ks)
                for (int ln = 0; ln < 64; ++ln) {
                    const int o = mt * 16 + (ln & 15), q = ln >> 4;
                    const int tap = cd == 1 ? q : ks / cs, c = cd == 1 ? 0 : 4 * (ks % cs) + q;
                    if (o < co && tap < 3 && c < cd) p[((size_t)mt * ks_n + ks) * 64 + ln] = W[sl.dw[l] + ((size_t)o * ct + c) * 3 + tap];
                }
        m->off_down[l] = m->packed.add(p.data(), p.size() * sizeof(float));
    }
    {   // hidden-gate conv (51, 17 data channels, 3) as A fragments too: [mt 4][ks 13][64], K slot = 3 c + tap (51 slots + 1 zero)
        std::vector<float> p((size_t)4 * 13 * 64, 0.0f);
        for (int mt = 0; mt < 4; ++mt)
            for (int ks = 0; ks < 13; ++ks)
                for (int ln = 0; ln < 64; ++ln) {
                    const int o = mt * 16 + (ln & 15), kk = 4 * ks + (ln >> 4), c = kk / 3, k = kk % 3;
                    if (o < 51 && c < 17) p[((size_t)mt * 13 + ks) * 64 + ln] = W[sl.gw + ((size_t)o * 23 + c) * 3 + k];
                }
        m->off_gh = m->packed.add(p.data(), p.size() * sizeof(float));
    }
    for (int l = 0; l < 4; ++l) {   // ConvTranspose1d weight (Ct, Cout, 3) -> [mt][tap set: k=1,k=2,k=0][ks][64], parts of 17 channels in fours
        static const int kTapOfSet[3] = {1, 2, 0};
        if (l == 3) {   // the last level (one output channel) runs on the VALU: [part: a, skip][17 channels][4] = w[k=1], w[k=2], w[k=0], 0
            std::vector<float> p((size_t)2 * 17 * 4, 0.0f);
            for (int part = 0; part < 2; ++part)
                for (int c = 0; c < 17; ++c)
                    for (int set = 0; set < 3; ++set) p[((size_t)part * 17 + c) * 4 + set] = W[sl.uw[3] + ((size_t)(part * 17 + c)) * 3 + kTapOfSet[set]];
            m->off_up[l] = m->packed.add(p.data(), p.size() * sizeof(float));
            continue;
        }
        const int parts = l == 0 ? 1 : 2, co = 17, ksu = parts * 5, mtiles = (co + 15) / 16;
        std::vector<float> p((size_t)mtiles * 3 * ksu * 64, 0.0f);
        for (int mt = 0; mt < mtiles; ++mt)
            for (int set = 0; set < 3; ++set)
                for (int ks = 0; ks < ksu; ++ks)
                    for (int ln = 0; ln < 64; ++ln) {
                        const int o = mt * 16 + (ln & 15), q = ln >> 4, cp = 4 * (ks % 5) + q, c = (ks / 5) * 17 + cp;
                        if (o < co && cp < 17)
                            p[(((size_t)mt * 3 + set) * ksu + ks) * 64 + ln] = W[sl.uw[l] + ((size_t)c * co + o) * 3 + kTapOfSet[set]];
                    }
        m->off_up[l] = m->packed.add(p.data(), p.size() * sizeof(float));
    }
    // bf16 fragments for v_mfma_f32_16x16x32_bf16 (config 3): lane l of k-step ks supplies 8 consecutive K slots
    // 8 (l >> 4) + j = channels of one tap (level 0: the taps themselves); round to nearest even.
    auto bf16 = [](float f) -> uint16_t {
        uint32_t u; memcpy(&u, &f, 4);
        u += 0x7FFFu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    };
    for (int l = 0; l < 4; ++l) {
        const int cd = l == 0 ? 1 : 17, ct = cd + 6, co = l == 3 ? 51 : 17, ks_n = cd == 1 ? 1 : 3, mtiles = (co + 15) / 16;
        std::vector<uint16_t> p((size_t)mtiles * ks_n * 64 * 8, 0);
        for (int mt = 0; mt < mtiles; ++mt)
            for (int ks = 0; ks < ks_n; ++ks)
                for (int ln = 0; ln < 64; ++ln)
                    for (int j = 0; j < 8; ++j) {
                        const int o = mt * 16 + (ln & 15), q = ln >> 4;
                        const int tap = cd == 1 ? j : ks, c = cd == 1 ? 0 : 8 * q + j;
                        const bool ok = o < co && (cd == 1 ? (q == 0 && j < 3) : c < cd);
                        if (ok) p[(((size_t)mt * ks_n + ks) * 64 + ln) * 8 + j] = bf16(W[sl.dw[l] + ((size_t)o * ct + c) * 3 + tap]);
                    }
        m->offb_down[l] = m->packed.add(p.data(), p.size() * sizeof(uint16_t));
    }
    for (int l = 0; l < 3; ++l) {
        const int parts = l == 0 ? 1 : 2, co = 17;
        static const int kTapOfSet[3] = {1, 2, 0};
        std::vector<uint16_t> p((size_t)2 * 3 * parts * 64 * 8, 0);
        for (int mt = 0; mt < 2; ++mt)
            for (int set = 0; set < 3; ++set)
                for (int part = 0; part < parts; ++part)
                    for (int ln = 0; ln < 64; ++ln)
                        for (int j = 0; j < 8; ++j) {
                            const int o = mt * 16 + (ln & 15), cp = 8 * (ln >> 4) + j, c = part * 17 + cp;
                            if (o < co && cp < 17)
                                p[((((size_t)mt * 3 + set) * parts + part) * 64 + ln) * 8 + j] = bf16(W[sl.uw[l] + ((size_t)c * co + o) * 3 + kTapOfSet[set]]);
                        }
        m->offb_up[l] = m->packed.add(p.data(), p.size() * sizeof(uint16_t));
    }
    hipError_t e = m->packed.upload();
    if (e != hipSuccess) { m->packed.release(); delete m; return fail(DN_ERR_HIP, std::string("weight upload: ") + hipGetErrorString(e)); }
    BiasSet* bs = nullptr;
    int rc = build_bias(m, cfg->num_compressed_bins, &bs);
    if (rc != DN_OK) { m->packed.release(); delete m; return rc; }
    *out = m;
    return DN_OK;
}

static void model_release(dn_model* m) {
    if (!m || m->refs.fetch_sub(1) != 1) return;
    for (auto& kv : m->bias) { kv.second->arena.release(); if (kv.second->view_dev) (void)hipFree(kv.second->view_dev); delete kv.second; }
    m->packed.release();
    delete m;
}

// Drops the creator's reference; the device arenas are freed once no dn_pipe is bound to the model any more.
void dn_model_destroy(dn_model* m) { model_release(m); }

int dn_cell_forward(const dn_model* m, const float* x, const float* hx_in, float* out, float* hx_out, int32_t B,
                    int32_t T, int32_t F, int32_t C, void* stream) {
    if (B == 0) return DN_OK;      // empty batch: nothing to do (pointers of empty tensors are null)
    if (!m || !x || !out || !hx_out) return fail(DN_ERR_INVALID, "dn_cell_forward: null argument");
    if (B < 0 || T < 0) return fail(DN_ERR_INVALID, "dn_cell_forward: negative size");
    if (C < 1 || C > dn::kMaxC) return fail(DN_ERR_UNSUPPORTED, "compressed bins C must be in 1.." + std::to_string(dn::kMaxC));
    if (F != 16 * C)
        return fail(DN_ERR_INVALID, "input has " + std::to_string(F) + " bins, which compress to " + std::to_string(F / 16) +
                                        ", but hx has " + std::to_string(C) + " compressed bins (need F == 16*C)");
    if (B == 0) return DN_OK;
    BiasSet* bs = nullptr;
    int rc = build_bias(const_cast<dn_model*>(m), C, &bs);
    if (rc != DN_OK) return rc;
    dn::launch_cell(bs->view, x, hx_in, out, hx_out, B, T, C, as_stream(stream));
    return check_launch("cell_kernel");
}

int dn_cell_forward_ex(const dn_model* m, const float* x, const float* hx_in, float* out, float* hx_out, int32_t B,
                       int32_t T, int32_t F, int32_t C, float hx_out_scale, void* stream) {
    if (B == 0) return DN_OK;
    if (!m || !x || !out || !hx_out) return fail(DN_ERR_INVALID, "dn_cell_forward_ex: null argument");
    if (B < 0 || T < 0) return fail(DN_ERR_INVALID, "dn_cell_forward_ex: negative size");
    if (C < 1 || C > dn::kMaxC) return fail(DN_ERR_UNSUPPORTED, "compressed bins C must be in 1.." + std::to_string(dn::kMaxC));
    if (F != 16 * C) return fail(DN_ERR_INVALID, "need F == 16*C");
    BiasSet* bs = nullptr;
    int rc = build_bias(const_cast<dn_model*>(m), C, &bs);
    if (rc != DN_OK) return rc;
    dn::launch_cell_ex(bs->view, x, hx_in, out, hx_out, B, T, C, hx_out_scale, as_stream(stream));
    return check_launch("cell_kernel_ex");
}

int dn_cell_forward_bf16(const dn_model* m, const float* x, const float* hx_in, float* out, float* hx_out, int32_t B,
                         int32_t T, int32_t F, int32_t C, void* stream) {
    if (B == 0) return DN_OK;
    if (!m || !x || !out || !hx_out) return fail(DN_ERR_INVALID, "dn_cell_forward_bf16: null argument");
    if (B < 0 || T < 0) return fail(DN_ERR_INVALID, "dn_cell_forward_bf16: negative size");
    if (C < 1 || C > dn::kMaxC) return fail(DN_ERR_UNSUPPORTED, "compressed bins C must be in 1.." + std::to_string(dn::kMaxC));
    if (F != 16 * C) return fail(DN_ERR_INVALID, "need F == 16*C");
    BiasSet* bs = nullptr;
    int rc = build_bias(const_cast<dn_model*>(m), C, &bs);
    if (rc != DN_OK) return rc;
    dn::launch_cell_bf16(bs->view, x, hx_in, out, hx_out, B, T, C, as_stream(stream));
    return check_launch("cell_kernel_bf16");
}

int dn_dsp_create(const dn_dsp_cfg* cfg, const float* fb_in, const float* pinv_in, const float* window_in, dn_dsp** out) {
    if (!cfg || !out) return fail(DN_ERR_INVALID, "dn_dsp_create: null argument");
    if ((cfg->n_fft != 1024 && cfg->n_fft != 1536) || cfg->hop != cfg->n_fft / 2)
        return fail(DN_ERR_UNSUPPORTED, "kernels are built for n_fft 1024 or 1536 with hop = n_fft/2 (got n_fft " +
                                            std::to_string(cfg->n_fft) + ", hop " + std::to_string(cfg->hop) + ")");
    const int N = cfg->n_fft, K = N / 2 + 1, M = cfg->n_mels;
    if (M < 0 || M > 128) return fail(DN_ERR_UNSUPPORTED, "n_mels must be in 0..128");
    if (M > 0 && cfg->sample_rate <= 0) return fail(DN_ERR_INVALID, "sample_rate must be positive");
    dn_dsp* d = new dn_dsp();
    d->cfg = *cfg;
    const double PI = 3.14159265358979323846;
    // window (torch.hann_window periodic) and istft envelope reciprocal
    d->window.resize(N);
    for (int n = 0; n < N; ++n) d->window[n] = window_in ? window_in[n] : (float)(0.5 - 0.5 * cos(2.0 * PI * n / N));
    std::vector<float> inv_env(N);
    for (int i = 0; i < N; ++i) {
        const float a = d->window[i], b = d->window[(i + N / 2) % N];
        const float env = a * a + b * b;
        if (!(env > 1e-11f)) { delete d; return fail(DN_ERR_INVALID, "window overlap-add envelope is ~0 (torch.istft would raise)"); }
        inv_env[i] = (float)(1.0 / (double)env);
    }
    const int NC = N / 2;
    std::vector<float> tw512(2 * NC), tw1024(2 * (NC / 2 + 1));     // exp(-2 pi i k / NC), exp(-2 pi i k / n_fft)
    for (int k = 0; k < NC; ++k) { tw512[2 * k] = (float)cos(2.0 * PI * k / NC); tw512[2 * k + 1] = (float)-sin(2.0 * PI * k / NC); }
    for (int k = 0; k <= NC / 2; ++k) { tw1024[2 * k] = (float)cos(2.0 * PI * k / N); tw1024[2 * k + 1] = (float)-sin(2.0 * PI * k / N); }
    size_t o_tw512 = d->arena.add(tw512.data(), tw512.size() * 4), o_tw1024 = d->arena.add(tw1024.data(), tw1024.size() * 4);
    size_t o_win = d->arena.add(d->window.data(), N * 4), o_env = d->arena.add(inv_env.data(), N * 4);
    size_t o_glw = 0;
    if (N == 1024) {
        // the same fp32 products the three-wave Griffin-Lim forms per lane (dn_gl_body.hpp: cw, wsyn), so the two schedules stay bit-identical
        const int H = N / 2;
        std::vector<float> t((size_t)4 * NC * 2);
        for (int m = 0; m < NC; ++m) {
            const float w0 = d->window[2 * m], w1 = d->window[2 * m + 1];
            const int n0 = 2 * m, n1 = n0 + 1;
            for (int c = 0; c < 3; ++c) {
                int i0, i1;
                if (c == 1) { i0 = n0; i1 = n1; }
                else if (c == 0) { i0 = n0 < H ? H - n0 : n0 - H; i1 = n1 < H ? H - n1 : n1 - H; }
                else { i0 = n0 < H ? n0 + H : 3 * H - 2 - n0; i1 = n1 < H ? n1 + H : 3 * H - 2 - n1; }
                t[((size_t)c * NC + m) * 2] = w0 * inv_env[i0];
                t[((size_t)c * NC + m) * 2 + 1] = w1 * inv_env[i1];
            }
            t[((size_t)3 * NC + m) * 2] = w0 * (1.0f / (float)NC);
            t[((size_t)3 * NC + m) * 2 + 1] = w1 * (1.0f / (float)NC);
        }
        o_glw = d->arena.add(t.data(), t.size() * 4);
    }
    size_t o_ms = 0, o_ml = 0, o_mw = 0, o_pinv = 0, o_ginv = 0, o_fb2 = 0;
    bool has_factors = false;
    int maxlen = 0, qsteps = 0;
    unsigned long long qlast = 0;
    size_t o_q = 0;
    const int pstride = ((K + 767) / 768) * 768;      // the contraction kernels stream rows in rounds of 192 or 256 bins (zero padded: tail loads stay in bounds)
    if (M > 0) {
        d->fb.resize((size_t)K * M);
        if (fb_in) memcpy(d->fb.data(), fb_in, d->fb.size() * 4);
        else {   // HTK triangles, f_min 0, f_max sr//2, norm None
            const double f_hi = (double)(cfg->sample_rate / 2);
            const double mmax = 2595.0 * log10(1.0 + f_hi / 700.0);
            std::vector<double> fp(M + 2);
            for (int i = 0; i < M + 2; ++i) fp[i] = 700.0 * (pow(10.0, (mmax * i / (M + 1)) / 2595.0) - 1.0);
            for (int k = 0; k < K; ++k) {
                const double f = f_hi * k / (K - 1);
                for (int mm = 0; mm < M; ++mm) {
                    const double up = (f - fp[mm]) / (fp[mm + 1] - fp[mm]), down = (fp[mm + 2] - f) / (fp[mm + 2] - fp[mm + 1]);
                    d->fb[(size_t)k * M + mm] = (float)fmax(0.0, fmin(up, down));
                }
            }
        }
        // band structure: [first nonzero bin, last nonzero bin] of each filter
        std::vector<int> start(M), len(M);
        for (int mm = 0; mm < M; ++mm) {
            int lo = K, hi = -1;
            for (int k = 0; k < K; ++k)
                if (d->fb[(size_t)k * M + mm] != 0.0f) { lo = k < lo ? k : lo; hi = k; }
            start[mm] = hi < 0 ? 0 : lo;
            len[mm] = hi < 0 ? 0 : hi - lo + 1;
            maxlen = len[mm] > maxlen ? len[mm] : maxlen;
        }
        std::vector<float> mw((size_t)(maxlen ? maxlen : 1) * M, 0.0f);
        for (int mm = 0; mm < M; ++mm)
            for (int i = 0; i < len[mm]; ++i) mw[(size_t)i * M + mm] = d->fb[(size_t)(start[mm] + i) * M + mm];
        // pseudo-inverse of fb^T: fb (fb^T fb)^-1, the minimum-norm least-squares operator (Appendix B.4)
        d->pinv.resize((size_t)K * M);
        std::vector<float> ginv, fb2;
        if (pinv_in) memcpy(d->pinv.data(), pinv_in, d->pinv.size() * 4);
        else {
            std::vector<double> G((size_t)M * M, 0.0), R((size_t)M * K);
            for (int a = 0; a < M; ++a)
                for (int b = 0; b <= a; ++b) {
                    double s = 0.0;
                    for (int k = 0; k < K; ++k) s += (double)d->fb[(size_t)k * M + a] * d->fb[(size_t)k * M + b];
                    G[(size_t)a * M + b] = G[(size_t)b * M + a] = s;
                }
            for (int a = 0; a < M; ++a)
                for (int k = 0; k < K; ++k) R[(size_t)a * K + k] = d->fb[(size_t)k * M + a];   // fb^T
            std::vector<double> G2 = G, I((size_t)M * M, 0.0);
            for (int a = 0; a < M; ++a) I[(size_t)a * M + a] = 1.0;
            if (!chol_solve(G, R, M, K)) { delete d; return fail(DN_ERR_INVALID, "mel filterbank is rank deficient; pass pinv explicitly"); }
            // the factored form needs at most two filters a bin
            bool two = chol_solve(G2, I, M, M);
            std::vector<float> f2((size_t)K * 4, 0.0f);
            for (int k = 0; k < K && two; ++k) {
                int n = 0;
                for (int a = 0; a < M; ++a) {
                    const float wv = d->fb[(size_t)k * M + a];
                    if (wv == 0.0f) continue;
                    if (n == 2) { two = false; break; }
                    f2[(size_t)k * 4 + n] = wv;
                    memcpy(&f2[(size_t)k * 4 + 2 + n], &a, 4);
                    ++n;
                }
            }
            if (two) {   // the diagonals beyond the band must be below fp32 resolution
                double big = 0.0, out = 0.0;
                for (int a = 0; a < M; ++a)
                    for (int c = 0; c < M; ++c) {
                        const double v = fabs(I[(size_t)a * M + c]);
                        big = v > big ? v : big;
                        if (abs(a - c) > dn::kInvBand) out = v > out ? v : out;
                    }
                two = out <= 1e-8 * big;
            }
            if (two) {
                const int taps = 2 * dn::kInvBand + 1;
                ginv.assign((size_t)taps * M, 0.0f);
                for (int t = 0; t < taps; ++t)
                    for (int a = 0; a < M; ++a) {
                        const int c = a - dn::kInvBand + t;
                        if (c >= 0 && c < M) ginv[(size_t)t * M + a] = (float)I[(size_t)a * M + c];
                    }
                fb2 = f2;
            }
            for (int k = 0; k < K; ++k)
                for (int a = 0; a < M; ++a) d->pinv[(size_t)k * M + a] = (float)R[(size_t)a * K + k];
        }
        std::vector<float> pt((size_t)((M + 15) / 16 * 16) * pstride, 0.0f);      // zero rows up to a multiple of 16: the dense loop walks 16 filters a round
        for (int k = 0; k < K; ++k)
            for (int a = 0; a < M; ++a) pt[(size_t)a * pstride + k] = d->pinv[(size_t)k * M + a];
        {   // packed schedule: groups of 16 filters, four lanes a filter, lane l of step u of a group takes tap 4 u + l % 4
            std::vector<float> q;
            for (int g = 0; g < (M + 15) / 16; ++g) {       // the last group may be partial: lanes of filters that do not exist carry zero weights
                int gl = 0;
                for (int f = 0; f < 16 && 16 * g + f < M; ++f) gl = len[16 * g + f] > gl ? len[16 * g + f] : gl;
                const int steps = gl ? (gl + 3) / 4 : 1;
                for (int u = 0; u < steps; ++u)
                    for (int l = 0; l < 64; ++l) {
                        const int mm = 16 * g + l / 4, i = 4 * u + l % 4;
                        const bool tap = mm < M && i < len[mm];
                        const int bin = tap ? start[mm] + i : 0;
                        float bits;
                        memcpy(&bits, &bin, 4);
                        q.push_back(tap ? mw[(size_t)i * M + mm] : 0.0f);
                        q.push_back(bits);
                    }
                qsteps += steps;
                if (qsteps <= dn::mel_q_steps(N)) qlast |= 1ull << (qsteps - 1);
            }
            if (qsteps > dn::mel_q_steps(N)) { qsteps = 0; qlast = 0; }
            else {
                q.resize((size_t)dn::mel_q_steps(N) * 64 * 2, 0.0f);         // padded to the full schedule with (weight 0, bin 0): the kernel loads every step unconditionally
                o_q = d->arena.add(q.data(), q.size() * 4);
            }
        }
        o_ms = d->arena.add(start.data(), M * 4);
        o_ml = d->arena.add(len.data(), M * 4);
        o_mw = d->arena.add(mw.data(), mw.size() * 4);
        o_pinv = d->arena.add(pt.data(), pt.size() * 4);
        if (!ginv.empty()) { o_ginv = d->arena.add(ginv.data(), ginv.size() * 4); o_fb2 = d->arena.add(fb2.data(), fb2.size() * 4); has_factors = true; }
    }
    hipError_t e = d->arena.upload();
    if (e != hipSuccess) { d->arena.release(); delete d; return fail(DN_ERR_HIP, std::string("plan upload: ") + hipGetErrorString(e)); }
    dn::DspDev& v = d->view;
    v.n_fft = N;
    v.twc = d->arena.ptr<float2>(o_tw512);
    v.twr = d->arena.ptr<float2>(o_tw1024);
    v.window = d->arena.ptr<float>(o_win);
    v.inv_env = d->arena.ptr<float>(o_env);
    v.glw_tables = N == 1024 ? d->arena.ptr<float2>(o_glw) : nullptr;
    v.n_mels = M;
    v.mel_maxlen = maxlen;
    v.mel_q = qsteps ? d->arena.ptr<float2>(o_q) : nullptr;
    v.mel_qsteps = qsteps;
    v.mel_qlast = qlast;
    v.pinv_stride = pstride;
    v.mel_start = M ? d->arena.ptr<int>(o_ms) : nullptr;
    v.mel_len = M ? d->arena.ptr<int>(o_ml) : nullptr;
    v.mel_w = M ? d->arena.ptr<float>(o_mw) : nullptr;
    v.pinv_t = M ? d->arena.ptr<float>(o_pinv) : nullptr;
    v.ginv_band = has_factors ? d->arena.ptr<float>(o_ginv) : nullptr;
    v.fb2 = has_factors ? d->arena.ptr<float4>(o_fb2) : nullptr;
    e = hipMalloc(reinterpret_cast<void**>(&d->view_dev), sizeof(dn::DspDev));
    if (e == hipSuccess) e = hipMemcpy(d->view_dev, &d->view, sizeof(dn::DspDev), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (d->view_dev) (void)hipFree(d->view_dev);
        d->arena.release(); delete d;
        return fail(DN_ERR_HIP, std::string("plan view: ") + hipGetErrorString(e));
    }
    *out = d;
    return DN_OK;
}

static void dsp_release(dn_dsp* d) {
    if (!d || d->refs.fetch_sub(1) != 1) return;
    d->arena.release();
    if (d->view_dev) (void)hipFree(d->view_dev);
    delete d;
}

void dn_dsp_destroy(dn_dsp* d) { dsp_release(d); }

int dn_dsp_get_tables(const dn_dsp* d, float* fb, float* pinv, float* window) {
    if (!d) return fail(DN_ERR_INVALID, "dn_dsp_get_tables: null plan");
    if (fb && !d->fb.empty()) memcpy(fb, d->fb.data(), d->fb.size() * 4);
    if (pinv && !d->pinv.empty()) memcpy(pinv, d->pinv.data(), d->pinv.size() * 4);
    if (window) memcpy(window, d->window.data(), d->window.size() * 4);
    return DN_OK;
}

int dn_stft(const dn_dsp* d, const float* frames, float* spec, int32_t B, uint32_t flags, void* stream) {
    if (B == 0) return DN_OK;      // empty batch: nothing to do (pointers of empty tensors are null)
    if (!d || !frames || !spec) return fail(DN_ERR_INVALID, "dn_stft: null argument");
    if (B < 0) return fail(DN_ERR_INVALID, "dn_stft: negative batch");
    if (B == 0) return DN_OK;
    dn::launch_stft(d->view, frames, spec, nullptr, nullptr, B, flags, as_stream(stream));
    return check_launch("stft_kernel");
}

int dn_stft_mel_log1p(const dn_dsp* d, const float* frames, float* mel, float* peak, int32_t B, uint32_t flags, void* stream) {
    if (B == 0) return DN_OK;      // empty batch: nothing to do (pointers of empty tensors are null)
    if (!d || !frames || !mel) return fail(DN_ERR_INVALID, "dn_stft_mel_log1p: null argument");
    if (d->cfg.n_mels <= 0) return fail(DN_ERR_INVALID, "plan was created without mel stages");
    if (B < 0) return fail(DN_ERR_INVALID, "dn_stft_mel_log1p: negative batch");
    if (B == 0) return DN_OK;
    dn::launch_stft(d->view, frames, nullptr, mel, peak, B, flags, as_stream(stream));
    return check_launch("stft_kernel");
}

int dn_mel_scale(const dn_dsp* d, const float* mag, float* mel, int32_t B, int32_t T, void* stream) {
    if (!d || !mag || !mel) return fail(DN_ERR_INVALID, "dn_mel_scale: null argument");
    if (d->cfg.n_mels <= 0) return fail(DN_ERR_INVALID, "plan was created without mel stages");
    if (B < 0 || T < 0) return fail(DN_ERR_INVALID, "dn_mel_scale: negative size");
    if ((int64_t)B * T == 0) return DN_OK;
    dn::launch_mel(d->view, mag, mel, B * T, as_stream(stream));
    return check_launch("mel_kernel");
}

int dn_invmel(const dn_dsp* d, const float* mel_mag, float* lin, int32_t B, int32_t T, void* stream) {
    if (!d || !mel_mag || !lin) return fail(DN_ERR_INVALID, "dn_invmel: null argument");
    if (d->cfg.n_mels <= 0) return fail(DN_ERR_INVALID, "plan was created without mel stages");
    if (B < 0 || T < 0) return fail(DN_ERR_INVALID, "dn_invmel: negative size");
    if ((int64_t)B * T == 0) return DN_OK;
    dn::launch_invmel(d->view, mel_mag, nullptr, lin, B * T, as_stream(stream));
    return check_launch("invmel_kernel");
}

int dn_residual_invmel(const dn_dsp* d, const float* x, const float* diff, float* lin, int32_t B, int32_t T, void* stream) {
    if (!d || !x || !diff || !lin) return fail(DN_ERR_INVALID, "dn_residual_invmel: null argument");
    if (d->cfg.n_mels <= 0) return fail(DN_ERR_INVALID, "plan was created without mel stages");
    if (B < 0 || T < 0) return fail(DN_ERR_INVALID, "dn_residual_invmel: negative size");
    if ((int64_t)B * T == 0) return DN_OK;
    dn::launch_invmel(d->view, x, diff, lin, B * T, as_stream(stream));
    return check_launch("invmel_kernel");
}

int dn_griffinlim(const dn_dsp* d, const float* mag, const float* init_angles, uint64_t seed, uint64_t stream_id0,
                  const float* scale, float* wave, int32_t B, int32_t n_iter, float momentum, void* stream) {
    if (B == 0) return DN_OK;      // empty batch: nothing to do (pointers of empty tensors are null)
    if (!d || !mag || !wave) return fail(DN_ERR_INVALID, "dn_griffinlim: null argument");
    if (B < 0 || n_iter < 0) return fail(DN_ERR_INVALID, "dn_griffinlim: negative size");
    if (!(momentum >= 0.0f && momentum < 1.0f)) return fail(DN_ERR_INVALID, "momentum must be in [0, 1)");
    if (B == 0) return DN_OK;
    dn::launch_griffinlim(d->view, mag, init_angles, seed, stream_id0, scale, wave, B, n_iter, momentum, as_stream(stream));
    return check_launch("griffinlim_kernel");
}

int dn_griffinlim_draw_phases(const dn_dsp* d, uint64_t seed, uint64_t stream_id0, float* angles_out, int32_t B, void* stream) {
    if (B == 0) return DN_OK;
    if (!d || !angles_out) return fail(DN_ERR_INVALID, "dn_griffinlim_draw_phases: null argument");
    if (B < 0) return fail(DN_ERR_INVALID, "dn_griffinlim_draw_phases: negative batch");
    dn::launch_draw_phases(d->view, angles_out, seed, stream_id0, B, as_stream(stream));
    return check_launch("draw_phases_kernel");
}

int dn_synthesis(const dn_dsp* d, const float* x, const float* diff, const float* init_angles, uint64_t seed, uint64_t stream_id0,
                 const float* scale, float* wave, int32_t B, int32_t n_iter, float momentum, void* stream) {
    if (B == 0) return DN_OK;      // empty batch: nothing to do (pointers of empty tensors are null)
    if (!d || !x || !diff || !wave) return fail(DN_ERR_INVALID, "dn_synthesis: null argument");
    if (d->cfg.n_mels <= 0) return fail(DN_ERR_INVALID, "plan was created without mel stages");
    if (B < 0 || n_iter < 0) return fail(DN_ERR_INVALID, "dn_synthesis: negative size");
    if (!(momentum >= 0.0f && momentum < 1.0f)) return fail(DN_ERR_INVALID, "momentum must be in [0, 1)");
    if (B == 0) return DN_OK;
    dn::launch_synthesis(d->view, x, diff, init_angles, seed, stream_id0, scale, wave, B, n_iter, momentum, as_stream(stream));
    return check_launch("griffinlim_kernel<from mel>");
}

int dn_istft(const dn_dsp* d, const float* spec, float* wave, int32_t B, void* stream) {
    if (B == 0) return DN_OK;      // empty batch: nothing to do (pointers of empty tensors are null)
    if (!d || !spec || !wave) return fail(DN_ERR_INVALID, "dn_istft: null argument");
    if (B < 0) return fail(DN_ERR_INVALID, "dn_istft: negative batch");
    if (B == 0) return DN_OK;
    dn::launch_griffinlim(d->view, nullptr, spec, 0, 0, nullptr, wave, B, 0, 0.0f, as_stream(stream));
    return check_launch("griffinlim_kernel(istft)");
}

static size_t frame_scratch_bytes(const dn_dsp* d, int32_t B) {
    const size_t M = (size_t)d->cfg.n_mels, K = (size_t)d->cfg.n_fft / 2 + 1;
    return ((size_t)B * (6 * M + 3 * K + 1) * sizeof(float) + 255) & ~size_t(255);
}

int dn_stft_general(const dn_dsp* d, const float* x, float* spec, float* logmel, int32_t B, int32_t L, void* stream) {
    if (B == 0) return DN_OK;
    if (!d || !x || (!spec && !logmel)) return fail(DN_ERR_INVALID, "dn_stft_general: null argument");
    if (B < 0) return fail(DN_ERR_INVALID, "dn_stft_general: negative batch");
    if (L <= d->cfg.n_fft / 2) return fail(DN_ERR_INVALID, "dn_stft_general: reflect padding needs more than n_fft/2 samples (as torch.stft)");
    if (logmel && d->cfg.n_mels <= 0) return fail(DN_ERR_INVALID, "plan was created without mel stages");
    dn::launch_stft_general(d->view, x, spec, logmel, B, L, as_stream(stream));
    return check_launch("stft_general_kernel");
}

int dn_server_rows(const dn_dsp* d, const float* logmel, const float* model_out, const float* spec_in, float* spec_out, int32_t rows,
                   void* stream) {
    if (rows == 0) return DN_OK;
    if (!d || !logmel || !model_out || !spec_in || !spec_out) return fail(DN_ERR_INVALID, "dn_server_rows: null argument");
    if (rows < 0) return fail(DN_ERR_INVALID, "dn_server_rows: negative size");
    if (d->cfg.n_mels <= 0) return fail(DN_ERR_INVALID, "plan was created without mel stages");
    dn::launch_server_rows(d->view, logmel, model_out, spec_in, spec_out, rows, as_stream(stream));
    return check_launch("server_rows_kernel");
}

int dn_istft_general(const dn_dsp* d, const float* spec, float* wave, int32_t B, int32_t T, void* stream) {
    if (B == 0) return DN_OK;
    if (!d || !spec || !wave) return fail(DN_ERR_INVALID, "dn_istft_general: null argument");
    if (B < 0 || T < 2) return fail(DN_ERR_INVALID, "dn_istft_general: need B >= 0 and at least 2 columns");
    dn::launch_istft_general(d->view, spec, wave, B, T, as_stream(stream));
    return check_launch("istft_general_kernel");
}

size_t dn_workspace_bytes(const dn_dsp* d, int32_t B) {
    if (!d || B <= 0) return 0;
    // per-hop scratch (mel in, mel residual, linear magnitude, peak) + one denoised frame for dn_stream_step
    return frame_scratch_bytes(d, B) + (((size_t)B * d->cfg.n_fft * sizeof(float) + 255) & ~size_t(255));
}

static int check_hop_args(const char* who, const dn_model* m, const dn_dsp* d, int32_t B, int32_t n_iter, float momentum, uint32_t flags) {
    if (!m || !d) return fail(DN_ERR_INVALID, std::string(who) + ": null handle");
    if (d->cfg.n_mels <= 0 || d->cfg.n_mels % 16) return fail(DN_ERR_INVALID, "n_mels must be a positive multiple of 16");
    if (B < 0 || n_iter < 0) return fail(DN_ERR_INVALID, std::string(who) + ": negative size");
    if (!(momentum >= 0.0f && momentum < 1.0f)) return fail(DN_ERR_INVALID, "momentum must be in [0, 1)");
    if (d->cfg.n_mels / 16 > dn::kMaxC) return fail(DN_ERR_UNSUPPORTED, "n_mels/16 exceeds the cell kernel's limit");
    if (flags & ~(uint32_t)DN_CONV_BF16) return fail(DN_ERR_INVALID, std::string(who) + ": unknown flag bits");
    return DN_OK;
}

int dn_process_frame(const dn_model* m, const dn_dsp* d, const float* frames, float* hx, float* out, float* mel_residual_out,
                     const float* init_angles, uint64_t seed, uint64_t stream_id0, int32_t n_iter, float momentum,
                     void* workspace, int32_t B, uint32_t flags, void* stream) {
    if (B == 0) return DN_OK;      // empty batch: nothing to do (pointers of empty tensors are null)
    if (!frames || !hx || !out || !workspace) return fail(DN_ERR_INVALID, "dn_process_frame: null argument");
    int rc = check_hop_args("dn_process_frame", m, d, B, n_iter, momentum, flags);
    if (rc != DN_OK) return rc;
    const int M = d->cfg.n_mels, K = d->cfg.n_fft / 2 + 1, C = M / 16;
    float* ws = static_cast<float*>(workspace);
    BiasSet* bs = nullptr;
    rc = build_bias(const_cast<dn_model*>(m), C, &bs);
    if (rc != DN_OK) return rc;
    dn::FrameArgs a{};
    a.frames = frames; a.hx = hx; a.out = out;
    a.mel = ws;
    a.diff = mel_residual_out ? mel_residual_out : ws + (size_t)B * 3 * M;
    a.peak = ws + (size_t)B * 6 * M + (size_t)B * 3 * K;
    a.init = init_angles; a.seed = seed; a.sid0 = stream_id0;
    a.n_iter = n_iter; a.mom = momentum / (1.0f + momentum); a.C = C;
    dn::launch_frame(d->view, bs->view, a, B, (flags & DN_CONV_BF16) != 0, as_stream(stream));      // P1-P12, one launch
    return check_launch("frame_kernel");
}

int dn_stream_step(const dn_model* m, const dn_dsp* d, const float* hop_in, float* ring, float* ola, float* hx, float* hop_out,
                   const float* init_angles, uint64_t seed, uint64_t stream_id0, int32_t n_iter, float momentum, void* workspace,
                   int32_t B, uint32_t flags, void* stream) {
    if (B == 0) return DN_OK;      // empty batch: nothing to do (pointers of empty tensors are null)
    // every argument is validated before the first (and only) launch, so a failed call leaves ring, ola and hx untouched
    if (!hop_in || !ring || !ola || !hx || !hop_out || !workspace) return fail(DN_ERR_INVALID, "dn_stream_step: null argument");
    int rc = check_hop_args("dn_stream_step", m, d, B, n_iter, momentum, flags);
    if (rc != DN_OK) return rc;
    const int M = d->cfg.n_mels, K = d->cfg.n_fft / 2 + 1, C = M / 16;
    float* ws = static_cast<float*>(workspace);
    BiasSet* bs = nullptr;
    rc = build_bias(const_cast<dn_model*>(m), C, &bs);
    if (rc != DN_OK) return rc;
    dn::FrameArgs a{};
    a.hx = hx;
    a.mel = ws; a.diff = ws + (size_t)B * 3 * M; a.peak = ws + (size_t)B * 6 * M + (size_t)B * 3 * K;
    a.init = init_angles; a.seed = seed; a.sid0 = stream_id0;
    a.n_iter = n_iter; a.mom = momentum / (1.0f + momentum); a.C = C;
    a.hop_in = hop_in; a.ring = ring; a.ola = ola; a.hop_out = hop_out;
    dn::launch_frame(d->view, bs->view, a, B, (flags & DN_CONV_BF16) != 0, as_stream(stream));
    return check_launch("frame_kernel(stream)");
}

// ------------------------------------------------------------------ software-pipelined hops
static size_t slot_floats(const dn_dsp* d, int32_t B) {
    return (size_t)B * (6 * (size_t)d->cfg.n_mels + 1 + dn::kSlotMeta + 3 * ((size_t)d->cfg.n_fft / 2 + 1));
}

// the head start a one-hop pipe runs with unless told otherwise (measured optima; none above 256 streams, none on a deep pipe)
static int default_head_start(const dn_pipe* p) {
    return p->depth == 1 && p->B <= 256 ? (p->d->cfg.n_fft == 1536 ? 12 : 8) : 0;
}

int dn_pipe_create(const dn_model* m, const dn_dsp* d, int32_t B, uint32_t flags, dn_pipe** out) {
    if (!out) return fail(DN_ERR_INVALID, "dn_pipe_create: null argument");
    if (B <= 0) return fail(DN_ERR_INVALID, "dn_pipe_create: batch must be positive");
    int rc = check_hop_args("dn_pipe_create", m, d, B, 0, 0.0f, flags);
    if (rc != DN_OK) return rc;
    dn_pipe* p = new dn_pipe();
    p->m = const_cast<dn_model*>(m); p->d = const_cast<dn_dsp*>(d); p->B = B; p->C = d->cfg.n_mels / 16;
    p->bf16 = (flags & DN_CONV_BF16) != 0;
    p->m->refs.fetch_add(1);       // the pipe keeps the packed weights and the plan alive (its launches read their device arenas)
    p->d->refs.fetch_add(1);
    rc = build_bias(p->m, p->C, &p->bs);
    if (rc != DN_OK) { dn_pipe_destroy(p); return rc; }
    p->slot_floats = (slot_floats(d, B) + 63) & ~size_t(63);
    p->init_elems = (size_t)B * 3 * (d->cfg.n_fft / 2 + 1);
    p->state_elems = (size_t)B * 3 * (2 * ((size_t)d->cfg.n_fft / 128) + 2) * 64;       // n_fft / 128 complex values per lane
    static_assert(sizeof(dn::PipeCtl) <= 256, "the control block fits its allocation");
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p->ctl), 256);
    if (e == hipSuccess) e = hipMemset(p->ctl, 0, 256);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->scratch), p->n_slots * p->slot_floats * sizeof(float));
    if (e != hipSuccess) { dn_pipe_destroy(p); return fail(DN_ERR_HIP, std::string("dn_pipe_create: ") + hipGetErrorString(e)); }
    {
        // Griffin-Lim head start: with at most one stream per CU (MI355X: 256 CUs) a front workgroup has slack at the end of a launch that
        // the pending hop's chain does not -- measured best at 8 iterations for n_fft 1024 (65.0 -> 56.0 us per batch-256 hop) and 12 for
        // n_fft 1536 (131 -> 97 us; round 3, profiles/r03_head_start_sweep.txt); with more streams than CUs every workgroup is busy throughout and it only adds traffic.
        // (The optimum moves up whenever the front half gets shorter: re-run tools/head_start_sweep.sh after changing either half.)
        const int it = default_head_start(p);       // (experiments: dn_pipe_set_head_start, tools/head_start_sweep.sh)
        if (it > 0) { rc = dn_pipe_set_head_start(p, it); if (rc != DN_OK) { dn_pipe_destroy(p); return rc; } }
    }
    *out = p;
    return DN_OK;
}

void dn_pipe_destroy(dn_pipe* p) {
    if (!p) return;
    if (p->hio) {
        HostIo* h = p->hio;
        if (h->h2d) (void)hipStreamSynchronize(h->h2d);
        if (h->d2h) (void)hipStreamSynchronize(h->d2h);
        for (int i = 0; i < HostIo::kRing; ++i) {
            if (h->ev_h2d[i]) (void)hipEventDestroy(h->ev_h2d[i]);
            if (h->ev_k[i]) (void)hipEventDestroy(h->ev_k[i]);
            if (h->ev_d2h[i]) (void)hipEventDestroy(h->ev_d2h[i]);
        }
        for (int i = 0; i < 2; ++i) {
            if (h->d_in[i]) (void)hipFree(h->d_in[i]);
            if (h->d_out[i]) (void)hipFree(h->d_out[i]);
            if (h->z_out[i]) (void)hipFree(h->z_out[i]);
        }
        if (h->h2d) (void)hipStreamDestroy(h->h2d);
        if (h->d2h) (void)hipStreamDestroy(h->d2h);
        if (h->done_host) (void)hipHostFree(h->done_host);
        delete h;
    }
    if (p->scratch) (void)hipFree(p->scratch);
    if (p->scratch_init) (void)hipFree(p->scratch_init);
    if (p->gl_state) (void)hipFree(p->gl_state);
    if (p->ctl) (void)hipFree(p->ctl);
    if (p->ring) (void)hipFree(p->ring);
    if (p->ola) (void)hipFree(p->ola);
    if (p->hx) (void)hipFree(p->hx);
    model_release(p->m);
    dsp_release(p->d);
    delete p;
}

int dn_pipe_set_head_start(dn_pipe* p, int32_t iterations) {
    if (!p || iterations < 0) return fail(DN_ERR_INVALID, "dn_pipe_set_head_start: bad argument");
    if (iterations > 0 && p->group > 0) return fail(DN_ERR_INVALID, "dn_pipe_set_head_start: a group pipe never parks a chain");
    if (iterations > 0 && p->split == DN_SPLIT_ON) return fail(DN_ERR_INVALID, "dn_pipe_set_head_start: the pipe splits every hop into two launches (dn_pipe_set_split(p, DN_SPLIT_OFF or DN_SPLIT_AUTO) first)");
    if (iterations > 0 && !p->gl_state) DN_HIP(hipMalloc(reinterpret_cast<void**>(&p->gl_state), p->n_slots * p->state_elems * sizeof(float2)));
    if (iterations > 0 && p->d->cfg.n_fft == 1536) {      // n_fft 1536: the front workgroup's spare wave draws the head start's initial phases into the slot
        int rc = dn_pipe_reserve_parity(p);
        if (rc != DN_OK) return rc;
    }
    p->gl_split = iterations;
    return DN_OK;
}

int dn_pipe_set_depth(dn_pipe* p, int32_t depth) {
    if (!p) return fail(DN_ERR_INVALID, "dn_pipe_set_depth: null pipe");
    if (depth < 1 || depth > DN_PIPE_MAX_DEPTH) return fail(DN_ERR_INVALID, "dn_pipe_set_depth: depth must be in 1.." + std::to_string(DN_PIPE_MAX_DEPTH));
    if (depth > 1 && p->d->cfg.n_fft != 1024)
        return fail(DN_ERR_UNSUPPORTED, "pipes deeper than one hop are built for n_fft 1024 (at 1536 the per-lane state of a stream does not fit a wavefront's registers)");
    if (depth > 1 && p->group > 0) return fail(DN_ERR_INVALID, "dn_pipe_set_depth: a group pipe runs whole chains per launch (dn_pipe_set_group(p, 0) first)");
    if (depth == p->depth) return DN_OK;
    // nothing may be in flight: the slots are re-laid out
    dn::PipeCtl h{};
    DN_HIP(hipDeviceSynchronize());
    DN_HIP(hipMemcpy(&h, p->ctl, sizeof(h), hipMemcpyDeviceToHost));
    if (h.pending != 0) return fail(DN_ERR_INVALID, "dn_pipe_set_depth: hops are in flight (flush first)");
    const int n_slots = depth + 1;
    // everything the new depth needs is allocated before anything of the old one is given up: a failure leaves the pipe as it was
    float* scratch = nullptr;
    float2* state = nullptr;
    float2* init = nullptr;         // a deep pipe's front workgroups leave every frame's initial phases in its slot (dn_hop.hip)
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&scratch), n_slots * p->slot_floats * sizeof(float));
    if (e == hipSuccess && depth > 1) e = hipMalloc(reinterpret_cast<void**>(&state), n_slots * p->state_elems * sizeof(float2));
    if (e == hipSuccess && depth > 1) e = hipMalloc(reinterpret_cast<void**>(&init), n_slots * p->init_elems * sizeof(float2));
    if (e != hipSuccess) {
        if (scratch) (void)hipFree(scratch);
        if (state) (void)hipFree(state);
        if (init) (void)hipFree(init);
        return fail(DN_ERR_HIP, std::string("dn_pipe_set_depth: ") + hipGetErrorString(e));
    }
    (void)hipFree(p->scratch);
    if (p->gl_state) (void)hipFree(p->gl_state);
    if (p->scratch_init) (void)hipFree(p->scratch_init);
    p->scratch = scratch; p->gl_state = state; p->scratch_init = init;
    p->depth = depth; p->n_slots = n_slots;
    p->gl_split = 0;                            // (a head start is set per depth: dn_pipe_set_head_start)
    h.slot_next = 0;
    DN_HIP(hipMemcpy(p->ctl, &h, sizeof(h), hipMemcpyHostToDevice));
    return dn_pipe_set_head_start(p, default_head_start(p));      // back to depth 1: the one-hop pipe's default head start again
}

int dn_pipe_set_group(dn_pipe* p, int32_t hops) {
    if (!p) return fail(DN_ERR_INVALID, "dn_pipe_set_group: null pipe");
    if (hops < 0 || hops > DN_PIPE_MAX_GROUP) return fail(DN_ERR_INVALID, "dn_pipe_set_group: hops must be in 0.." + std::to_string(DN_PIPE_MAX_GROUP));
    if (hops > 0 && p->d->cfg.n_fft != 1024)
        return fail(DN_ERR_UNSUPPORTED, "hop groups run whole Griffin-Lim chains one wavefront per stream, which is built for n_fft 1024 (at 1536 that form measured slower than a wavefront per column)");
    if (hops > 0 && p->depth > 1) return fail(DN_ERR_INVALID, "dn_pipe_set_group: the pipe is deeper than one hop (dn_pipe_set_depth(p, 1) first)");
    if (hops > 0 && p->hio) return fail(DN_ERR_UNSUPPORTED, "dn_pipe_set_group: the host-buffer transport moves single hops");
    if (hops == p->group) return DN_OK;
    // nothing may be in flight: the slots are re-laid out (two groups of the largest size: one being fronted, one whose chains run)
    dn::PipeCtl h{};
    DN_HIP(hipDeviceSynchronize());
    DN_HIP(hipMemcpy(&h, p->ctl, sizeof(h), hipMemcpyDeviceToHost));
    if (h.pending != 0) return fail(DN_ERR_INVALID, "dn_pipe_set_group: hops are in flight (flush first)");
    const int n_slots = hops > 0 ? 2 * DN_PIPE_MAX_GROUP : p->depth + 1;
    float* scratch = nullptr;
    float2* init = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&scratch), n_slots * p->slot_floats * sizeof(float));
    if (e == hipSuccess && p->scratch_init) e = hipMalloc(reinterpret_cast<void**>(&init), n_slots * p->init_elems * sizeof(float2));
    if (e != hipSuccess) {
        if (scratch) (void)hipFree(scratch);
        if (init) (void)hipFree(init);
        return fail(DN_ERR_HIP, std::string("dn_pipe_set_group: ") + hipGetErrorString(e));
    }
    (void)hipFree(p->scratch);
    if (p->scratch_init) (void)hipFree(p->scratch_init);
    if (p->gl_state) { (void)hipFree(p->gl_state); p->gl_state = nullptr; }
    p->scratch = scratch; p->scratch_init = init;
    p->n_slots = n_slots;
    p->group = hops; p->group_pending = 0;
    p->gl_split = 0;
    h.slot_next = 0;
    DN_HIP(hipMemcpy(p->ctl, &h, sizeof(h), hipMemcpyHostToDevice));
    return hops > 0 ? DN_OK : dn_pipe_set_head_start(p, default_head_start(p));
}

int dn_pipe_set_gl_schedule(dn_pipe* p, int32_t schedule) {
    if (!p) return fail(DN_ERR_INVALID, "dn_pipe_set_gl_schedule: null pipe");
    if (schedule != DN_GL_AUTO && schedule != DN_GL_WAVE_PER_COLUMN && schedule != DN_GL_WAVE_PER_STREAM)
        return fail(DN_ERR_INVALID, "dn_pipe_set_gl_schedule: unknown schedule");
    if (schedule == DN_GL_WAVE_PER_STREAM && p->d->cfg.n_fft != 1024)
        return fail(DN_ERR_UNSUPPORTED, "the wavefront-per-stream Griffin-Lim is built for n_fft 1024 (at 1536 the per-lane state of a stream does not fit a wavefront's registers)");
    if (schedule == DN_GL_WAVE_PER_COLUMN && p->depth > 1)
        return fail(DN_ERR_INVALID, "a pipe deeper than one hop runs a wavefront per stream and chain segment");
    p->gl_schedule = schedule;
    return DN_OK;
}

int dn_pipe_set_split(dn_pipe* p, int32_t mode) {
    if (!p) return fail(DN_ERR_INVALID, "dn_pipe_set_split: null pipe");
    if (mode != DN_SPLIT_AUTO && mode != DN_SPLIT_OFF && mode != DN_SPLIT_ON) return fail(DN_ERR_INVALID, "dn_pipe_set_split: unknown mode");
    if (mode == DN_SPLIT_ON && p->d->cfg.n_fft != 1024)
        return fail(DN_ERR_UNSUPPORTED, "the split hop belongs to the wavefront-per-stream Griffin-Lim, which is built for n_fft 1024");
    if (mode == DN_SPLIT_ON && p->gl_split > 0)
        return fail(DN_ERR_INVALID, "dn_pipe_set_split: the pipe runs a head start (a front workgroup that goes on with its frame's chain cannot be a launch of its own): "
                                    "dn_pipe_set_head_start(p, 0) first");
    if (mode == DN_SPLIT_ON && p->group > 0) return fail(DN_ERR_INVALID, "dn_pipe_set_split: a group pipe is one launch per group");
    p->split = mode;
    return DN_OK;
}

int dn_pipe_set_model(dn_pipe* p, const dn_model* m) {
    if (!p || !m) return fail(DN_ERR_INVALID, "dn_pipe_set_model: null argument");
    if (m == p->m) return DN_OK;
    BiasSet* bs = nullptr;
    int rc = build_bias(const_cast<dn_model*>(m), p->C, &bs);
    if (rc != DN_OK) return rc;
    const_cast<dn_model*>(m)->refs.fetch_add(1);
    model_release(p->m);           // launches already enqueued read the old arenas: the caller keeps the old model alive until they ran
    p->m = const_cast<dn_model*>(m);
    p->bs = bs;
    return DN_OK;
}

int dn_pipe_stream_create(const dn_model* m, const dn_dsp* d, int32_t B, uint32_t flags, dn_pipe** out) {
    dn_pipe* p = nullptr;
    int rc = dn_pipe_create(m, d, B, flags, &p);
    if (rc != DN_OK) return rc;
    const size_t line = (size_t)B * d->cfg.n_fft * sizeof(float), hxb = (size_t)B * dn::kHidden * p->C * sizeof(float);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p->ring), line);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->ola), line);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->hx), hxb);
    if (e == hipSuccess) e = hipMemset(p->ring, 0, line);
    if (e == hipSuccess) e = hipMemset(p->ola, 0, line);
    if (e == hipSuccess) e = hipMemset(p->hx, 0, hxb);
    if (e != hipSuccess) { dn_pipe_destroy(p); return fail(DN_ERR_HIP, std::string("dn_pipe_stream_create: ") + hipGetErrorString(e)); }
    *out = p;
    return DN_OK;
}

static int stream_state_copy(dn_pipe* p, float* ring, float* ola, float* hx, bool to_pipe, void* stream) {
    if (!p || !p->ring) return fail(DN_ERR_INVALID, "not a streaming pipe");
    const size_t line = (size_t)p->B * p->d->cfg.n_fft * sizeof(float), hxb = (size_t)p->B * dn::kHidden * p->C * sizeof(float);
    hipStream_t st = as_stream(stream);
    if (ring) DN_HIP(hipMemcpyAsync(to_pipe ? p->ring : ring, to_pipe ? ring : p->ring, line, hipMemcpyDeviceToDevice, st));
    if (ola) DN_HIP(hipMemcpyAsync(to_pipe ? p->ola : ola, to_pipe ? ola : p->ola, line, hipMemcpyDeviceToDevice, st));
    if (hx) DN_HIP(hipMemcpyAsync(to_pipe ? p->hx : hx, to_pipe ? hx : p->hx, hxb, hipMemcpyDeviceToDevice, st));
    return DN_OK;
}

int dn_pipe_stream_get_state(dn_pipe* p, float* ring, float* ola, float* hx, void* stream) {
    return stream_state_copy(p, ring, ola, hx, false, stream);
}

int dn_pipe_stream_set_state(dn_pipe* p, const float* ring, const float* ola, const float* hx, uint64_t frames_done, void* stream) {
    int rc = stream_state_copy(p, const_cast<float*>(ring), const_cast<float*>(ola), const_cast<float*>(hx), true, stream);
    if (rc != DN_OK) return rc;
    // a restored ring is a primed ring; a hop that was pending is dropped (flush before taking a snapshot)
    p->group_pushes = (unsigned long long)(p->d->cfg.n_fft / p->d->cfg.hop - 1);          // (the host's own bookkeeping of a group pipe: primed, nothing pending)
    p->group_pending = 0;
    dn::launch_ctl_set(p->ctl, (unsigned long long)(p->d->cfg.n_fft / p->d->cfg.hop - 1), frames_done, 0, as_stream(stream));
    return check_launch("ctl_set_kernel");
}

int dn_pipe_get_counters(dn_pipe* p, uint64_t* pushes, uint64_t* frames, int32_t* pending, void* stream) {
    if (!p) return fail(DN_ERR_INVALID, "dn_pipe_get_counters: null pipe");
    dn::PipeCtl h{};
    DN_HIP(hipMemcpyAsync(&h, p->ctl, sizeof(h), hipMemcpyDeviceToHost, as_stream(stream)));
    DN_HIP(hipStreamSynchronize(as_stream(stream)));
    if (pushes) *pushes = h.pushes;
    if (frames) *frames = h.frames;
    if (pending) *pending = (int32_t)h.pending;
    return DN_OK;
}

// parity mode: a frame's injected phases travel with its scratch slot (so the caller's buffer is free again after the push)
int dn_pipe_reserve_parity(dn_pipe* p) {
    if (!p) return fail(DN_ERR_INVALID, "dn_pipe_reserve_parity: null pipe");
    if (!p->scratch_init) DN_HIP(hipMalloc(reinterpret_cast<void**>(&p->scratch_init), p->n_slots * p->init_elems * sizeof(float2)));
    return DN_OK;
}

// the parts of a launch that are the same for every kind of push/submit/flush
static int fill_hop_args(dn_pipe* p, dn::HopArgs& a, const float* init_angles, uint64_t seed, uint64_t stream_id0, int32_t n_iter,
                         float momentum) {
    if (n_iter < 0) return fail(DN_ERR_INVALID, "negative n_iter");
    if (!(momentum >= 0.0f && momentum < 1.0f)) return fail(DN_ERR_INVALID, "momentum must be in [0, 1)");
    if (init_angles && !p->scratch_init) {
        int rc = dn_pipe_reserve_parity(p);
        if (rc != DN_OK) return rc;
    }
    a.ctl = p->ctl;
    a.slots = p->scratch; a.slot_stride = p->slot_floats;
    a.slot_init = p->scratch_init; a.init_stride = p->init_elems;
    a.gl_state = p->gl_state; a.state_stride = p->state_elems;
    a.n_slots = p->n_slots;
    a.gl_split = p->gl_split;
    a.init_in = init_angles; a.seed = seed; a.sid0 = stream_id0;
    a.n_iter = n_iter; a.mom = momentum / (1.0f + momentum);
    a.B = p->B; a.C = p->C; a.back_B = p->B;
    // a wavefront per stream (four streams a workgroup) from four streams per CU up; a wavefront per column (the shortest chain) below
    const bool per_stream = p->depth > 1 || (p->d->cfg.n_fft == 1024 && (p->gl_schedule == DN_GL_WAVE_PER_STREAM ||
                                                                        (p->gl_schedule == DN_GL_AUTO && p->B >= dn::kGlwAutoStreams)));
    a.glw = per_stream ? 1 : 0;
    a.depth = p->depth;
    a.spb = per_stream ? 4 / p->depth : 1;          // a workgroup is four wavefronts: streams x chain segments
    a.back_blocks = (p->B + a.spb - 1) / a.spb;
    // two launches per hop where the chip holds only a fraction of a launch's workgroups at once (its chains and its front halves then run as
    // two phases anyway): from dn::kSplitAutoChains chain wavefronts on -- 2,048 fill every SIMD of an MI355X twice
    const bool can_split = per_stream && p->gl_split == 0;
    a.split = can_split && (p->split == DN_SPLIT_ON || (p->split == DN_SPLIT_AUTO && (long)p->B * p->depth >= dn::kSplitAutoChains)) ? 1 : 0;
    a.prime = p->d->cfg.n_fft / p->d->cfg.hop - 1;
    a.n_mels = p->d->cfg.n_mels; a.d_dev = p->d->view_dev; a.c_dev = p->bs->view_dev;          // (the views in device memory: what the front halves read)
    if (p->group > 0) {          // whole chains (group_kernel): a workgroup's four wavefronts = spb streams x the pending frames of each
        a.glw = 1; a.depth = 1; a.split = 0; a.gl_split = 0;
        a.spb = p->group == 1 ? 4 : p->group == 2 ? 2 : 1;
        a.back_blocks = (p->B + a.spb - 1) / a.spb;
    }
    return DN_OK;
}

int dn_pipe_stream_push(dn_pipe* p, const void* hop_in, int32_t in_is_s16, void* hop_out, int32_t out_is_s16,
                        const float* init_angles, uint64_t seed, uint64_t stream_id0, int32_t n_iter, float momentum, void* stream) {
    if (!p || !p->ring) return fail(DN_ERR_INVALID, "dn_pipe_stream_push: not a streaming pipe");
    if (!hop_in || !hop_out) return fail(DN_ERR_INVALID, "dn_pipe_stream_push: null argument");
    if (p->group > 0) return fail(DN_ERR_INVALID, "dn_pipe_stream_push: this pipe takes groups of hops (dn_pipe_stream_push_group)");
    dn::HopArgs a{};
    int rc = fill_hop_args(p, a, init_angles, seed, stream_id0, n_iter, momentum);
    if (rc != DN_OK) return rc;
    a.front_B = p->B; a.hx = p->hx;
    a.host_done = p->host_done; a.host_done_value = p->host_done_value;
    a.host_copy_src = static_cast<const uint4*>(p->host_copy_src); a.host_copy_dst = static_cast<uint4*>(p->host_copy_dst);
    a.host_copy_n16 = (unsigned int)(p->host_copy_bytes / 16);
    a.hop_in = hop_in; a.ring = p->ring; a.in_s16 = in_is_s16;
    a.ola = p->ola; a.hop_out = hop_out; a.out_s16 = out_is_s16;
    dn::launch_hop(p->d->view, p->bs->view, a, p->bf16, as_stream(stream));
    return check_launch("hop_kernel(stream)");
}

// ---- host-buffer transport (app3.py:168-172,189,215,244-250: the reference crosses host <-> device every hop)
static int host_io_init(dn_pipe* p) {
    if (p->hio) return DN_OK;
    HostIo* h = new HostIo();
    p->hio = h;                       // (dn_pipe_destroy releases whatever was created)
    const size_t bytes = (size_t)p->B * p->d->cfg.hop * sizeof(float);
    DN_HIP(hipStreamCreateWithFlags(&h->h2d, hipStreamNonBlocking));
    DN_HIP(hipStreamCreateWithFlags(&h->d2h, hipStreamNonBlocking));
    for (int i = 0; i < HostIo::kRing; ++i) {
        DN_HIP(hipEventCreateWithFlags(&h->ev_h2d[i], hipEventDisableTiming));
        DN_HIP(hipEventCreateWithFlags(&h->ev_k[i], hipEventDisableTiming));
        DN_HIP(hipEventCreateWithFlags(&h->ev_d2h[i], hipEventDisableTiming));
    }
    for (int i = 0; i < 2; ++i) {
        DN_HIP(hipMalloc(&h->d_in[i], bytes));
        DN_HIP(hipMalloc(&h->d_out[i], bytes));
    }
    DN_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->done_host), 64, 0));
    *h->done_host = 0;
    void* dv = nullptr;
    DN_HIP(hipHostGetDevicePointer(&dv, h->done_host, 0));
    h->done_dev = static_cast<unsigned long long*>(dv);
    return DN_OK;
}

// the device's own address of page-locked host memory, or null when `host` is pageable
static void* device_view(const void* host) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, host) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (at.type != hipMemoryTypeHost) return nullptr;
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, const_cast<void*>(host), 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return dp;
}

// the deferred output of the newest push, when no further launch will carry it: a copy launch and a one-thread launch that publishes it
static int host_defer_drain(HostIo* h, hipStream_t st) {
    if (!h->defer_pending) return DN_OK;
    dn::launch_host_copy(static_cast<const uint4*>(h->z_out[h->defer_ticket & 1]), static_cast<uint4*>(h->defer_dst), (unsigned int)(h->defer_bytes / 16),
                         h->done_dev, h->defer_ticket + 1, st);
    h->defer_pending = false;
    return check_launch("host_copy_kernel");
}

int dn_pipe_stream_push_host(dn_pipe* p, const void* hop_in_host, int32_t in_is_s16, void* hop_out_host, int32_t out_is_s16, uint64_t seed,
                             uint64_t stream_id0, int32_t n_iter, float momentum, uint32_t flags, void* stream, uint64_t* ticket) {
    if (!p || !p->ring) return fail(DN_ERR_INVALID, "dn_pipe_stream_push_host: not a streaming pipe");
    if (!hop_in_host || !hop_out_host) return fail(DN_ERR_INVALID, "dn_pipe_stream_push_host: null argument");
    if (flags & ~(uint32_t)(DN_HOST_STAGED | DN_HOST_DEFER)) return fail(DN_ERR_INVALID, "dn_pipe_stream_push_host: unknown flag bits");
    int rc = host_io_init(p);
    if (rc != DN_OK) return rc;
    HostIo* h = p->hio;
    hipStream_t cs = as_stream(stream);
    const size_t out_bytes = (size_t)p->B * p->d->cfg.hop * (out_is_s16 ? sizeof(short) : sizeof(float));
    if (p->group > 0) return fail(DN_ERR_UNSUPPORTED, "dn_pipe_stream_push_host: the host-buffer transport moves single hops (dn_pipe_set_group(p, 0))");
    if (!(flags & DN_HOST_STAGED)) {
        // zero copy: page-locked host memory is in the device's address space.  The front workgroups read their stream's hop (1 KB) straight from it
        // and the Griffin-Lim workgroups store the emitted hop straight into it -- no copy engine, no second queue, no cross-queue event.
        void* din = device_view(hop_in_host);
        void* dout = device_view(hop_out_host);
        if (din && dout) {
            const bool defer = (flags & DN_HOST_DEFER) != 0 && out_bytes % 16 == 0;
            void* kout = dout;
            if (defer) {
                // the launch leaves its emitted hop in a device staging buffer; the NEXT launch (or dn_pipe_stream_host_wait) moves it to dout
                void*& z = h->z_out[h->pushes & 1];
                if (!z) DN_HIP(hipMalloc(&z, (size_t)p->B * p->d->cfg.hop * sizeof(float)));
                kout = z;
            }
            if (h->defer_pending) {                              // this launch carries the previous push's hop out, spread over its threads
                p->host_copy_src = h->z_out[h->defer_ticket & 1];
                p->host_copy_dst = h->defer_dst;
                p->host_copy_bytes = h->defer_bytes;
            }
            // picked up by the launch below: its last workgroup publishes how many pushes have their samples in host memory once it returns
            p->host_done = defer && !h->defer_pending ? nullptr : h->done_dev;
            p->host_done_value = defer ? h->pushes : h->pushes + 1;
            rc = dn_pipe_stream_push(p, din, in_is_s16, kout, out_is_s16, nullptr, seed, stream_id0, n_iter, momentum, stream);
            p->host_done = nullptr;
            p->host_copy_src = nullptr; p->host_copy_dst = nullptr; p->host_copy_bytes = 0;
            if (rc != DN_OK) return rc;
            h->defer_pending = defer;
            h->defer_stream = cs;                                  // (also what dn_pipe_stream_host_wait polls for errors)
            if (defer) { h->defer_ticket = h->pushes; h->defer_dst = dout; h->defer_bytes = out_bytes; }
            h->last_zero_copy[h->pushes % HostIo::kRing] = true;
            if (ticket) *ticket = h->pushes;
            ++h->pushes;
            return DN_OK;
        }
    }
    rc = host_defer_drain(h, cs);                                // (a staged push behind a deferred one: its hop leaves first)
    if (rc != DN_OK) return rc;
    const int s = (int)(h->pushes & 1);                                  // device staging buffer
    const int e = (int)(h->pushes % HostIo::kRing), e2 = (int)((h->pushes + HostIo::kRing - 2) % HostIo::kRing);      // this push's events; those of two pushes ago
    const size_t n = (size_t)p->B * p->d->cfg.hop;
    const size_t bin = n * (in_is_s16 ? sizeof(short) : sizeof(float)), bout = n * (out_is_s16 ? sizeof(short) : sizeof(float));
    h->last_zero_copy[h->pushes % HostIo::kRing] = false;
    // upload: the staging buffer was read by the launch two pushes ago
    if (h->pushes >= 2) DN_HIP(hipStreamWaitEvent(h->h2d, h->ev_k[e2], 0));
    DN_HIP(hipMemcpyAsync(h->d_in[s], hop_in_host, bin, hipMemcpyHostToDevice, h->h2d));
    DN_HIP(hipEventRecord(h->ev_h2d[e], h->h2d));
    // the hop: after its samples arrived, and after the download of the result it is about to overwrite
    DN_HIP(hipStreamWaitEvent(cs, h->ev_h2d[e], 0));
    if (h->pushes >= 2) DN_HIP(hipStreamWaitEvent(cs, h->ev_d2h[e2], 0));
    rc = dn_pipe_stream_push(p, h->d_in[s], in_is_s16, h->d_out[s], out_is_s16, nullptr, seed, stream_id0, n_iter, momentum, stream);
    if (rc != DN_OK) return rc;
    DN_HIP(hipEventRecord(h->ev_k[e], cs));
    // download
    DN_HIP(hipStreamWaitEvent(h->d2h, h->ev_k[e], 0));
    DN_HIP(hipMemcpyAsync(hop_out_host, h->d_out[s], bout, hipMemcpyDeviceToHost, h->d2h));
    DN_HIP(hipEventRecord(h->ev_d2h[e], h->d2h));
    if (ticket) *ticket = h->pushes;
    ++h->pushes;
    return DN_OK;
}

int dn_pipe_stream_host_wait(dn_pipe* p, uint64_t ticket) {
    if (!p || !p->hio) return fail(DN_ERR_INVALID, "dn_pipe_stream_host_wait: no host push was made on this pipe");
    HostIo* h = p->hio;
    if (ticket >= h->pushes) return fail(DN_ERR_INVALID, "dn_pipe_stream_host_wait: no such push");
    if (h->defer_pending && ticket >= h->defer_ticket) {         // its samples are still in the staging buffer and no launch is queued to move them
        int rc = host_defer_drain(h, h->defer_stream);
        if (rc != DN_OK) return rc;
    }
    // Wait for one push by its own transport: a zero-copy push through the completion word its launch (or the launch that carried its deferred
    // samples out) publishes, a staged one through the event behind its download.
    auto wait_entry = [&](uint64_t e) -> int {
        if (h->last_zero_copy[e % HostIo::kRing]) {
            // launches of one stream complete in order: the word only grows.  Spin briefly, then yield (a hop is tens of microseconds); a launch that
            // faulted or hangs never publishes, so the stream is polled beside the word: anything but "not ready yet" ends the wait, and "idle" with
            // the word still short of the ticket (the launches ran, the publication did not) is an error too
            const volatile unsigned long long* w = h->done_host;
            for (unsigned spins = 0; __atomic_load_n(w, __ATOMIC_ACQUIRE) < e + 1; ++spins) {
                if (spins > 2000) std::this_thread::yield();
                if ((spins & 0x3fff) == 0x3fff) {
                    const hipError_t q = hipStreamQuery(h->defer_stream);
                    if (q == hipSuccess) {
                        if (__atomic_load_n(w, __ATOMIC_ACQUIRE) >= e + 1) break;
                        return fail(DN_ERR_HIP, "dn_pipe_stream_host_wait: the stream is idle but push " + std::to_string(e) + " was never published");
                    }
                    (void)hipGetLastError();          // (hipErrorNotReady is sticky in the thread's last-error slot)
                    if (q != hipErrorNotReady) return fail(DN_ERR_HIP, std::string("dn_pipe_stream_host_wait: ") + hipGetErrorString(q));
                }
            }
            return DN_OK;
        }
        DN_HIP(hipEventSynchronize(h->ev_d2h[e % HostIo::kRing]));
        return DN_OK;
    };
    // Only the last kRing pushes still own their events and their transport flag.  Either transport completes in order, so an older ticket is
    // complete once the OLDEST surviving push of EACH transport is -- and where the ring holds no staged push any more, once the download queue has
    // drained (a zero-copy launch says nothing about a download on another queue; a staged push does cover the zero-copy launches before it: its
    // download follows its own launch on the caller's stream).
    const uint64_t oldest = h->pushes > (uint64_t)HostIo::kRing ? h->pushes - HostIo::kRing : 0;
    if (ticket >= oldest) return wait_entry(ticket);
    bool seen[2] = {false, false};
    for (uint64_t e = oldest; e < h->pushes; ++e) {
        const int kind = h->last_zero_copy[e % HostIo::kRing] ? 1 : 0;
        if (seen[kind]) continue;
        seen[kind] = true;
        int rc = wait_entry(e);
        if (rc != DN_OK) return rc;
    }
    if (!seen[0] && h->d2h) DN_HIP(hipStreamSynchronize(h->d2h));
    return DN_OK;
}

int dn_pipe_stream_flush(dn_pipe* p, void* hop_out, int32_t out_is_s16, int32_t n_iter, float momentum, void* stream) {
    if (!p || !p->ring) return fail(DN_ERR_INVALID, "dn_pipe_stream_flush: not a streaming pipe");
    if (!hop_out) return fail(DN_ERR_INVALID, "dn_pipe_stream_flush: null argument");
    if (p->group > 0) return fail(DN_ERR_INVALID, "dn_pipe_stream_flush: this pipe emits groups of hops (dn_pipe_stream_flush_group)");
    dn::HopArgs a{};
    int rc = fill_hop_args(p, a, nullptr, 0, 0, n_iter, momentum);
    if (rc != DN_OK) return rc;
    a.front_B = 0;
    a.ola = p->ola; a.hop_out = hop_out; a.out_s16 = out_is_s16;
    dn::launch_hop(p->d->view, p->bs->view, a, p->bf16, as_stream(stream));
    return check_launch("hop_kernel(stream flush)");
}

int dn_pipe_submit(dn_pipe* p, const float* frames, float* hx, float* out, const float* init_angles, uint64_t seed,
                   uint64_t stream_id0, int32_t n_iter, float momentum, void* stream) {
    if (!p || !frames || !hx || !out) return fail(DN_ERR_INVALID, "dn_pipe_submit: null argument");
    if (p->ring) return fail(DN_ERR_INVALID, "dn_pipe_submit: this is a streaming pipe (use dn_pipe_stream_push)");
    if (p->group > 0) return dn_pipe_submit_group(p, frames, 0, hx, out, 0, init_angles, 0, seed, stream_id0, 1, n_iter, momentum, stream);
    dn::HopArgs a{};
    int rc = fill_hop_args(p, a, init_angles, seed, stream_id0, n_iter, momentum);
    if (rc != DN_OK) return rc;
    a.front_B = p->B; a.frames = frames; a.hx = hx;
    a.gl_out = out;                  // THIS hop's destination: its front workgroups leave it in the slot for the Griffin-Lim workgroups of the next launch
    dn::launch_hop(p->d->view, p->bs->view, a, p->bf16, as_stream(stream));
    rc = check_launch("hop_kernel");
    if (rc != DN_OK) return rc;
    p->submitted = true;
    return DN_OK;
}

int dn_pipe_flush(dn_pipe* p, int32_t n_iter, float momentum, void* stream) {
    if (!p) return fail(DN_ERR_INVALID, "dn_pipe_flush: null pipe");
    if (p->ring) return fail(DN_ERR_INVALID, "dn_pipe_flush: this is a streaming pipe (use dn_pipe_stream_flush)");
    if (!p->submitted) return DN_OK; // nothing was ever submitted
    dn::HopArgs a{};
    int rc = fill_hop_args(p, a, nullptr, 0, 0, n_iter, momentum);
    if (rc != DN_OK) return rc;
    a.front_B = 0;
    if (p->group > 0) {                             // one launch: the whole chains of the group in flight
        dn::launch_group(p->d->view, p->bs->view, a, p->bf16, as_stream(stream));
        return check_launch("group_kernel(flush)");
    }
    for (int i = 0; i < p->depth; ++i) {            // every launch advances each hop in flight by one chain segment
        dn::launch_hop(p->d->view, p->bs->view, a, p->bf16, as_stream(stream));
        rc = check_launch("hop_kernel(flush)");
        if (rc != DN_OK) return rc;
    }
    return DN_OK;
}

int dn_pipe_submit_group(dn_pipe* p, const float* frames, int64_t frames_stride, float* hx, float* out, int64_t out_stride,
                         const float* init_angles, int64_t init_stride, uint64_t seed, uint64_t stream_id0, int32_t hops,
                         int32_t n_iter, float momentum, void* stream) {
    if (!p || !frames || !hx || !out) return fail(DN_ERR_INVALID, "dn_pipe_submit_group: null argument");
    if (p->ring) return fail(DN_ERR_INVALID, "dn_pipe_submit_group: this is a streaming pipe (use dn_pipe_stream_push_group)");
    if (p->group <= 0) return fail(DN_ERR_INVALID, "dn_pipe_submit_group: not a group pipe (dn_pipe_set_group)");
    if (hops < 1 || hops > p->group) return fail(DN_ERR_INVALID, "dn_pipe_submit_group: hops must be in 1.." + std::to_string(p->group));
    if (frames_stride < 0 || out_stride < 0 || init_stride < 0) return fail(DN_ERR_INVALID, "dn_pipe_submit_group: negative stride");
    const int64_t line = (int64_t)p->B * p->d->cfg.n_fft;
    if (hops > 1 && out_stride < line) return fail(DN_ERR_INVALID, "dn_pipe_submit_group: the output frames of a group overlap (out_stride < B * n_fft)");
    dn::HopArgs a{};
    int rc = fill_hop_args(p, a, init_angles, seed, stream_id0, n_iter, momentum);
    if (rc != DN_OK) return rc;
    a.front_B = p->B; a.frames = frames; a.hx = hx; a.gl_out = out;
    a.group_hops = hops; a.frames_stride = frames_stride; a.out_stride = out_stride; a.init_in_stride = init_stride;
    dn::launch_group(p->d->view, p->bs->view, a, p->bf16, as_stream(stream));
    rc = check_launch("group_kernel");
    if (rc != DN_OK) return rc;
    p->submitted = true;
    return DN_OK;
}

int dn_pipe_stream_push_group(dn_pipe* p, const void* hop_in, int64_t in_stride, int32_t in_is_s16, void* hop_out, int64_t out_stride,
                              int32_t out_is_s16, const float* init_angles, int64_t init_stride, uint64_t seed, uint64_t stream_id0,
                              int32_t n_iter, float momentum, void* stream) {
    if (!p || !p->ring) return fail(DN_ERR_INVALID, "dn_pipe_stream_push_group: not a streaming pipe");
    if (!hop_in || !hop_out) return fail(DN_ERR_INVALID, "dn_pipe_stream_push_group: null argument");
    if (p->group <= 0) return fail(DN_ERR_INVALID, "dn_pipe_stream_push_group: not a group pipe (dn_pipe_set_group)");
    const int64_t row = (int64_t)p->B * p->d->cfg.hop;
    if (in_stride < 0 || init_stride < 0 || (p->group > 1 && out_stride < row))
        return fail(DN_ERR_INVALID, "dn_pipe_stream_push_group: bad stride (the emitted hops of a group must not overlap)");
    dn::HopArgs a{};
    int rc = fill_hop_args(p, a, init_angles, seed, stream_id0, n_iter, momentum);
    if (rc != DN_OK) return rc;
    a.front_B = p->B; a.hx = p->hx;
    a.hop_in = hop_in; a.ring = p->ring; a.in_s16 = in_is_s16;
    a.ola = p->ola; a.hop_out = hop_out; a.out_s16 = out_is_s16;
    a.group_hops = p->group; a.group_out = p->group; a.filler_first = 1;
    a.hop_in_stride = in_stride; a.hop_out_stride = out_stride; a.init_in_stride = init_stride;
    dn::launch_group(p->d->view, p->bs->view, a, p->bf16, as_stream(stream));
    rc = check_launch("group_kernel(stream)");
    if (rc != DN_OK) return rc;
    const unsigned long long before = p->group_pushes;
    p->group_pushes += (unsigned long long)p->group;
    const unsigned long long prime = (unsigned long long)a.prime;
    const unsigned long long primed_before = before < prime ? before : prime, primed_after = p->group_pushes < prime ? p->group_pushes : prime;
    p->group_pending = p->group - (int)(primed_after - primed_before);
    return DN_OK;
}

int dn_pipe_stream_flush_group(dn_pipe* p, void* hop_out, int64_t out_stride, int32_t out_is_s16, int32_t* hops_valid, void* stream) {
    if (!p || !p->ring) return fail(DN_ERR_INVALID, "dn_pipe_stream_flush_group: not a streaming pipe");
    if (!hop_out) return fail(DN_ERR_INVALID, "dn_pipe_stream_flush_group: null argument");
    if (p->group <= 0) return fail(DN_ERR_INVALID, "dn_pipe_stream_flush_group: not a group pipe (dn_pipe_set_group)");
    if (p->group > 1 && out_stride < (int64_t)p->B * p->d->cfg.hop) return fail(DN_ERR_INVALID, "dn_pipe_stream_flush_group: the emitted hops overlap");
    dn::HopArgs a{};
    int rc = fill_hop_args(p, a, nullptr, 0, 0, 0, 0.0f);
    if (rc != DN_OK) return rc;
    a.front_B = 0;
    a.ola = p->ola; a.hop_out = hop_out; a.out_s16 = out_is_s16;
    a.group_hops = 0; a.group_out = p->group; a.filler_first = 0; a.hop_out_stride = out_stride;
    dn::launch_group(p->d->view, p->bs->view, a, p->bf16, as_stream(stream));
    rc = check_launch("group_kernel(stream flush)");
    if (rc != DN_OK) return rc;
    if (hops_valid) *hops_valid = p->group_pending;
    p->group_pending = 0;
    return DN_OK;
}

}  // extern "C"
