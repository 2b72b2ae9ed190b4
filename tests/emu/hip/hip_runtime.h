// TEST INFRASTRUCTURE ONLY -- a functional stand-in for <hip/hip_runtime.h> that lets g++
// compile the package's .hip kernel sources for the HOST so kernel *logic* (index maths, FFT
// passes, LDS hand-offs) can be checked in the GPU-less build container:
//   * one workgroup at a time; one std::thread per work-item
//   * __syncthreads -> a workgroup barrier; wave-level primitives -> per-64-thread barriers
//   * __shared__ -> static storage (valid because workgroups run one after another)
//   * hipMalloc/hipMemcpy/... -> malloc/memcpy
// Nothing in the product (audio-denoising_amd/) includes or links this; the shipped library is
// built by hipcc from the same sources against the real HIP headers.  It proves nothing about
// memory-model races or performance -- that is what the -m gpu tests are for.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <tuple>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define DN_CONST_AS

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct uint4 { unsigned int x, y, z, w; };
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
typedef struct dn_emu_stream* hipStream_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); return *p ? 0 : 1; }
inline hipError_t hipFree(void* p) { free(p); return 0; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return 0; }
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return 0; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
struct short4 { short x, y, z, w; };
// streams and events: the emulation is synchronous, so these only have to exist
typedef struct dn_emu_event* hipEvent_t;
constexpr unsigned hipStreamNonBlocking = 1, hipEventDisableTiming = 2;
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return 0; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
constexpr hipError_t hipErrorNotReady = 600;
inline hipError_t hipStreamQuery(hipStream_t) { return 0; }          // (launches run to completion inside the launch call)
inline hipError_t hipDeviceSynchronize() { return 0; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return 0; }
inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
// pointer attributes: everything is "host pageable" here, so the staged path of dn_pipe_stream_push_host is what the emulation runs
enum hipMemoryType { hipMemoryTypeHost, hipMemoryTypeDevice, hipMemoryTypeUnregistered };
struct hipPointerAttribute_t { hipMemoryType type; };
// DN_EMU_PINNED=1 (set by a test around its calls): every host pointer counts as page-locked, so the zero-copy transports can be emulated
inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void*) {
    const char* e = getenv("DN_EMU_PINNED");
    a->type = (e && e[0] == '1') ? hipMemoryTypeHost : hipMemoryTypeUnregistered;
    return 0;
}
inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return 0; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n); return *p ? 0 : 1; }
inline hipError_t hipHostFree(void* p) { free(p); return 0; }
// system-scope fence / store: the emulation is one process, sequentially consistent atomics will do
inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
#define __HIP_MEMORY_SCOPE_SYSTEM 5
template <typename T> inline void __hip_atomic_store(T* p, T v, int, int) { __atomic_store_n(p, v, __ATOMIC_SEQ_CST); }
template <typename T> inline T __hip_atomic_fetch_add(T* p, T v, int, int) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
#define DN_WAIT_VMEM() ((void)0)
#define DN_OPAQUE_ZERO(z) ((z) = 0)
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }

namespace dn_emu {

class Barrier {
    std::mutex m_;
    std::condition_variable cv_;
    int n_, count_ = 0, gen_ = 0;
public:
    explicit Barrier(int n) : n_(n) {}
    void wait() {
        std::unique_lock<std::mutex> lk(m_);
        int g = gen_;
        if (++count_ == n_) { count_ = 0; ++gen_; cv_.notify_all(); }
        else cv_.wait(lk, [&] { return gen_ != g; });
    }
    // a work-item that has returned no longer counts (as a terminated wavefront does not count at s_barrier)
    void leave() {
        std::unique_lock<std::mutex> lk(m_);
        --n_;
        if (n_ > 0 && count_ == n_) { count_ = 0; ++gen_; cv_.notify_all(); }
    }
};

struct Ctx {
    Barrier* block;
    Barrier* wave;
    float* shfl;          // per-wave 64-slot scratch
};
inline thread_local Ctx ctx;
inline thread_local dim3 tIdx, bIdx, bDim, gDim;

template <typename K, typename... A>
void launch(K kernel, dim3 grid, dim3 block, A... args) {
    const int nt = (int)block.x, nw = (nt + 63) / 64;
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        Barrier bb(nt);
        std::vector<std::unique_ptr<Barrier>> wb;
        std::vector<std::vector<float>> ws(nw, std::vector<float>(64));
        for (int w = 0; w < nw; ++w) wb.emplace_back(new Barrier(std::min(64, nt - 64 * w)));
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t)
            th.emplace_back([&, t] {
                ctx = Ctx{&bb, wb[t / 64].get(), ws[t / 64].data()};
                tIdx = dim3(t); bIdx = dim3(bx); bDim = block; gDim = grid;
                kernel(args...);
                bb.leave();
                wb[t / 64]->leave();
            });
        for (auto& x : th) x.join();
    }
}

inline float shfl(float v, int src) {
    ctx.shfl[tIdx.x & 63] = v;
    ctx.wave->wait();
    float r = ctx.shfl[src & 63];
    ctx.wave->wait();
    return r;
}

}  // namespace dn_emu

#define threadIdx dn_emu::tIdx
#define blockIdx dn_emu::bIdx
#define blockDim dn_emu::bDim
#define gridDim dn_emu::gDim
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) dn_emu::launch(kernel, grid, block, __VA_ARGS__)

inline void __syncthreads() { dn_emu::ctx.block->wait(); }
// workgroups run one after another and one thread per workgroup takes the ticket: a plain read-modify-write is enough
inline unsigned int atomicAdd(unsigned int* p, unsigned int v) { unsigned int o = *p; *p = o + v; return o; }
#define DN_LDS_BARRIER() __syncthreads()
#define DN_WAVE_REDUCE_SHFL 1
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_wave_barrier() dn_emu::ctx.wave->wait()
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
inline float __shfl_xor(float v, int mask) { return dn_emu::shfl(v, (int)(dn_emu::tIdx.x & 63) ^ mask); }
inline float __shfl(float v, int src) { return dn_emu::shfl(v, src); }
using std::max;
using std::min;
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
inline float __builtin_amdgcn_exp2f(float x) { return std::exp2(x); }
inline float __builtin_amdgcn_logf(float x) { return std::log2(x); }
inline float __builtin_amdgcn_sqrtf(float x) { return std::sqrt(x); }
inline float __builtin_amdgcn_rsqf(float x) { return 1.0f / std::sqrt(x); }

// v_mfma_f32_16x16x4_f32 emulated as a wave-collective op: D = A(16x4) B(4x16) + C with
// lane l supplying A[l & 15][l >> 4], B[l >> 4][l & 15] and holding D[(l >> 4) * 4 + r][l & 15].
#define DN_F32X4
typedef float f32x4 __attribute__((vector_size(16)));
inline f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, f32x4 c, int, int, int) {
    using namespace dn_emu;
    const int l = tIdx.x & 63;
    static std::vector<std::vector<float>> pool(16, std::vector<float>(128));   // [wave of the block][A | B]
    float* s = pool[tIdx.x / 64].data();
    s[l] = a;
    s[64 + l] = b;
    ctx.wave->wait();
    const int col = l & 15, q = l >> 4;
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = q * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = std::fmaf(s[k * 16 + row], s[64 + k * 16 + col], acc);
        d[r] = acc;
    }
    ctx.wave->wait();
    return d;
}

// v_mfma_f32_16x16x32_bf16: lane l supplies A[l & 15][8 (l >> 4) + j], B[8 (l >> 4) + j][l & 15], j = 0..7.
// (bf16x8 comes from the dn_cpx.hpp stand-in; declared as a template so this header does not depend on it.)
template <typename FRAG>
inline f32x4 __builtin_amdgcn_mfma_f32_16x16x32_bf16(const FRAG& a, const FRAG& b, f32x4 c, int, int, int) {
    using namespace dn_emu;
    const int l = tIdx.x & 63;
    static std::vector<std::vector<float>> pool(16, std::vector<float>(1024));   // [wave][A 16x32 | B 32x16]
    float* s = pool[tIdx.x / 64].data();
    for (int j = 0; j < 8; ++j) {
        s[(l & 15) * 32 + 8 * (l >> 4) + j] = a[j].to_float();
        s[512 + (8 * (l >> 4) + j) * 16 + (l & 15)] = b[j].to_float();
    }
    ctx.wave->wait();
    const int col = l & 15, q = l >> 4;
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = q * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) acc = std::fmaf(s[row * 32 + k], s[512 + k * 16 + col], acc);
        d[r] = acc;
    }
    ctx.wave->wait();
    return d;
}
