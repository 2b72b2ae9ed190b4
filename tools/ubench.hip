// ubench.hip -- gfx950 micro-benchmarks behind the design decisions of the Griffin-Lim loop (tools only, not product):
// issue cost of packed / scalar fp32 VALU for 1..3 waves per SIMD, latency of the wave-private LDS tile exchange, of
// ds_bpermute rounds and of v_permlane32_swap, and the price of a workgroup barrier.
//   hipcc -O3 --offload-arch=gfx950 -o tools/build/ubench tools/ubench.hip && tools/build/ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")

constexpr int kIters = 256;

template <int MODE>
__global__ void valu_kernel(float* out, uint64_t* cyc, float seed) {
    v2f a[8], b, c;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = v2f{seed + i + threadIdx.x, seed * i};
    b = v2f{seed * 0.5f, seed * 0.25f};
    c = v2f{seed * 0.125f, seed};
    uint64_t t0, t1;
    __builtin_amdgcn_sched_barrier(0);
    STAMP(t0);
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (MODE == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i][0]) : "v"(b[0]), "v"(c[0]));
                if (MODE == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i][0]) : "v"(b[0]));
                if (MODE == 5) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[i]) : "v"(b));
                if (MODE == 6) {   // alternating pk and scalar
                    if (i & 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i][0]) : "v"(b[0]), "v"(c[0]));
                }
                if (MODE == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i][0]));
                if (MODE == 8) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i][0]) : "v"(b[0]));
                if (MODE == 9) asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xc" : "+v"(a[i][0]) : "v"(b[0]));
            }
    }
    __builtin_amdgcn_sched_barrier(0);
    STAMP(t1);
    __builtin_amdgcn_sched_barrier(0);
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i][0] + a[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// dependent chain of one instruction kind (latency)
template <int MODE>
__global__ void chain_kernel(float* out, uint64_t* cyc, float seed) {
    v2f a = v2f{seed + threadIdx.x, seed}, b = v2f{seed * 0.5f, seed * 0.25f}, c = v2f{seed * 0.125f, seed};
    uint64_t t0, t1;
    __builtin_amdgcn_sched_barrier(0);
    STAMP(t0);
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
            if (MODE == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b[0]), "v"(c[0]));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    STAMP(t1);
    __builtin_amdgcn_sched_barrier(0);
    out[blockIdx.x * blockDim.x + threadIdx.x] = a[0] + a[1];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// MODE 0: wave-private tile exchange (8 ds_write_b64 scattered, 8 ds_read_b64 strided), dependent round trips
// MODE 1: 8 x 2 ds_bpermute_b32 round
// MODE 2: 8 x 2 v_permlane32_swap
// MODE 3: exchange + __syncthreads (all waves)
// MODE 4: 16 ds_write_b32 + 16 ds_read_b32 (same bytes as mode 0)
// MODE 5: 4 ds_write_b128 + 4 ds_read_b128
template <int MODE>
__global__ void xchg_kernel(float* out, uint64_t* cyc, float seed) {
    __shared__ v2f tile[8][640];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v2f* t = tile[w];
    v2f v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = v2f{seed + i + lane, seed * i};
    uint64_t t0, t1;
    const int partner = (64 - lane) & 63;
    __builtin_amdgcn_sched_barrier(0);
    STAMP(t0);
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < kIters; ++it) {
        if (MODE == 0 || MODE == 3) {
            wave_sync();
#pragma unroll
            for (int i = 0; i < 8; ++i) { const int c = 8 * lane + i; t[c + (c >> 4)] = v[i]; }
            wave_sync();
            if (MODE == 3) __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; ++i) { const int c = lane + 64 * i; v[i] = t[c + (c >> 4)]; }
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = v[i] + v[(i + 1) & 7] * 0.5f;
        }
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = v2f{__shfl(v[7 - i][0], partner), __shfl(v[7 - i][1], partner)};
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = v[i] + v[(i + 1) & 7] * 0.5f;
        }
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(v[i][0]), "+v"(v[i + 4][0]));
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(v[i][1]), "+v"(v[i + 4][1]));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = v[i] + v[(i + 1) & 7] * 0.5f;
        }
        if (MODE == 4) {
            float* tf = reinterpret_cast<float*>(t);
            wave_sync();
#pragma unroll
            for (int i = 0; i < 8; ++i) { tf[16 * lane + 2 * i + (lane >> 1)] = v[i][0]; tf[16 * lane + 2 * i + 1 + (lane >> 1)] = v[i][1]; }
            wave_sync();
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[i][0] = tf[lane + 128 * i]; v[i][1] = tf[lane + 128 * i + 64]; }
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = v[i] + v[(i + 1) & 7] * 0.5f;
        }
        if (MODE == 5) {
            float4* t4 = reinterpret_cast<float4*>(t);
            wave_sync();
#pragma unroll
            for (int i = 0; i < 4; ++i) t4[lane * 4 + i + (lane >> 3)] = make_float4(v[2 * i][0], v[2 * i][1], v[2 * i + 1][0], v[2 * i + 1][1]);
            wave_sync();
#pragma unroll
            for (int i = 0; i < 4; ++i) { float4 q = t4[lane + 64 * i + i]; v[2 * i] = v2f{q.x, q.y}; v[2 * i + 1] = v2f{q.z, q.w}; }
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = v[i] + v[(i + 1) & 7] * 0.5f;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    STAMP(t1);
    __builtin_amdgcn_sched_barrier(0);
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + w] = t1 - t0;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// MODE 0: one dependent accumulator chain of v_mfma_f32_16x16x4_f32; 1: four independent accumulators; 2: dependent chain whose A and B
// operands come from ds_read_b32 issued just before (the conv tiles of the cell kernel); 3: v_mfma_f32_16x16x32_bf16 dependent chain
template <int MODE>
__global__ void mfma_kernel(float* out, uint64_t* cyc, float seed) {
    __shared__ float lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = seed * i;
    __syncthreads();
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{seed, seed, seed, seed};
    float a = seed + lane, b = seed * lane;
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(seed + i); bb[i] = (__bf16)(seed * i); }
    uint64_t t0, t1;
    __builtin_amdgcn_sched_barrier(0);
    STAMP(t0);
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (MODE == 0) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[0], 0, 0, 0);
            if (MODE == 1) acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[r & 3], 0, 0, 0);
            if (MODE == 2) {
                const float aa = lds[(r * 64 + lane + it) & 4095], bb2 = lds[(r * 67 + 2 * lane + it) & 4095];
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa, bb2, acc[0], 0, 0, 0);
            }
            if (MODE == 3) acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[0], 0, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    STAMP(t1);
    __builtin_amdgcn_sched_barrier(0);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <typename K>
static int run(const char* name, K kern, int blocks, int threads, double per, float* out, uint64_t* cyc) {
    const int nw = blocks * threads / 64;
    CK(hipMemset(cyc, 0, nw * 8));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0f);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0f);
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> h(nw);
    CK(hipMemcpy(h.data(), cyc, nw * 8, hipMemcpyDeviceToHost));
    double mn = 1e30, mx = 0, sum = 0;
    for (auto c : h) { mn = c < mn ? c : mn; mx = c > mx ? c : mx; sum += c; }
    // s_memtime ticks at 100 MHz-derived "shader clock"? report raw ticks per unit
    printf("%-46s blocks %4d thr %4d : ticks/unit min %.2f avg %.2f max %.2f\n", name, blocks, threads, mn / per, sum / nw / per, mx / per);
    return 0;
}

int main() {
    float* out; uint64_t* cyc;
    CK(hipMalloc(&out, 1 << 24));
    CK(hipMalloc(&cyc, 1 << 20));
    const double nv = (double)kIters * 32;
    const int thr[] = {64, 256, 512};
    for (int t : thr) {
        run("v_pk_fma_f32 indep (per instr)", valu_kernel<0>, 256, t, nv, out, cyc);
        run("v_pk_add_f32 indep", valu_kernel<1>, 256, t, nv, out, cyc);
        run("v_pk_mul_f32 indep", valu_kernel<2>, 256, t, nv, out, cyc);
        run("v_fma_f32 indep", valu_kernel<3>, 256, t, nv, out, cyc);
        run("v_add_f32 indep", valu_kernel<4>, 256, t, nv, out, cyc);
        run("v_pk_add_f32 op_sel/neg indep", valu_kernel<5>, 256, t, nv, out, cyc);
        run("alternating pk_fma / fma", valu_kernel<6>, 256, t, nv, out, cyc);
        run("v_rcp_f32", valu_kernel<7>, 256, t, nv, out, cyc);
        run("v_mov_b32", valu_kernel<8>, 256, t, nv, out, cyc);
        run("v_mov_b32_dpp row_ror:8 bank_mask", valu_kernel<9>, 256, t, nv, out, cyc);
    }
    run("v_pk_fma_f32 dependent chain (per instr)", chain_kernel<0>, 256, 64, nv, out, cyc);
    run("v_fma_f32 dependent chain", chain_kernel<1>, 256, 64, nv, out, cyc);
    run("v_pk_fma_f32 dependent chain 2w/SIMD", chain_kernel<0>, 256, 512, nv, out, cyc);
    for (int t : {64, 256}) {
        run("mfma f32 16x16x4 dependent chain (per mfma)", mfma_kernel<0>, 256, t, (double)kIters * 16, out, cyc);
        run("mfma f32 16x16x4 four accumulators", mfma_kernel<1>, 256, t, (double)kIters * 16, out, cyc);
        run("mfma f32 16x16x4 dependent + 2 ds_read_b32 each", mfma_kernel<2>, 256, t, (double)kIters * 16, out, cyc);
        run("mfma bf16 16x16x32 dependent chain", mfma_kernel<3>, 256, t, (double)kIters * 16, out, cyc);
    }
    const int thr2[] = {64, 192, 256, 384, 512};
    for (int t : thr2) {
        run("tile exchange b64 round trip (per round)", xchg_kernel<0>, 256, t, kIters, out, cyc);
        run("bpermute round 16 dwords", xchg_kernel<1>, 256, t, kIters, out, cyc);
        run("permlane32_swap round 8 swaps", xchg_kernel<2>, 256, t, kIters, out, cyc);
        run("tile exchange + __syncthreads", xchg_kernel<3>, 256, t, kIters, out, cyc);
        run("tile exchange b32 x16", xchg_kernel<4>, 256, t, kIters, out, cyc);
        run("tile exchange b128 x4", xchg_kernel<5>, 256, t, kIters, out, cyc);
    }
    return 0;
}
