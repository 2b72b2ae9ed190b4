// wave_place.hip -- where the dispatcher puts the wavefronts of a workgroup (tools only, not product).
// Every wave records its SIMD / CU / SE / XCC (HW_ID, XCC_ID) and the interval it was resident; the summary says, per
// configuration (block size, waves that exit at once, LDS per workgroup), how the long-running waves are spread over the four
// SIMDs of a CU -- the question behind "SIMD 3 idles" (DESIGN.md section 8).
//   hipcc -O3 --offload-arch=gfx950 -o tools/build/wave_place tools/wave_place.hip && tools/build/wave_place
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <vector>

struct Rec { uint32_t hw, xcc, block, wave; uint64_t t0, t1; };

// waves with index >= live exit at once; the others spin ~spin iterations of dependent FMAs
__global__ void place_kernel(Rec* rec, int live, int first_live, int spin, int lds_words, float seed) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int rel = (wave - first_live + nw) % nw;
    if (rel >= live) return;
    uint32_t hw, xcc;
    uint64_t t0, t1;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    float a = seed + threadIdx.x;
    for (int i = 0; i < spin; ++i) a = fmaf(a, 1.0000001f, 0.5f);
    if (lds_words > 0) lds[threadIdx.x % lds_words] = a;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if ((threadIdx.x & 63) == 0) {
        Rec r{hw, xcc, blockIdx.x, (uint32_t)wave, t0, t1};
        rec[blockIdx.x * nw + wave] = r;
    }
    if (a == 12345.678f) rec[0].hw = 0;
}

static void run(const char* name, int grid, int block, int live, int rotate, int lds_bytes, int spin) {
    const int nw = block / 64;
    Rec* d;
    hipMalloc(&d, sizeof(Rec) * grid * nw);
    hipMemset(d, 0xff, sizeof(Rec) * grid * nw);
    hipFuncSetAttribute(reinterpret_cast<const void*>(place_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    // rotate: the first live wave of block b is b % nw (so the exiting wave is not always the last one)
    if (!rotate) hipLaunchKernelGGL(place_kernel, dim3(grid), dim3(block), lds_bytes, 0, d, live, 0, spin, lds_bytes / 4, 1.0f);
    else {
        // one launch per rotation value would change the dispatch; emulate with first_live = 0 only when rotate == 0
        hipLaunchKernelGGL(place_kernel, dim3(grid), dim3(block), lds_bytes, 0, d, live, rotate, spin, lds_bytes / 4, 1.0f);
    }
    hipDeviceSynchronize();
    std::vector<Rec> h(grid * nw);
    hipMemcpy(h.data(), d, sizeof(Rec) * grid * nw, hipMemcpyDeviceToHost);
    hipFree(d);
    // histogram: SIMD of wave index w; waves per (xcc, se, cu) per simd
    long simd_of_wave[8][4];
    memset(simd_of_wave, 0, sizeof(simd_of_wave));
    std::map<uint32_t, std::vector<long>> per_cu;
    long total = 0;
    for (auto& r : h) {
        if (r.hw == 0xffffffffu) continue;
        const int simd = (r.hw >> 4) & 3, cu = (r.hw >> 8) & 15, sh = (r.hw >> 12) & 1, se = (r.hw >> 13) & 7;
        simd_of_wave[r.wave & 7][simd]++;
        const uint32_t key = ((r.xcc & 15) << 16) | (se << 8) | (sh << 4) | cu;
        auto& v = per_cu[key];
        if (v.empty()) v.assign(4, 0);
        v[simd]++;
        ++total;
    }
    printf("== %s: grid %d x %d threads, %d live waves (first live %d), lds %d B, %ld waves recorded, %zu CUs seen\n", name, grid, block, live,
           rotate, lds_bytes, total, per_cu.size());
    for (int w = 0; w < nw; ++w)
        printf("   wave %d -> SIMD0 %ld  SIMD1 %ld  SIMD2 %ld  SIMD3 %ld\n", w, simd_of_wave[w][0], simd_of_wave[w][1], simd_of_wave[w][2], simd_of_wave[w][3]);
    long s[4] = {0, 0, 0, 0};
    for (auto& kv : per_cu)
        for (int i = 0; i < 4; ++i) s[i] += kv.second[i];
    printf("   all CUs: waves per SIMD %ld %ld %ld %ld\n", s[0], s[1], s[2], s[3]);
    // how the first half of the grid (the Griffin-Lim workgroups of a hop launch) and the second half (the front workgroups) pair up on the CUs
    {
        std::map<uint32_t, std::pair<int, int>> kinds;
        for (int b = 0; b < grid; ++b) {
            const Rec* r = nullptr;
            for (int w = 0; w < nw; ++w) if (h[b * nw + w].hw != 0xffffffffu) { r = &h[b * nw + w]; break; }
            if (!r) continue;
            const uint32_t key = ((r->xcc & 15) << 16) | (((r->hw >> 13) & 7) << 8) | (((r->hw >> 12) & 1) << 4) | ((r->hw >> 8) & 15);
            if (b < grid / 2) kinds[key].first++; else kinds[key].second++;
        }
        std::map<std::pair<int, int>, int> hist;
        for (auto& kv : kinds) hist[kv.second]++;
        printf("   CUs by (workgroups of the first half of the grid, of the second half):");
        for (auto& kv : hist) printf("  (%d,%d) x %d", kv.first.first, kv.first.second, kv.second);
        printf("\n");
    }
    // the first three workgroups: which SIMDs their waves got, and on which CU
    for (int b = 0; b < 3 && b < grid; ++b) {
        printf("   block %d:", b);
        for (int w = 0; w < nw; ++w) {
            const Rec& r = h[b * nw + w];
            if (r.hw == 0xffffffffu) { printf("  w%d -", w); continue; }
            printf("  w%d simd %u cu %u se %u xcc %u", w, (r.hw >> 4) & 3, (r.hw >> 8) & 15, (r.hw >> 13) & 7, r.xcc & 15);
        }
        printf("\n");
    }
}

int main() {
    const int spin = 20000;
    run("4 waves, all live, 2 wg/CU by LDS", 2048, 256, 4, 0, 74 * 1024, spin);
    run("4 waves, wave 3 exits, 2 wg/CU by LDS", 2048, 256, 3, 0, 74 * 1024, spin);
    run("4 waves, wave 0 exits (live = 1,2,3)", 2048, 256, 3, 1, 74 * 1024, spin);
    run("3 waves (192 threads), 2 wg/CU by LDS", 2048, 192, 3, 0, 74 * 1024, spin);
    run("3 waves (192 threads), 4 wg/CU by LDS", 4096, 192, 3, 0, 36 * 1024, spin);
    run("3 waves, one wg per CU (grid 256)", 256, 192, 3, 0, 74 * 1024, spin);
    run("4 waves, grid 512 (the batch-256 launch shape)", 512, 256, 4, 0, 74 * 1024, spin);
    run("4 waves, grid 512, short", 512, 256, 4, 0, 74 * 1024, 200);
    run("4 waves, grid 2048 + 8192 shape / 4", 2560, 256, 4, 0, 74 * 1024, 2000);
    return 0;
}
