// A host that is NOT Python: the per-hop path driven through the C ABI alone (include/dn_denoise.h) -- what a cgo /
// JNI / N-API binding would do.  Reads a flat GRUUNet2 state_dict blob, runs n hops of B streams with the library's
// native filterbank / window / device RNG, writes the raw float32 output of the last hop.
//   hipcc -O2 examples/denoise_hop.cpp -Iinclude -Laudio-denoising_amd/lib -ldn_denoise -Wl,-rpath,$PWD/audio-denoising_amd/lib -o /tmp/denoise_hop
//   /tmp/denoise_hop tests/golden/weights_dari_tult.bin out.f32 [B] [hops] [group]
// group = 0 (default): one dn_process_frame per hop (no added latency).  group = 1..4: the same hops through a pipe in groups of that many hops per
// launch (dn_pipe_set_group / dn_pipe_submit_group: whole Griffin-Lim chains per launch) -- the f-th frame of a pipe draws from seed + f, so both
// ways write the same bits.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "dn_denoise.h"

#define CHECK_DN(x) do { int rc_ = (x); if (rc_ != DN_OK) { fprintf(stderr, "%s: %s\n", #x, dn_last_error()); return 2; } } while (0)
#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s weights.bin out.f32 [B] [hops] [group]\n", argv[0]); return 1; }
    const int B = argc > 3 ? atoi(argv[3]) : 4, hops = argc > 4 ? atoi(argv[4]) : 3, group = argc > 5 ? atoi(argv[5]) : 0;
    const int N = 1024, HOP = 512, M = 80, C = M / 16;
    std::vector<float> w(DN_MODEL_N_FLOATS);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(w.data(), sizeof(float), w.size(), f) != w.size()) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
    fclose(f);

    dn_model_cfg mc = {C, 1, 4, 17, 3, 2, 1, 6};
    dn_dsp_cfg dc = {16000, N, HOP, M};
    dn_model* model = nullptr;
    dn_dsp* plan = nullptr;
    CHECK_DN(dn_model_create(w.data(), w.size(), &mc, &model));
    CHECK_DN(dn_dsp_create(&dc, nullptr, nullptr, nullptr, &plan));          // native HTK filterbank, pinv, periodic Hann

    float *frames, *hx, *out;
    void* ws;
    const size_t line = (size_t)B * N, n_lines = group > 0 ? (size_t)hops : 1;          // (a group pipe reads and writes whole groups: every hop keeps its buffers)
    CHECK_HIP(hipMalloc((void**)&frames, sizeof(float) * line * n_lines));
    CHECK_HIP(hipMalloc((void**)&out, sizeof(float) * line * n_lines));
    CHECK_HIP(hipMalloc((void**)&hx, sizeof(float) * B * 17 * C));
    CHECK_HIP(hipMemset(hx, 0, sizeof(float) * B * 17 * C));
    CHECK_HIP(hipMalloc(&ws, dn_workspace_bytes(plan, B)));
    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));

    std::vector<float> h(B * N);
    for (int hop = 0; hop < hops; ++hop) {
        for (int b = 0; b < B; ++b)
            for (int n = 0; n < N; ++n) {
                const double t = (hop * HOP + n) / 16000.0;
                h[b * N + n] = (float)(0.3 * sin(2 * M_PI * (220.0 + 110.0 * b) * t) + 0.05 * sin(2 * M_PI * 3300.0 * t + b));
            }
        if (group > 0) {
            CHECK_HIP(hipMemcpyAsync(frames + line * hop, h.data(), sizeof(float) * line, hipMemcpyHostToDevice, st));
            CHECK_HIP(hipStreamSynchronize(st));          // (h is reused for the next hop)
            continue;
        }
        CHECK_HIP(hipMemcpyAsync(frames, h.data(), sizeof(float) * B * N, hipMemcpyHostToDevice, st));
        CHECK_DN(dn_process_frame(model, plan, frames, hx, out, nullptr, nullptr, /*seed*/ 2024 + hop, /*stream_id0*/ 0, 32, 0.99f, ws, B, /*flags*/ 0, st));
    }
    dn_pipe* pipe = nullptr;
    if (group > 0) {
        CHECK_DN(dn_pipe_create(model, plan, B, 0, &pipe));
        CHECK_DN(dn_pipe_set_group(pipe, group));
        for (int hop = 0; hop < hops; hop += group) {
            const int k = hops - hop < group ? hops - hop : group;
            CHECK_DN(dn_pipe_submit_group(pipe, frames + line * hop, (int64_t)line, hx, out + line * hop, (int64_t)line, nullptr, 0, /*seed*/ 2024,
                                          /*stream_id0*/ 0, k, 32, 0.99f, st));
        }
        CHECK_DN(dn_pipe_flush(pipe, 32, 0.99f, st));
    }
    CHECK_HIP(hipMemcpyAsync(h.data(), out + (group > 0 ? line * (hops - 1) : 0), sizeof(float) * B * N, hipMemcpyDeviceToHost, st));
    CHECK_HIP(hipStreamSynchronize(st));
    double s2 = 0;
    for (float v : h) s2 += (double)v * v;
    printf("B=%d hops=%d out rms=%.6f\n", B, hops, sqrt(s2 / h.size()));
    f = fopen(argv[2], "wb");
    fwrite(h.data(), sizeof(float), h.size(), f);
    fclose(f);
    if (pipe) dn_pipe_destroy(pipe);
    dn_model_destroy(model);
    dn_dsp_destroy(plan);
    return 0;
}
