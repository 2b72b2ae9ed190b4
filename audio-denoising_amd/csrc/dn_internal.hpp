// dn_internal.hpp -- launcher prototypes shared between dn_api.hip and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dn_denoise.h"
#include "dn_plan.hpp"

// Wave-uniform read-only tables (conv weights) are read through the constant address space so that
// hipcc emits scalar loads (s_load_dwordx*, scalar cache) and the FMAs take the weight as an SGPR
// operand, instead of 64 lanes fetching the same dword through the vector memory path.
#ifndef DN_CONST_AS
#define DN_CONST_AS __attribute__((address_space(4)))
#endif

namespace dn {

typedef const DN_CONST_AS float* cfloat_ptr;

// accumulator fragment of v_mfma_f32_16x16x4_f32: lane l holds D[row = (l >> 4) * 4 + r][col = l & 15], r = 0..3
#ifndef DN_F32X4
#define DN_F32X4
typedef float f32x4 __attribute__((ext_vector_type(4)));
#endif

// Workgroup barrier that orders LDS traffic only: global loads issued before it stay in flight across it (a plain
// __syncthreads() drains vmcnt first).  For phases that hand over LDS data while prefetches for later phases are pending.
#ifndef DN_LDS_BARRIER
#define DN_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

// Every vector-memory operation of the calling wave has completed (loads returned, stores acknowledged).
#ifndef DN_WAIT_VMEM
#define DN_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

// Wave-wide max / sum, the result in every lane.  On the device: six DPP steps on the VALU (row_shr 1, 2, 4, 8, then the row
// broadcasts 15 and 31 leave the total in lane 63) -- __shfl_xor compiles to ds_bpermute, an LDS round trip of ~230 cycles a step.
// (DN_WAVE_REDUCE_SHFL selects the __shfl_xor form, for a build whose target has no DPP.)
#ifdef DN_WAVE_REDUCE_SHFL
__device__ __forceinline__ float wave_max(float v) {
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float quad_sum(float v) {       // the sum of each aligned group of four lanes, in all four
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    return v;
}
#else
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_take(float v) {       // lanes without a source keep their own value
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_take<0x111, 0xf>(v));     // row_shr:1
    v = fmaxf(v, dpp_take<0x112, 0xf>(v));     // row_shr:2
    v = fmaxf(v, dpp_take<0x114, 0xf>(v));     // row_shr:4
    v = fmaxf(v, dpp_take<0x118, 0xf>(v));     // row_shr:8   -> lane 15 of every row of 16 holds the row's max
    v = fmaxf(v, dpp_take<0x142, 0xa>(v));     // row_bcast:15 into rows 1 and 3
    v = fmaxf(v, dpp_take<0x143, 0xc>(v));     // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's max
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// (the shifted-in operand must be 0 where a lane has no source: `old` = 0 with the lane's own value added outside)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_take0(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_take0<0x111, 0xf>(v);
    v += dpp_take0<0x112, 0xf>(v);
    v += dpp_take0<0x114, 0xf>(v);
    v += dpp_take0<0x118, 0xf>(v);
    v += dpp_take0<0x142, 0xa>(v);
    v += dpp_take0<0x143, 0xc>(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float quad_sum(float v) {       // the sum of each aligned group of four lanes, in all four
    v += dpp_take0<0xB1, 0xf>(v);     // quad_perm [1,0,3,2]
    v += dpp_take0<0x4E, 0xf>(v);     // quad_perm [2,3,0,1]
    return v;
}
#endif

// Elementwise transcendentals of the front half on the hardware units (v_exp_f32 / v_log_f32 / v_sqrt_f32 / v_rcp_f32, 1 ulp each)
// instead of libm's correctly rounded sequences (30-130 instructions a call).  The front workgroup shares its SIMDs with a Griffin-Lim
// chain that wins issue arbitration, so it pays ~11 cycles per instruction it issues: these calls were ~600 of its instructions.
// Absolute errors ~1e-7 relative to values of order 1-10 (log-mel features, linear magnitudes), against a parity tolerance of 1e-4.
__device__ __forceinline__ float fast_log1p(float x) { return __builtin_amdgcn_logf(1.0f + x) * 0.693147180559945309f; }   // x >= 0 here
__device__ __forceinline__ float fast_expm1(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f) - 1.0f; }
__device__ __forceinline__ float fast_abs2(float re, float im) { return __builtin_amdgcn_sqrtf(fmaf(re, re, im * im)); }

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

constexpr int kHidden = 17;      // H
constexpr int kGates = 51;       // 3H
constexpr int kGauss = 6;        // G
constexpr int kCellChunk = 3;    // time steps whose encoder/decoder run batched (= columns per hop)
constexpr int kMaxC = 5;         // compressed bins supported by the cell kernel (51*C gate lanes <= 256; F <= 80)
// most steps of the packed mel schedule (DspDev::mel_q) a wavefront unrolls: 19 are needed for 80 mels at 16 kHz / n_fft 1024, 39 for 64 mels at
// 48 kHz / n_fft 1536 (measured: a 48-step unroll costs the n_fft-1024 hop 1 %, and gives the 1536 one 0.8 %)
constexpr int mel_q_steps(int n_fft) { return n_fft == 1536 ? 48 : 32; }
constexpr int kInvBand = 16;      // diagonals of (fb^T fb)^-1 kept on either side of the main one (DspDev::ginv_band)
constexpr int kSlotMeta = 16;     // u32 of per-stream hand-over data in a pipe's scratch slot (dn_hop.hip: SlotLayout)
constexpr int kGlwAutoStreams = 768;
constexpr long kSplitAutoChains = 4096;         // DN_SPLIT_AUTO: chain wavefronts per launch from which a hop goes out as two launches (measured: 2,048 -2.4 %, 4,096 +2.1 %, 8,192 +5.1 %)   // DN_GL_AUTO: from this many streams per pipe on (three per CU of an MI355X; measured crossover between 512 and 768) the back half runs a wavefront per stream
constexpr int kArenaSlack = 8192; // zero bytes behind every device arena: kernels that move whole rounds of an array read past its end (dn_cell_body.hpp)

void launch_stft(const DspDev& d, const float* frames, float* spec, float* mel, float* peak, int B, uint32_t flags,
                 hipStream_t st);
void launch_mel(const DspDev& d, const float* mag, float* mel, int rows, hipStream_t st);
void launch_invmel(const DspDev& d, const float* x, const float* diff, float* lin, int rows, hipStream_t st);
void launch_griffinlim(const DspDev& d, const float* mag, const float* init, uint64_t seed, uint64_t sid0,
                       const float* scale, float* wave, int B, int n_iter, float momentum, hipStream_t st);
void launch_draw_phases(const DspDev& d, float* out, uint64_t seed, uint64_t sid0, int B, hipStream_t st);
void launch_synthesis(const DspDev& d, const float* x, const float* diff, const float* init, uint64_t seed, uint64_t sid0,
                      const float* scale, float* wave, int B, int n_iter, float momentum, hipStream_t st);
void launch_cell(const CellDev& c, const float* x, const float* hx_in, float* out, float* hx_out, int B, int T,
                 int C, hipStream_t st);
// Device-resident control block of a dn_pipe.  Every launch of hop_kernel derives its scratch slot, the Griffin-Lim seed
// and whether a hop is pending from it, and its last workgroup advances it -- so a launch captured in a hipGraph can
// be replayed indefinitely (nothing that changes from hop to hop is baked into the kernel arguments).
struct PipeCtl {
    unsigned long long pushes;   // launches that carried a front half (streaming: the first n_fft/hop - 1 only fill the ring)
    unsigned long long frames;   // frames whose front half (P1-P10) has run
    unsigned long long launches; // launches of the pipe so far (submits, pushes and flushes)
    unsigned int pending;        // frames whose Griffin-Lim has not finished: the most recent `pending` ones (0/1; up to the pipe's depth)
    unsigned int done;           // workgroup ticket of the launch in flight (0 between launches)
    unsigned int slot_next;      // scratch slot the next front half writes (slots are used round robin)
    unsigned int pad_;
    unsigned long long front_launch[8];   // launch index in which frame f's front half ran, at [f & 7] (its chain segment s runs in launch + 1 + s)
};

// Arguments of the software-pipelined hop launch (dn_hop.hip).
struct HopArgs {
    PipeCtl* ctl;
    // scratch slots (n_slots of them, used round robin; depth + 1), each: mel [B][3][M] | residual [B][3][M] | peak [B] | meta [B][kSlotMeta] (u32, see
    // SlotLayout) | lin [B][3][K]; per slot also the frame's initial phases [B][3][K] complex (parity mode; or null) and its parked
    // Griffin-Lim chain (head start, chain segments; or null)
    float* slots; size_t slot_stride;
    float2* slot_init; size_t init_stride;
    float2* gl_state; size_t state_stride;
    int n_slots;
    // front half: this hop's analysis + model + inverse mel, written to slot ctl->slot_next
    const float* frames; float* hx;
    const float* init_in;    // this hop's initial phases [B][3][K] complex (copied into the slot), or null = device RNG
    uint64_t seed, sid0;     // the frame's Griffin-Lim draws from (seed + frame index, sid0 + stream)
    // these three describe THIS hop's frame too: the back half finishes a pending hop with the n_iter, momentum and destination its own front
    // workgroup left in the slot
    float* gl_out;
    int n_iter; float mom;
    // head start: the front workgroup, done with P1-P10 long before the launch ends, runs the first `gl_split` Griffin-Lim iterations of
    // ITS frame and parks the chain in the slot's gl_state; the back workgroup of the next launch resumes there (0 = no head start)
    int gl_split;
    int front_B, back_B, B, C;
    // back half: back_blocks workgroups.  A wavefront per column: back_B of them, one pending hop.  A wavefront per stream (dn_glw_body.hpp): each
    // holds spb streams x depth chain segments (spb x depth <= 4), back_blocks = ceil(back_B / spb)
    int back_blocks, spb, depth, glw;
    // split: the hop goes out as two launches (launch_hop does that: chains, then front halves); front_only marks the second of them
    int split, front_only;
    // streaming mode (pipe-owned per-stream state): the front half first shifts `hop_in` into `ring` and uses the ring
    // as its frame (app3.py:178,226); the back half folds its frame into `ola` and emits `hop_out` (app3.py:219-224)
    const void* hop_in; float* ring; int in_s16; int prime;
    float* ola; void* hop_out; int out_s16;
    // zero-copy host transport: hop_in / hop_out are page-locked HOST memory; the last workgroup of the launch, after every workgroup's stores have
    // been fenced to system scope, publishes host_done_value in *host_done (page-locked too) -- the host polls that word instead of a HIP event
    unsigned long long* host_done; unsigned long long host_done_value;
    // deferred host output (DN_HOST_DEFER): the hop the PREVIOUS push emitted sits in a device staging buffer; the front workgroup of stream b
    // moves that stream's row (host_copy_n16 / B 16-byte units) to that push's page-locked buffer before it starts on its own hop, so the PCIe
    // writes overlap the hop instead of ending it as one burst (null = nothing to move; only launches with front workgroups carry one)
    const uint4* host_copy_src; uint4* host_copy_dst; unsigned int host_copy_n16;
    // hop groups (group_kernel; dn_pipe_set_group): ONE launch carries `group_hops` consecutive new hops of every stream (their front halves run one
    // after the other in the stream's front workgroup, hx carried) and the WHOLE Griffin-Lim chains of the hops the previous launch fronted, one
    // wavefront each -- no chain is ever parked.  Hop h of the group reads frames + h * frames_stride (hop_in + h * hop_in_stride when streaming),
    // takes its injected phases from init_in + h * init_in_stride and is written to gl_out + h * out_stride; a streaming launch emits `group_out`
    // hops at hop_out + i * hop_out_stride (strides in elements of the respective type); filler_first: hops without a frame behind them (start of a
    // stream) are the leading ones of a push and the trailing ones of a flush
    int group_hops, group_out, filler_first, n_mels;
    const DspDev* d_dev; const CellDev* c_dev;          // device copies of the plan's and the model's views (group_kernel reads them where it needs them)
    long long frames_stride, out_stride, init_in_stride, hop_in_stride, hop_out_stride;
};
void launch_host_copy(const uint4* src, uint4* dst, unsigned int n16, unsigned long long* done, unsigned long long value, hipStream_t st);
void launch_hop(const DspDev& d, const CellDev& c, const HopArgs& a, bool bf16, hipStream_t st);
void launch_group(const DspDev& d, const CellDev& c, const HopArgs& a, bool bf16, hipStream_t st);
void launch_ctl_set(PipeCtl* ctl, unsigned long long pushes, unsigned long long frames, unsigned int pending, hipStream_t st);

// The whole hop for one batch in ONE launch, nothing overlapped (dn_process_frame / dn_stream_step).
struct FrameArgs {
    const float* frames; float* hx; float* out;
    float* mel; float* diff; float* peak;        // workspace (diff may be the caller's mel_residual_out)
    const float* init; uint64_t seed, sid0;
    int n_iter; float mom; int C;
    const float* hop_in; float* ring; float* ola; float* hop_out;    // dn_stream_step (ring != null)
};
void launch_frame(const DspDev& d, const CellDev& c, const FrameArgs& a, int B, bool bf16, hipStream_t st);
void launch_cell_bf16(const CellDev& c, const float* x, const float* hx_in, float* out, float* hx_out, int B, int T,
                      int C, hipStream_t st);
void launch_cell_ex(const CellDev& c, const float* x, const float* hx_in, float* out, float* hx_out, int B, int T,
                    int C, float hx_scale, hipStream_t st);
void launch_stft_general(const DspDev& d, const float* x, float* spec, float* logmel, int B, int L, hipStream_t st);
void launch_server_rows(const DspDev& d, const float* logmel, const float* model_out, const float* spec_in, float* spec_out, int rows,
                        hipStream_t st);
void launch_istft_general(const DspDev& d, const float* spec, float* wave, int B, int T, hipStream_t st);

}  // namespace dn
