/*
 * dn_denoise.h -- C ABI of the MI355X-native per-hop speech-denoising path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch types.
 * Every buffer argument marked [dev] is a DEVICE pointer owned by the caller
 * (e.g. torch.Tensor.data_ptr() of a contiguous fp32 tensor on the GPU); the
 * library never frees or retains it beyond the call.  Arguments marked
 * [host] are host pointers read during the call only.  All kernels are
 * enqueued on the HIP stream passed in (`stream`, a hipStream_t cast to
 * void*; NULL = the null stream); no call synchronises the device.
 * Handles are immutable after creation and may be shared by threads.
 *
 * Reference interfaces replaced (file:line in belacks/audio-denoising):
 *   dn_model_create / dn_cell_forward   gruunet2.py:246-306  GRUUNet2.__init__/forward
 *                                       (called at app3.py:112-116, 200-201; server.py:151,212)
 *   dn_momo_create / dn_momo_forward    momo3.py:247-324     MOMO3.__init__/forward (sibling model)
 *   dn_dsp_create                       app3.py:135-155  construction of the torchaudio
 *                                       Spectrogram/MelScale/InverseMelScale/GriffinLim + Hann window
 *   dn_stft                             app3.py:191      Spectrogram(power=None)(x)
 *   dn_stft_mel_log1p                   app3.py:179-195  peak-normalise, Hann, STFT, |.|, MelScale, log1p, transpose
 *   dn_mel_scale                        app3.py:193      MelScale(mag)
 *   dn_invmel / dn_residual_invmel      app3.py:203-211  leaky_relu(in-out), expm1, clamp, InverseMelScale, clamp
 *   dn_griffinlim                       app3.py:213-217  GriffinLim(power=1)(lin) (* peak)
 *   dn_synthesis                        app3.py:203-217  residual .. InverseMelScale .. GriffinLim (* peak), one launch
 *   dn_istft                            server.py:174,216 InverseSpectrogram
 *   dn_stft_general / dn_server_rows / dn_istft_general / dn_cell_forward_ex   server.py:199-217  the socket server's variant
 *   dn_process_frame                    app3.py:178-217  the whole per-hop loop body for B streams
 *   dn_stream_step                      app3.py:178-226  the same plus ring buffer / overlap-add state (P12)
 *   dn_pipe_*                           app3.py:167-250  the same hop, consecutive hops software-pipelined in one launch per hop
 *                                       (dn_pipe_stream_*: the steady-state recv loop with its per-stream buffers)
 *
 * Memory layouts (row-major, fp32; "complex" = interleaved re,im float pairs):
 *   frames      [B][n_fft]
 *   spec        [B][T][K] complex      K = n_fft/2+1, T = 3 columns per frame.  The reference's
 *                                      (B,K,T) tensors are the transpose(-1,-2) VIEW of this buffer.
 *   mel / x     [B][T][M]              exactly the (B,T,F) tensor GRUUNet2.forward takes
 *   hx          [B][17][C]             C = M/16
 *   mag         [B][T][K]
 *   wave        [B][n_fft]
 *
 * Return value: 0 on success, negative dn_status on failure; the message of
 * the last failure on the calling thread is returned by dn_last_error().
 */
#ifndef DN_DENOISE_H
#define DN_DENOISE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DN_ABI_VERSION 4

typedef enum dn_status {
    DN_OK = 0,
    DN_ERR_INVALID = -1,      /* bad argument / shape mismatch */
    DN_ERR_UNSUPPORTED = -2,  /* configuration outside what the kernels are built for */
    DN_ERR_HIP = -3,          /* HIP runtime failure (message has hipGetErrorString) */
    DN_ERR_NOMEM = -4
} dn_status;

typedef struct dn_model dn_model; /* GRUUNet2 weights, packed and resident on one device */
typedef struct dn_dsp dn_dsp;     /* STFT/mel/inverse-mel/Griffin-Lim plan, resident on one device */

/* Constructor arguments of gruunet2.GRUUNet2 (gruunet2.py:248-255).  The kernels are built for
 * the architecture of every GRUUNet2 checkpoint in the reference: 4 levels, hidden 17, k3 s2 p1,
 * 6 gaussians; anything else returns DN_ERR_UNSUPPORTED. */
typedef struct dn_model_cfg {
    int32_t num_compressed_bins; /* C; default hx is zeros(B,17,C) */
    int32_t in_size;             /* must be 1 (gruunet2.py:257) */
    int32_t n_levels;            /* 4 */
    int32_t hidden_size;         /* 17 (all levels equal) */
    int32_t kernel_size;         /* 3 */
    int32_t stride;              /* 2 */
    int32_t padding;             /* 1 */
    int32_t num_gaussians;       /* 6 */
} dn_model_cfg;

#define DN_MODEL_N_FLOATS 15337  /* state_dict order, SURVEY.md Appendix A.4 */

/* weights [host]: the reference state_dict flattened in its own key order (15,337 fp32).
 * The handle lives on the current HIP device. */
int dn_model_create(const float* weights, size_t n_floats, const dn_model_cfg* cfg, dn_model** out);
void dn_model_destroy(dn_model* m);

/* GRUUNet2.forward for B independent streams, T sequential steps.
 *   x [dev][B][T][F], hx_in [dev][B][17][C] (NULL = zeros), out [dev][B][T][F], hx_out [dev][B][17][C].
 * F must equal 16*C (else DN_ERR_INVALID, mirroring the reference's shape error).  hx_in is not
 * modified (gruunet2.py:240 builds a new tensor); hx_out may alias hx_in. */
int dn_cell_forward(const dn_model* m, const float* x, const float* hx_in, float* out, float* hx_out,
                    int32_t B, int32_t T, int32_t F, int32_t C, void* stream);

/* dn_cell_forward with the returned state scaled: hx_out = hx' * hx_out_scale (the `hx = hx * 0.9` of server.py:214). */
int dn_cell_forward_ex(const dn_model* m, const float* x, const float* hx_in, float* out, float* hx_out,
                       int32_t B, int32_t T, int32_t F, int32_t C, float hx_out_scale, void* stream);

/* BASELINE config 3: the same forward with bf16 MFMA conv tiles (v_mfma_f32_16x16x32_bf16): encoder / decoder conv
 * inputs and weights are rounded to bf16 (round to nearest even), accumulation, biases, the recurrent gate conv, the
 * GRU math and the last decoder level stay fp32.  Not bit-compatible with the fp32 reference: tolerance is restated
 * (tests/test_gpu_parity.py: <= 1e-2 relative RMS, <= 5e-1 max-abs on the mel residual; measured 4e-3 / 0.07-0.31). */
int dn_cell_forward_bf16(const dn_model* m, const float* x, const float* hx_in, float* out, float* hx_out,
                         int32_t B, int32_t T, int32_t F, int32_t C, void* stream);

/* ---- Sibling model MOMO3 (momo3.py:247-324; checkpoint saves/MOMO3-4d4ea0) on the same fp32 MFMA conv tiles -------------
 * Differences from GRUUNet2: a second input channel, the frame delta x_t - prev (momo3.py:285-289); the position code
 * enters only at the encoder input and at the hidden-gate conv (momo3.py:138-145); 3 levels, hidden 16, per-level
 * paddings (1, 0, 1) -- 22 input bins compress to 3.  The kernels are built for that architecture (paddings 0 or 1, up to
 * 64 input bins); anything else returns DN_ERR_UNSUPPORTED. */
typedef struct dn_momo dn_momo;
typedef struct dn_momo_cfg {
    int32_t num_compressed_bins; /* default hx is zeros(B,16,num_compressed_bins) */
    int32_t in_size;             /* 1 (the delta channel is added internally, momo3.py:260) */
    int32_t n_levels;            /* 3 */
    int32_t hidden_size;         /* 16 */
    int32_t kernel_size;         /* 3 */
    int32_t stride;              /* 2 */
    int32_t paddings[3];         /* (1, 0, 1) in the checkpoint */
    int32_t num_gaussians;       /* 6 */
} dn_momo_cfg;
#define DN_MOMO3_N_FLOATS 9197   /* state_dict order of momo3.MOMO3 */
int dn_momo_create(const float* weights, size_t n_floats, const dn_momo_cfg* cfg, dn_momo** out);
void dn_momo_destroy(dn_momo* m);
/* MOMO3.forward for B streams, T sequential steps: x [dev][B][T][F], hx_in [dev][B][16][C] (NULL = zeros), prev_in [dev][B][F]
 * (the frame before x[:,0]; NULL = the first delta is zero, momo3.py:277-278), out [dev][B][T][F], hx_out [dev][B][16][C],
 * prev_out [dev][B][F] or NULL (the last frame: what the caller passes as prev_in to continue the sequence).  C must be what F
 * compresses to under the model's paddings (else DN_ERR_INVALID, mirroring the reference's broadcast error). */
int dn_momo_forward(const dn_momo* m, const float* x, const float* hx_in, const float* prev_in, float* out, float* hx_out,
                    float* prev_out, int32_t B, int32_t T, int32_t F, int32_t C, void* stream);

typedef struct dn_dsp_cfg {
    int32_t sample_rate;
    int32_t n_fft;   /* 1024 (hop = n_fft/2, win_length = n_fft) */
    int32_t hop;
    int32_t n_mels;  /* 0 = no mel stages */
} dn_dsp_cfg;

/* fb [host][K][n_mels]: mel filterbank as MelScale.fb (NULL = HTK triangles computed natively, f_min 0,
 * f_max sr//2, norm None).  pinv [host][K][n_mels]: pseudo-inverse of fb^T used by the inverse-mel GEMM
 * (NULL = computed natively in double precision from fb; the plan then also keeps the operator in factors, fb and the
 * significant diagonals of (fb^T fb)^-1, and the kernels apply those when fb has at most two filters per bin -- the same
 * min-norm least-squares solution to fp32 rounding; an explicit pinv is always applied as the dense matrix).
 * window [host][n_fft] (NULL = periodic Hann). */
int dn_dsp_create(const dn_dsp_cfg* cfg, const float* fb, const float* pinv, const float* window, dn_dsp** out);
void dn_dsp_destroy(dn_dsp* d);
/* Copies of the plan's host-side tables, for inspection/tests: fb and pinv are [K][n_mels]. */
int dn_dsp_get_tables(const dn_dsp* d, float* fb, float* pinv, float* window);

/* flags for the analysis entry points */
#define DN_PEAK_NORMALIZE 1u /* P1: x /= max|x| when that is > 1e-6 (app3.py:181-186) */
#define DN_PRE_WINDOW 2u     /* P2: multiply by the window before the STFT windows again (app3.py:188) */

/* Spectrogram(power=None): frames [dev][B][n_fft] -> spec [dev][B][3][K] complex. */
int dn_stft(const dn_dsp* d, const float* frames, float* spec, int32_t B, uint32_t flags, void* stream);

/* P1,P2,P4,P5,P6 fused: frames -> mel [dev][B][3][M] = log1p(MelScale(|STFT|)), peak [dev][B]
 * (peak may be NULL; it is 1 where normalisation was skipped). */
int dn_stft_mel_log1p(const dn_dsp* d, const float* frames, float* mel, float* peak, int32_t B,
                      uint32_t flags, void* stream);

/* MelScale alone: mag [dev][B][T][K] -> mel [dev][B][T][M] (no log). */
int dn_mel_scale(const dn_dsp* d, const float* mag, float* mel, int32_t B, int32_t T, void* stream);

/* InverseMelScale: mel_mag [dev][B][T][M] -> lin [dev][B][T][K] = relu(pinv(fb^T) @ mel). */
int dn_invmel(const dn_dsp* d, const float* mel_mag, float* lin, int32_t B, int32_t T, void* stream);

/* P8,P9,P10 fused: lin = relu(pinv @ clamp(expm1(leaky_relu(x - diff, 0.2)), 0)).
 * x, diff [dev][B][T][M]; lin [dev][B][T][K]. */
int dn_residual_invmel(const dn_dsp* d, const float* x, const float* diff, float* lin, int32_t B, int32_t T,
                       void* stream);

/* GriffinLim(power=1, n_iter, momentum, length=None) on 3-column magnitudes.
 *   mag [dev][B][3][K]; init_angles [dev][B][3][K] complex or NULL; when NULL the initial phases are drawn
 *   on the device (real, imag ~ U[0,1) independently, as the reference's torch.rand(complex64)) from a
 *   counter-based generator keyed by (seed, stream_id0 + b, element), so results do not depend on how
 *   streams are sharded.  scale [dev][B] or NULL multiplies the output (the `* peak` of app3.py:217).
 *   wave [dev][B][n_fft]. */
int dn_griffinlim(const dn_dsp* d, const float* mag, const float* init_angles, uint64_t seed,
                  uint64_t stream_id0, const float* scale, float* wave, int32_t B, int32_t n_iter,
                  float momentum, void* stream);

/* The initial phases dn_griffinlim / dn_synthesis / the fused hops draw when init_angles is NULL, for streams stream_id0 .. stream_id0 + B - 1:
 * angles_out [dev][B][3][K] complex, real and imaginary part ~ U[0,1) independently (torchaudio's GriffinLim(rand_init=True) =
 * torch.rand(complex64), app3.py:149-153).  Generator: Philox4x32-10 (Salmon et al., SC'11), one block per bin PAIR (m, K-1-m), m = 0..(K-1)/2:
 * counter = (m, column, stream id lo, hi), key = (seed lo, hi); words 0, 1 of the block are bin m, words 2, 3 bin K-1-m (the self-paired
 * middle bin takes words 0, 1), top 24 bits each -- so the draw does not depend on how streams are batched or sharded.
 * Feeding the result back as init_angles reproduces the NULL launch bit for bit (and lets a CPU reference run with the same phases). */
int dn_griffinlim_draw_phases(const dn_dsp* d, uint64_t seed, uint64_t stream_id0, float* angles_out, int32_t B, void* stream);

/* P8..P12 in ONE launch (app3.py:203-217): x, diff [dev][B][3][M] (model input and model output) ->
 * leaky_relu(x - diff, 0.2), expm1, clamp, inverse mel, relu, Griffin-Lim, * scale.  The linear magnitudes
 * stay in LDS.  init_angles / seed / stream_id0 / scale as for dn_griffinlim. */
int dn_synthesis(const dn_dsp* d, const float* x, const float* diff, const float* init_angles, uint64_t seed,
                 uint64_t stream_id0, const float* scale, float* wave, int32_t B, int32_t n_iter, float momentum,
                 void* stream);

/* torch.istft(center=True, length=None) of 3 columns: spec [dev][B][3][K] complex -> wave [dev][B][n_fft]. */
int dn_istft(const dn_dsp* d, const float* spec, float* wave, int32_t B, void* stream);

/* ---- Arbitrary-length chunks: the request loop of the reference's socket server (server.py:199-217) ----
 * dn_stft_general: Spectrogram(power=None) of x [dev][B][L] (any L > n_fft/2) -> spec [dev][B][T][K] complex (may be NULL),
 *   and/or logmel [dev][B][T][M] = log1p(MelScale(|spec|)) (may be NULL); T = 1 + L / hop.          server.py:207-210
 * dn_server_rows: per (b,t) row: relu(model_out) * 3, exp(logmel - .) - 1, InverseMelScale, torch.polar(., angle(spec_in))
 *   -> spec_out [dev][rows][K] complex (may alias spec_in).                                         server.py:213-216
 * dn_istft_general: InverseSpectrogram (length=None), spec [dev][B][T][K], T >= 2 -> wave [dev][B][hop*(T-1)].  server.py:216 */
int dn_stft_general(const dn_dsp* d, const float* x, float* spec, float* logmel, int32_t B, int32_t L, void* stream);
int dn_server_rows(const dn_dsp* d, const float* logmel, const float* model_out, const float* spec_in, float* spec_out,
                   int32_t rows, void* stream);
int dn_istft_general(const dn_dsp* d, const float* spec, float* wave, int32_t B, int32_t T, void* stream);

/* flags of the fused entry points */
#define DN_CONV_BF16 1u /* BASELINE config 3: encoder/decoder convs on bf16 MFMA tiles (as dn_cell_forward_bf16); 0 = exact fp32 */

/* Scratch the fused entry points need, in bytes, for a batch of B streams. */
size_t dn_workspace_bytes(const dn_dsp* d, int32_t B);

/* The whole per-hop body for B streams (app3.py:178-217): frames [dev][B][n_fft] raw samples,
 * hx [dev][B][17][C] in/out, out [dev][B][n_fft] = GriffinLim(...) * peak.
 * mel_residual_out [dev][B][3][M] or NULL receives the model output (predicted_diff_mel).
 * workspace [dev] of dn_workspace_bytes(d, B) bytes.  ONE launch (one workgroup per stream runs P1-P12 back to back);
 * no hop of added latency.  flags: DN_CONV_BF16 or 0. */
int dn_process_frame(const dn_model* m, const dn_dsp* d, const float* frames, float* hx, float* out,
                     float* mel_residual_out, const float* init_angles, uint64_t seed, uint64_t stream_id0,
                     int32_t n_iter, float momentum, void* workspace, int32_t B, uint32_t flags, void* stream);

/* Streaming step (app3.py:178-226): ring [dev][B][n_fft] holds the last n_fft input samples per stream;
 * hop_in [dev][B][hop] new samples are shifted in first.  ola [dev][B][n_fft] is the output
 * overlap-add buffer; hop_out [dev][B][hop] receives ola[:hop] before the shift (app3.py:219-224).
 * ONE launch; every argument is validated before it, so a failed call leaves ring, ola and hx untouched. */
int dn_stream_step(const dn_model* m, const dn_dsp* d, const float* hop_in, float* ring, float* ola, float* hx,
                   float* hop_out, const float* init_angles, uint64_t seed, uint64_t stream_id0, int32_t n_iter,
                   float momentum, void* workspace, int32_t B, uint32_t flags, void* stream);

/* ---- Software-pipelined hops ------------------------------------------------------------------------
 * Consecutive hops depend on each other only through hx (the model); hop n's Griffin-Lim (~3/4 of a hop) is
 * independent of hop n+1's analysis + model + inverse mel.  A dn_pipe overlaps them inside ONE launch per hop:
 * dn_pipe_submit(hop n+1) launches a grid whose first B workgroups run hop n's Griffin-Lim and whose next B run
 * hop n+1's P1-P10 (double-buffered scratch); dn_pipe_flush launches the last pending Griffin-Lim.  Everything is
 * enqueued on `stream`; the output of a submitted hop is complete (in stream order) after the NEXT submit or the
 * flush.  `frames`, `hx` and `out` of a submit must stay valid and untouched until then (hx is advanced in place, hop by
 * hop); `init_angles` is copied during the submit and is free again once that launch ran.
 *
 * Replayable launches: everything that changes from hop to hop (scratch-slot parity, whether a hop is pending, the ring
 * priming of new streams, the frame index that keys the Griffin-Lim seed) lives in a device-resident control block that the
 * last workgroup of every launch advances.  A dn_pipe_submit / dn_pipe_stream_push captured into a hipGraph can therefore be
 * replayed indefinitely (BASELINE config 5: the captured streaming step); the caller rewrites the contents of the buffers
 * whose pointers were captured.  The Griffin-Lim of frame f draws its phases from (seed + f, stream_id0 + stream), f counted
 * per pipe from 0 (dn_pipe_stream_set_state can restore it), so a replayed launch does not repeat phases and a resumed
 * stream continues the sequence of the uninterrupted one.
 *
 * A pipe holds a reference on its model and plan: dn_model_destroy / dn_dsp_destroy while a pipe still uses them only drop
 * the creator's reference.  dn_pipe_set_model rebinds a pipe to other weights between two launches (launches already
 * enqueued keep reading the old model: keep it alive until they have run).  flags: DN_CONV_BF16 or 0. */
typedef struct dn_pipe dn_pipe;
int dn_pipe_create(const dn_model* m, const dn_dsp* d, int32_t B, uint32_t flags, dn_pipe** out);
void dn_pipe_destroy(dn_pipe* p);
int dn_pipe_set_model(dn_pipe* p, const dn_model* m);
/* Head start: a front workgroup is done with P1-P10 well before the pending hop's Griffin-Lim chain (same launch) is; with
 * `iterations` > 0 it goes on with the first iterations of ITS frame's chain and parks it in HBM, and the next launch resumes there.
 * Same results bit for bit, no added latency; pays when there is about one stream per CU.  dn_pipe_create turns it on by itself up to
 * 256 streams (8 iterations at n_fft 1024, 12 at 1536: the measured optima).  Call between launches (0 = off). */
int dn_pipe_set_head_start(dn_pipe* p, int32_t iterations);
/* Depth of the pipe = hops of ONE stream in flight (n_fft 1024; DN_ERR_UNSUPPORTED at 1536, where the per-lane state of a stream does not fit one wavefront).  1 (default): hop n's Griffin-Lim runs beside hop n+1's front half, as
 * described above.  D > 1: a frame's Griffin-Lim chain is cut into D segments that run in the D launches after its submit -- one wavefront
 * per stream and segment (DN_GL_WAVE_PER_STREAM), the chain parked in HBM between launches at the top of an iteration, so the result is
 * bit-identical to depth 1 -- and a launch carries the segments of D different hops of every stream.  With about one stream per CU (the
 * batch-256 metric) that is what fills the CU: the throughput of the saturated regime for D - 1 more hops of latency.  The output of a submit
 * is then complete after the D-th launch that follows it (dn_pipe_flush runs as many as are needed); a stream push emits the samples of the
 * frame delivered D pushes earlier, and dn_pipe_stream_flush must be called D times to drain (each call emits one hop).
 * Call while nothing is in flight (after create or flush); synchronises the device. */
#define DN_PIPE_MAX_DEPTH 4
int dn_pipe_set_depth(dn_pipe* p, int32_t depth);
/* Hop groups (n_fft 1024; DN_ERR_UNSUPPORTED at 1536, where the form measured slower than the one-hop pipe): the loop body of app3.py:178-226 for `hops` CONSECUTIVE hops of every stream in ONE launch.
 * A deep pipe (above) pays for every launch boundary inside a Griffin-Lim chain: the chain parks in HBM and comes back at the head of the next
 * launch.  With dn_pipe_set_group(p, H), 1 <= H <= DN_PIPE_MAX_GROUP, a launch is as long as a chain instead: dn_pipe_submit_group carries up to H
 * new hops of every stream -- hop h of the group reads frames + h * frames_stride, its front half (P1-P10) runs behind hop h-1's with hx handed on,
 * its result goes to out + h * out_stride, its injected phases (parity mode) come from init_angles + h * init_stride (strides in floats) -- beside
 * the WHOLE chains of the hops the previous launch fronted, one wavefront each.  Nothing is parked between launches; frames, hx and samples are
 * those of the one-hop pipe bit for bit (same seeds: the f-th frame of the pipe draws from seed + f).  The output of a group is complete after
 * the next dn_pipe_submit_group or one dn_pipe_flush.  Throughput of the deep pipe without its hand-off, for input that arrives H hops at a time.
 * Streaming form: dn_pipe_stream_push_group takes exactly H hops (hop_in + h * in_stride, elements of the input type) and emits H hops
 * (hop_out + i * out_stride); the emitted stream is the one-hop pipe's delayed by H - 1 more hops (zeros until then), and
 * dn_pipe_stream_flush_group emits the frames still pending first and zero hops behind them (always H hops; *hops_valid, may be NULL, says how
 * many carry samples as far as this host thread's own pushes tell -- after graph replays ask dn_pipe_get_counters for `pending` before the flush).
 * With H < 4 a workgroup's four wavefronts serve 4 / H streams (H = 2: two streams a workgroup -- what 512 to 768 streams per GPU want).
 * dn_pipe_set_group: call while nothing is in flight; H = 0 returns the pipe to single hops; synchronises the device; excludes depth > 1, a head
 * start and the host-buffer transport.  dn_pipe_submit on a group pipe is a group of one hop. */
#define DN_PIPE_MAX_GROUP 4
int dn_pipe_set_group(dn_pipe* p, int32_t hops);
int dn_pipe_submit_group(dn_pipe* p, const float* frames, int64_t frames_stride, float* hx, float* out, int64_t out_stride,
                         const float* init_angles, int64_t init_stride, uint64_t seed, uint64_t stream_id0, int32_t hops,
                         int32_t n_iter, float momentum, void* stream);
int dn_pipe_stream_push_group(dn_pipe* p, const void* hop_in, int64_t in_stride, int32_t in_is_s16, void* hop_out, int64_t out_stride,
                              int32_t out_is_s16, const float* init_angles, int64_t init_stride, uint64_t seed, uint64_t stream_id0,
                              int32_t n_iter, float momentum, void* stream);
int dn_pipe_stream_flush_group(dn_pipe* p, void* hop_out, int64_t out_stride, int32_t out_is_s16, int32_t* hops_valid, void* stream);
/* How the pending hop's Griffin-Lim is laid out on the GPU (n_fft 1024; results are bit-identical either way):
 *   DN_GL_WAVE_PER_COLUMN  three wavefronts per stream, one per STFT column: the shortest chain for one stream -- right when there is
 *                          about one stream per CU (the batch-256 metric);
 *   DN_GL_WAVE_PER_STREAM  one wavefront per stream, the three columns interleaved inside it, four streams per workgroup: no workgroup
 *                          barrier, the overlap-add in registers, four times the streams in flight -- right for several streams per CU;
 *   DN_GL_AUTO (default)   per stream from 768 streams per pipe on (three per CU of an MI355X: the measured crossover), per column below.
 * Call between launches. */
#define DN_GL_AUTO 0
#define DN_GL_WAVE_PER_COLUMN 1
#define DN_GL_WAVE_PER_STREAM 2
int dn_pipe_set_gl_schedule(dn_pipe* p, int32_t schedule);
/* A hop as TWO launches instead of one: first the Griffin-Lim chains of the hops in flight (what a flush launch is), then the new hop's front
 * halves (P1-P10) as a launch of their own.  Where a launch carries several times the workgroups the GPU holds at once its chains and its front
 * halves run as two phases anyway, and a front workgroup compiled into the same kernel as a chain inherits the chain's register budget (two
 * workgroups per CU); on their own four fit.  Only with the wavefront-per-stream schedule (n_fft 1024) and no head start; same control block,
 * same samples, still capturable (two kernel nodes).  DN_SPLIT_AUTO (default) decides from the number of chain wavefronts per launch. */
#define DN_SPLIT_AUTO (-1)
#define DN_SPLIT_OFF 0
#define DN_SPLIT_ON 1
int dn_pipe_set_split(dn_pipe* p, int32_t mode);
/* Allocates the per-slot buffers for injected initial phases now (otherwise the first submit/push with init_angles does it):
 * call before capturing a parity-mode launch into a hipGraph, where allocation is not allowed. */
int dn_pipe_reserve_parity(dn_pipe* p);
int dn_pipe_submit(dn_pipe* p, const float* frames, float* hx, float* out, const float* init_angles, uint64_t seed,
                   uint64_t stream_id0, int32_t n_iter, float momentum, void* stream);
/* Runs the pending hop's Griffin-Lim; a no-op launch when nothing is pending.  A pending hop is always finished with the n_iter,
 * momentum and `out` of ITS OWN submit (they travel with the frame in its scratch slot): the n_iter / momentum arguments of a flush
 * are validated and otherwise unused, a later submit with other values does not change a hop already submitted, and a captured
 * submit may be replayed between eager submits to other `out` buffers. */
int dn_pipe_flush(dn_pipe* p, int32_t n_iter, float momentum, void* stream);
/* Host copy of the control block (pushes, frames, pending; any may be NULL).  Synchronises `stream`. */
int dn_pipe_get_counters(dn_pipe* p, uint64_t* pushes, uint64_t* frames, int32_t* pending, void* stream);

/* Streaming form (BASELINE config 5; app3.py:168-250 without the av container): the pipe owns the per-stream state
 * (input ring, output overlap-add line, hx -- app3.py:130-133) in HBM and every push is ONE launch.
 *   hop_in  [dev][B][hop]  float32 samples, or int16 PCM when in_is_s16 (converted as app3.py:172: x / 32767)
 *   hop_out [dev][B][hop]  float32, or int16 when out_is_s16 (clip to [-1,1], * 32767, truncate: app3.py:244-245)
 * The first n_fft/hop - 1 pushes only fill the ring.  Because hops are software-pipelined, the samples the reference
 * would emit while processing frame f (ola[:hop] before frame f is added, app3.py:219-220) come out of the push
 * that follows the one that delivered frame f's last hop -- one hop of extra latency; zeros until then.
 * dn_pipe_stream_flush emits the last pending hop (zeros when none is pending); as in frame mode a pending hop is finished with
 * the n_iter / momentum of the push that delivered it. */
int dn_pipe_stream_create(const dn_model* m, const dn_dsp* d, int32_t B, uint32_t flags, dn_pipe** out);
int dn_pipe_stream_push(dn_pipe* p, const void* hop_in, int32_t in_is_s16, void* hop_out, int32_t out_is_s16,
                        const float* init_angles, uint64_t seed, uint64_t stream_id0, int32_t n_iter, float momentum,
                        void* stream);
int dn_pipe_stream_flush(dn_pipe* p, void* hop_out, int32_t out_is_s16, int32_t n_iter, float momentum, void* stream);
/* Host-buffer transport of the streaming form: the reference hands every hop across the host/device boundary (frames arrive as int16 arrays on
 * the host, app3.py:168-172; x.to(device) / y.cpu() around the hop, app3.py:189,215; int16 out, app3.py:244-250).  hop_in_host / hop_out_host
 * [host][B][hop] (int16 or float32).  The call enqueues on `stream` and returns without waiting.
 *   Page-locked buffers (hipHostMalloc, hipHostRegister, torch pin_memory): ZERO COPY -- the launch reads each stream's hop (1 KB) straight from
 *   host memory and stores the emitted hop straight into it; no copy engine and no second queue are involved.
 *   Pageable buffers, or flags = DN_HOST_STAGED: an upload on a copy queue of the pipe, the hop on `stream` behind it, a download on a second copy
 *   queue; device staging is double-buffered, so the upload of hop i+1 and the download of hop i-1 run while hop i computes.
 * `*ticket` (may be NULL) names the push: dn_pipe_stream_host_wait(p, ticket) blocks the calling thread until that push's samples are in ITS
 * hop_out_host.  hop_in_host must stay unchanged, and hop_out_host untouched, until then (wait for push i before reusing the buffers of push i;
 * with four sets in rotation the host can stay two pushes ahead of the result it waits for, which keeps the GPU busy back to back).  Initial phases
 * come from the device generator; mix freely with dn_pipe_stream_push / _flush on the same `stream`.
 *   flags = DN_HOST_DEFER (zero copy only; ignored otherwise): the launch leaves its emitted hop in a device staging buffer and the NEXT push's
 *   launch moves it to hop_out_host, each front workgroup its stream's row before it starts on its own hop -- the PCIe writes then overlap a hop's
 *   arithmetic instead of ending the launch as one burst (1,024 streams: 2 MB in 14 us of a 170 us launch).  The samples of push i are in host
 *   memory once push i + 1 has run; dn_pipe_stream_host_wait on the NEWEST push enqueues the move itself (on the stream of that push), so no
 *   result is ever stranded.  Same samples, one push later: for a host that stays ahead of the results it waits for, not for the lowest latency.
 */
#define DN_HOST_STAGED 1u
#define DN_HOST_DEFER 2u
int dn_pipe_stream_push_host(dn_pipe* p, const void* hop_in_host, int32_t in_is_s16, void* hop_out_host, int32_t out_is_s16,
                             uint64_t seed, uint64_t stream_id0, int32_t n_iter, float momentum, uint32_t flags, void* stream,
                             uint64_t* ticket);
int dn_pipe_stream_host_wait(dn_pipe* p, uint64_t ticket);
/* Checkpoint / resume of live streams: copy the pipe-owned state out to / in from caller buffers [dev]
 * (ring [B][n_fft], ola [B][n_fft], hx [B][17][C]; any may be NULL), ordered on `stream`.  Take snapshots after
 * dn_pipe_stream_flush: set_state drops a pending hop.  A restored ring counts as primed; frames_done (the `frames`
 * counter of dn_pipe_get_counters at snapshot time) makes the resumed stream continue the seed sequence. */
int dn_pipe_stream_get_state(dn_pipe* p, float* ring, float* ola, float* hx, void* stream);
int dn_pipe_stream_set_state(dn_pipe* p, const float* ring, const float* ola, const float* hx, uint64_t frames_done,
                             void* stream);

const char* dn_last_error(void);
int dn_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DN_DENOISE_H */
