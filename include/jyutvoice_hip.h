/* libjyutvoice_hip.so -- C ABI of the MI355X-native JyutVoice synthesis hot path.
 *
 * Plain C: opaque context, raw pointers, explicit shapes, a HIP stream passed as void*.  No torch
 * types, no exceptions across the boundary: every call returns JV_OK (0) or an error code and
 * jv_last_error() returns the message.  All work is enqueued on the caller's stream; device
 * pointers are caller-owned unless stated.  The library owns only its context: packed weights and a
 * workspace sized at jv_create().
 *
 * Each entry point names the reference interface it stands in for (paths relative to the
 * indiejoseph/JyutVoice tree).  The precedent for a raw-pointer estimator seam in the reference is the
 * TensorRT path of ConditionalCFM.forward_estimator (jyutvoice/flow/flow_matching.py:267-297), which
 * hands data_ptr()s of x/mask/mu/t/spks/cond to an engine and reads the result from x.
 *
 * Tensor layouts at this boundary are the reference's own: channels-first fp32, e.g. mel [B,80,T].
 */
#ifndef JYUTVOICE_HIP_H
#define JYUTVOICE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JV_OK 0
#define JV_ERR_ARG 1    /* bad argument                                   */
#define JV_ERR_STATE 2  /* weights missing / not finalized                */
#define JV_ERR_HIP 3    /* a HIP runtime call failed                      */
#define JV_ERR_SHAPE 4  /* shape mismatch or capacity exceeded            */
#define JV_ERR_NAME 5   /* unknown tensor name                            */

#define JV_MODEL_TTS 0  /* JyutVoiceTTS state-dict: encoder.*, dp.*, decoder.estimator.*, spk_embed_affine_layer.* */
#define JV_MODEL_HIFT 1 /* HiFTGenerator state-dict                                                                */
#define JV_MODEL_PROMPT 2 /* FlowEncoder state-dict (infer.py:35-83: input_embedding.*, encoder.*, encoder_proj.*) plus the
                             host-computed "pos_enc.div_term" [256] (jyutvoice/transformer/embedding.py:239-242)     */

typedef struct jv_context jv_context;

/* ---- lifetime -------------------------------------------------------------------------------------
 * Replaces the object graph configs/base.yaml:26-110 builds (JyutVoiceTTS + HiFTGenerator).  Capacity:
 * at most max_batch utterances of max_frames mel frames / max_tokens text tokens per call. */
int jv_create(jv_context** out, int device, int max_batch, int max_frames, int max_tokens);
void jv_destroy(jv_context* ctx);
/* jv_reserve: change the capacities of a live context.  Only the workspace (activation buffers, masks, captured step
 * graphs) is re-created; loaded and finalized weights, the noise tensor and the mel filterbank stay where they are --
 * nothing is uploaded or packed again.  Waits for the device to drain first (it frees buffers queued work may use), so
 * pre-size once with the largest (batch, frames, tokens) a service expects rather than growing call by call.  The
 * reference has no counterpart: its nn.Modules allocate activations per call (infer.py:419-433). */
int jv_reserve(jv_context* ctx, int max_batch, int max_frames, int max_tokens);
/* jv_reserve is failure-atomic: when the new workspace does not fit, the previous capacities are restored and the error is
 * returned; only if that fails as well is the context marked unusable -- jv_usable() then returns 0 and every other entry
 * point JV_ERR_STATE until jv_destroy.  (1 otherwise; 0 for a null context.) */
int jv_usable(const jv_context* ctx);
/* message of the last failing call on this thread (valid until the next failure) */
const char* jv_last_error(void);

/* ---- weights ---------------------------------------------------------------------------------------
 * Replaces tts.load_state_dict(ckpt["state_dict"]) / hift.load_state_dict(torch.load(hift.pt))
 * (infer.py:343-351).  Names and shapes are the checkpoint's own keys; the registry is enumerable so
 * a loader can validate a checkpoint before touching the GPU. */
int jv_num_tensors(const jv_context* ctx);
const char* jv_tensor_name(const jv_context* ctx, int i);
int jv_tensor_model(const jv_context* ctx, int i);
int jv_tensor_ndim(const jv_context* ctx, int i);
int64_t jv_tensor_dim(const jv_context* ctx, int i, int d);
/* copy one fp32 tensor (host or device memory) into the context */
int jv_load_tensor(jv_context* ctx, const char* name, const float* data, const int64_t* shape, int ndim, int on_device,
                   void* stream);
/* the CFM's fixed noise tensor [1,80,15000] (CausalConditionalCFM.rand_noise, flow_matching.py:353-354: a plain
 * attribute, not a state-dict entry; the host regenerates it from torch seed 0) */
int jv_load_noise(jv_context* ctx, const float* data, int64_t numel, int on_device, void* stream);
/* all tensors of `model` present -> fold weight-norm, pack GEMM operands.  Errors name the missing key. */
int jv_finalize(jv_context* ctx, int model, void* stream);

/* ---- flow decoder ------------------------------------------------------------------------------------
 * jv_flow_estimator_step: CausalConditionalDecoder.forward (jyutvoice/flow/decoder.py:917-1018) through the
 * forward_estimator seam (flow_matching.py:267-297).  x, mu, cond, out: [B2,80,T]; t: [B2]; spks: [B2,80];
 * lens: [B2] int32 valid frames per row (NULL = all T), the key-padding mask of decoder.py:951-959.
 * out may alias x (the TRT seam writes its result into x). */
int jv_flow_estimator_step(jv_context* ctx, const float* x, const int32_t* lens, const float* mu, const float* t,
                           const float* spks, const float* cond, int B2, int T, float* out, void* stream);
/* jv_flow_estimator_masked: the same call with the six named tensors of the TensorRT seam exactly as
 * ConditionalCFM.forward_estimator binds them (flow_matching.py:270-290; profile shapes scripts/export_onnx.py:343-346):
 * mask is the reference's own float mask [B2,1,T] (1 = frame, 0 = padding) instead of lengths; the result may be written
 * into x (the seam passes x.data_ptr() as the output address).  jyutvoice_amd.flow.estimator.HipEstimator wraps it. */
int jv_flow_estimator_masked(jv_context* ctx, const float* x, const float* mask, const float* mu, const float* t,
                             const float* spks, const float* cond, int B2, int T, float* out, void* stream);
/* jv_flow_set_streaming: the `streaming=True` mode of CausalConditionalDecoder.forward (decoder.py:951-954, 976-979,
 * 999-1002 -> utils/mask.py:91-126,192-198): chunk-causal attention with static_chunk_size = chunk_frames (50 in
 * configs/base.yaml:98) and all left chunks; 0 restores full attention.  Applies to the following estimator / solver
 * calls on this context. */
int jv_flow_set_streaming(jv_context* ctx, int chunk_frames);
/* jv_flow_set_graph: jv_cfm_solve replays one Euler step (step scalars -> estimator input -> estimator -> CFG update,
 * flow_matching.py:230-265) as a captured hipGraph per (B, T, attention mode), on a private stream fenced against
 * `stream` with events; the step reads (t, dt) through a device-side counter so one graph serves every step.  Off by
 * default (on = 1 here, or JV_STEP_GRAPH=1 in the environment at jv_create, turns it on): on MI355X / ROCm 7.2 the replay
 * measured 0.4 % (B = 32, T = 300) to 5 % (B = 1, T = 128) slower than the eager launches, which already run ahead of
 * the GPU.  The in-library profiler (jv_profile_enable) forces the eager path because it brackets every
 * launch with events.  Results are bit-identical either way. */
int jv_flow_set_graph(jv_context* ctx, int on);
/* jv_flow_set_contraction: how fp32 contractions are carried out on the 16-bit matrix cores.  exact_range = 0
 * (default): fp16x3 -- each operand split into two fp16 planes after an exact power-of-two scaling (22 significant bits,
 * three MFMA products, fp32 accumulate) -- wherever an upper bound of the operand is known, so that fp16's range cannot be
 * exceeded: PROVEN from the weights at load time for the estimator's Linear layers and attention (transformer.py:355-443:
 * LayerNorm outputs, Linears of bounded inputs, attention outputs, GELU of a bounded Linear), MEASURED on the device by the
 * kernels that produce the operand for the estimator's convolutions (decoder.py:110-115, 767-788) and the vocoder
 * (generator.py:90-97, 396-432).  Every other contraction, and all of them with exact_range = 1 (or JV_EXACT_RANGE=1 in
 * the environment at jv_create), runs bf16x6 (three bf16 planes, 24 bits, six products), which takes any fp32 operand.
 * Both meet the same operator-level bound against fp64 (tests/test_gpu_ops.py).  Applies to this context's estimator,
 * solver and vocoder calls. */
int jv_flow_set_contraction(jv_context* ctx, int exact_range);
/* jv_flow_contraction_info: what the load-time range proofs of the last jv_finalize(JV_MODEL_TTS) concluded for the
 * estimator's 56 transformer blocks (transformer.py:355-443; registry.hip), so that a caller -- or a test with a hostile
 * checkpoint -- can see which layers took the fp16x3 engine and which stayed on bf16x6 because their bound was unusable
 * (non-finite or beyond 1e30).  out[0] = blocks, out[1] = blocks whose four linears AND attention all have a usable bound,
 * out[2] = linears (of 4 per block) with a usable bound, out[3] = blocks whose attention operands (q, k, v) have one.
 * n = number of int32 slots in out (>= 4). */
int jv_flow_contraction_info(const jv_context* ctx, int32_t* out, int n);
/* jv_cfm_solve: CausalConditionalCFM.forward + ConditionalCFM.solve_euler (flow_matching.py:356-401, 215-265):
 * fixed noise prefix * temperature, cosine schedule, n_timesteps Euler steps with CFG rate 0.7.
 * mu, cond, mel: [B,80,T]; spks: [B,80]; lens: [B] int32 or NULL.  t_span_host: optional n_timesteps+1 host floats
 * (the caller's own 1-cos(linspace*pi/2)); NULL = computed here.  B > 1 is the batched extension, defined as the
 * per-utterance loop of the batch-1-only reference. */
int jv_cfm_solve(jv_context* ctx, const float* mu, const int32_t* lens, const float* spks, const float* cond, int B, int T,
                 int n_timesteps, float temperature, const float* t_span_host, float* mel, void* stream);

/* ---- prompt (voice-cloning) branch ---------------------------------------------------------------------
 * jv_prompt_encoder_fwd: FlowEncoder.forward of the reference's infer.py:35-83 -- Embedding(clamp(token, 0)) * mask ->
 * UpsampleConformerEncoder(streaming=False) (jyutvoice/transformer/upsample_encoder.py:329-375) -> Linear(512, 80): the
 * `prompt_h` that JyutVoiceTTS.synthesise prepends to mu (jyutvoice_tts.py:213-225).
 * tokens: int64 [B,Tk] speech-token ids (< 6561), token_len: int64 [B]; prompt_h: [B, 2*Tk, 80] (batch, time, channel),
 * zero beyond 2*token_len[b].  B > 1 is the per-utterance loop of the B = 1 reference usage.  2*Tk <= max_frames. */
int jv_prompt_encoder_fwd(jv_context* ctx, const int64_t* tokens, const int64_t* token_len, int B, int Tk, float* prompt_h,
                          void* stream);

/* jv_load_mel_basis / jv_mel_spectrogram: the prompt-mel front-end, `extract_speech_feat` of infer.py:166-186 ->
 * `mel_spectrogram` of jyutvoice/utils/audio.py:18-63 (24 kHz, n_fft = win = 1920 periodic Hann, hop 480, reflect pad 720,
 * center=False, |.| with the 1e-9 floor, 80 mel bands, log(clamp(., 1e-5))).
 * basis: the [80][961] fp32 mel filterbank (librosa.filters.mel(sr=24000, n_fft=1920, n_mels=80, fmin=0, fmax=8000) in the
 * reference -- data to this library); wav: [B, n_samples] in [-1, 1], n_samples > 720; mel: [B, 80, T],
 * T = 1 + (n_samples - 480) / 480. */
int jv_load_mel_basis(jv_context* ctx, const float* basis, int64_t numel, int on_device, void* stream);
int jv_mel_spectrogram(jv_context* ctx, const float* wav, int B, int n_samples, float* mel, void* stream);

/* ---- text encoder + duration predictor + length regulation ---------------------------------------------
 * jv_encoder_fwd: spk_embed_affine_layer(normalize(spk)) + TextEncoder.forward + DurationPredictor.forward
 * (jyutvoice/models/jyutvoice_tts.py:175-182, text_encoder.py:406-451, duration_predictor.py:48-60).
 * ids: int64 [B,Tt] each; x_lengths: int64 [B]; spk: [B,192] (raw, un-normalised).
 * outputs: x [B,576,Tt], mu_x [B,80,Tt], logw [B,1,Tt], spks_proj [B,80]. */
int jv_encoder_fwd(jv_context* ctx, const int64_t* phone, const int64_t* lang, const int64_t* tone, const int64_t* word_pos,
                   const int64_t* syllable_pos, const int64_t* x_lengths, const float* spk, int B, int Tt, float* x,
                   float* mu_x, float* logw, float* spks_proj, void* stream);
/* jv_length_regulate: w_ceil = ceil(exp(logw)*mask)*length_scale, y_lengths, generate_path, mu_y = attn^T mu_x
 * (jyutvoice_tts.py:184-203, utils/model.py:29-46).  Two-phase because T_max is data dependent:
 *   phase 1 (attn == NULL): writes w_ceil [B,1,Tt] and y_lengths [B] int64 (device); the caller reads max(y_lengths)
 *   phase 2: writes attn [B,Tt,Ty] (dense 0/1 path) and mu_y [B,80,Ty] for the given Ty. */
int jv_length_regulate(jv_context* ctx, const float* logw, const int64_t* x_lengths, const float* mu_x, int B, int Tt,
                       float length_scale, float* w_ceil, int64_t* y_lengths, int Ty, float* attn, float* mu_y,
                       void* stream);

/* ---- HiFT vocoder -------------------------------------------------------------------------------------------
 * jv_hift_f0:     ConvRNNF0Predictor.forward (jyutvoice/hifigan/f0_predictor.py:52-55): mel [B,80,T] -> f0 [B,T]
 * jv_hift_source: f0_upsamp + SourceModuleHnNSF/SineGen (generator.py:459-461, 141-176, 220-236) with the random
 *                 draws supplied: phase [B,9] (harmonic 0 ignored, treated as 0), noise [B,9,480T] ~ N(0,1);
 *                 -> s [B,1,480T]
 * jv_hift_decode: HiFTGenerator.decode (generator.py:396-432): mel [B,80,T], s [B,1,480T] -> wav [B,480T]
 * lens: [B] int32 valid mel frames per utterance or NULL. */
int jv_hift_f0(jv_context* ctx, const float* mel, const int32_t* lens, int B, int T, float* f0, void* stream);
int jv_hift_source(jv_context* ctx, const float* f0, const float* phase, const float* noise, int B, int T, float* s,
                   void* stream);
/* jv_hift_source_seeded: the same with the noise drawn INSIDE the kernel -- generator.py:171 (`torch.randn_like`) draws
 * values no caller can depend on, and a [B,9,480T] tensor written by one kernel only to be read once by the next is 166 MB
 * per pass at the headline size.  Each sample's nine N(0,1) draws are Philox4x32-10 + Box-Muller of (seed, call, utterance,
 * sample): reproducible for a given (seed, call), independent across calls (the caller counts them).  Parity tests inject the
 * oracle's noise through jv_hift_source instead. */
int jv_hift_source_seeded(jv_context* ctx, const float* f0, const float* phase, uint64_t seed, uint32_t call, int B, int T, float* s,
                          void* stream);
int jv_hift_decode(jv_context* ctx, const float* mel, const float* s, const int32_t* lens, int B, int T, float* wav,
                   void* stream);

/* ---- operator-level entry points (used by the parity tests; same kernels the stages above launch) -------------
 * jv_op_conv_gemm: out[m,n] = act(sum_{j,ci} A[m + tap_row0 + j*dil, ci] * W[n, j*Cin + ci] + bias[n]) (+ res[m,n])
 *                  A [a_rows, Cin] rows, W [N, ntaps*Cin], optional LayerNorm over N (N == 256) before act.
 * jv_op_attention: softmax(q k^T / 8 over keys < lens[b]) v for qkv rows [G + b*S + t][1536], 8 heads x 64.
 * jv_op_layernorm: rows [rows, C].
 * jv_op_linear_h3: out = act(A W^T + bias) (+ res) through the fp16x3 main loop (jv_flow_set_contraction); A [rows, K],
 *                  W [N, K], a_bound = the caller's bound on |A| (the kernel's contract: |A| <= a_bound); presplit = 1:
 *                  A is first written as fp16 planes (what LayerNorm / attention / the GELU epilogue do in the estimator)
 *                  and both operands travel by LDS-DMA; 2: reuse the previous call's planes (timing only).
 * jv_op_attention_h3: jv_op_attention through the fp16x3 kernel, with the caller's bounds on |q|, |k|, |v|.
 * jv_op_conv_h3_measured: jv_op_conv_gemm through the fp16x3 main loop with a bound measured on the device: amax_in points
 *                  at a float >= max |A| (the amax_out of A's producer), a_extra bounds what the prologue adds; amax_out
 *                  (optional, zero it first) receives max |out|.  amax_in = NULL: bf16x6, tracking only. */
int jv_op_conv_gemm(const float* A, int64_t a_rows, int M, int Cin, int ntaps, int tap_row0, int dil, const float* W, int N,
                    const float* bias, int act, int prologue, const float* alpha, float slope, const float* ln_g,
                    const float* ln_b, float ln_eps, const uint8_t* rowmask, const float* res, float* out, void* stream);
int jv_op_attention(const float* qkv, const int32_t* lens, int B, int G, int S, int L, float* out, void* stream);
float jv_h3_scale_for_bound(float bound);   /* host only: the power of two chosen for a proven bound (0 = unusable) */
int jv_op_conv_h3_measured(const float* A, int64_t a_rows, int M, int Cin, int ntaps, int tap_row0, int dil, const float* W,
                           int N, const float* bias, int act, int prologue, const float* alpha, float slope,
                           const uint8_t* rowmask, const float* res, const float* amax_in, float a_extra, float* amax_out,
                           float* out, void* stream);
int jv_op_attention_h3(const float* qkv, const int32_t* lens, int B, int G, int S, int L, float q_bound, float k_bound,
                       float v_bound, float* out, void* stream);
int jv_op_linear_h3(const float* A, int64_t rows, int M, int K, const float* W, int N, const float* bias, int act,
                    const float* res, float a_bound, int presplit, float* out, void* stream);
/* jv_op_rowgemm: the row-owning fp16x3 GEMM the estimator's transformer linears run on at batch sizes that fill the chip
 * (transformer.py:355-443), with each epilogue: epi 0 plain -> out fp32 [M,N]; 1 exact GELU -> out2 = fp16 planes
 * [2][M][N] of value * out2_scale; 2 + res -> out; 3 + res -> out, then LayerNorm_256 -> out2 = planes [2][M][256].
 * a_bound: the caller's bound on |A|; presplit = 2 reuses the previous call's A planes (timing).  N % 256 == 0, K % 32 == 0. */
int jv_op_rowgemm(const float* A, int64_t rows, int M, int K, const float* W, int N, const float* bias, int epi,
                  const float* res, const float* ln_g, const float* ln_b, float a_bound, float out2_scale, int presplit,
                  float* out, uint16_t* out2, float* amax_out, void* stream);
/* jv_op_attention_planes: the estimator's attention kernel at chip-filling batch sizes (attention_pl.hip: K / V taken as
 * fp16 planes by LDS-DMA, transposed LDS reads for V) on an fp32 qkv matrix [rows,1536]; chunk > 0: chunk-causal mask;
 * out2 != NULL: result as fp16 planes [2][rows][512] of value * out2_scale, else fp32 rows in out [rows,512]. */
int jv_op_attention_planes(const float* qkv, int64_t rows, const int32_t* lens, int B, int G, int S, int L, float q_bound,
                           float k_bound, float v_bound, int chunk, float out2_scale, float* out, uint16_t* out2, void* stream);
/* jv_op_hiftconv: the vocoder's ResBlock convolution (hiftconv_kernel.h; jyutvoice/hifigan/generator.py:90-97):
 * out = ((Conv1d(C, C, ntaps, dilation dil, same padding)(Snake_alpha(A)) + bias) + res1 + res2) * out_scale (+ out when
 * accumulate) on a [rows, C] row buffer, C = 64 / 128 / 256, W [C][ntaps * C] tap-major, rows with rowmask == 0 read as zero;
 * amax_in: one device float >= max |A|, a_extra: max 1 / (alpha + 1e-9); amax_out (optional): max |out| is folded into it. */
int jv_op_hiftconv(const float* A, int64_t rows, int C, int ntaps, int dil, const float* W, const float* bias, const float* alpha,
                   const uint8_t* rowmask, const float* res1, const float* res2, float out_scale, int accumulate,
                   const float* amax_in, float a_extra, float* amax_out, float* out, void* stream);
/* jv_op_rowconv: the estimator's causal k = 3 convolution to 256 channels at chip-filling batch sizes (rowconv_kernel.h):
 * out[m] = tail(sum_j A[m - 2 + j] W_j + bias), tail = LayerNorm_256 (ln_g != NULL) -> act -> rows with rowmask == 0 := 0 ->
 * + rowvec (one [256] vector here) -> + res; A's fp16x3 scale comes from *amax_in (>= max |A|), amax_out receives max |out|
 * over the unmasked rows (decoder.py:110-115, 767-788). */
int jv_op_rowconv(const float* A, int64_t rows, int M, int Cin, const float* W, const float* bias, const float* ln_g,
                  const float* ln_b, int act, const uint8_t* rowmask, const float* rowvec, const float* res,
                  const float* amax_in, float* amax_out, float* out, void* stream);
int jv_op_layernorm(const float* x, const float* g, const float* b, float eps, int64_t rows, int C, float* out,
                    void* stream);

/* ---- measurement -------------------------------------------------------------------------------------------
 * HIP events on the launch stream around every conv_gemm / attention launch (replaces nothing in the reference,
 * whose only timing is the unsynchronised wall clock of jyutvoice_tts.py:171-172,243-244).
 * jv_profile_report synchronises the device and writes {"kernel":{"launches":n,"ms":t,"flops":f,"bytes":b},...}. */
int jv_profile_enable(int on);
int jv_profile_report(char* json, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* JYUTVOICE_HIP_H */
