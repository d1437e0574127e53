// Prompt (voice-cloning) branch: FlowEncoder of the reference's infer.py:35-83 -- speech tokens [B,Tk] -> prompt_h
// [B, 2*Tk, 80] -- on row buffers.  jyutvoice/transformer/upsample_encoder.py:329-375 (forward), :20-61 (Upsample1D),
// :64-134 (PreLookaheadLayer), subsampling.py:84-115, embedding.py:201-296, encoder_layer.py:241-319,
// attention.py:204-334 (relative-position attention, rel_shift), positionwise_feed_forward.py:47-55.
//
// Two stages of the same shape: T1 = Tk rows per utterance (6 blocks), then T2 = 2*Tk rows (4 blocks).  Every Linear /
// Conv1d is the conv_gemm kernel on the bf16x6 path; the 8 x 64 relative-position attention is three batched fp32-MFMA GEMMs
// ((q + u) K^T, (q + v) P^T, softmax V) around one rel-shift + masked-softmax kernel -- the stage is ~2 % of a prompted
// synthesis, so no dedicated attention kernel is tuned for it (same choice as the text encoder, encoder.hip).
// Batches are per-utterance loops of the B = 1 reference: rows at and beyond an utterance's length are zero for the
// look-ahead / upsampling convolutions (select-zero row mask) and masked as attention keys.
#include <math.h>

#include "../../include/jyutvoice_hip.h"
#include "jv_model.h"
#include "jv_ops.h"

namespace jv {

int prompt_embed(const long* tok, const long* len, const float* emb, float* rows, int B, int T, int G, int S, int vocab,
                 hipStream_t st);
int rel_pos_table(float* pe, const float* div, int T, hipStream_t st);
int add_pos_bias(const float* qkv, const float* u, const float* v, float* qu, float* qv, long rows, hipStream_t st);
int rel_softmax(float* ac, const float* bd, const long* len, int len_mul, int B, int H, int T, int ld, int ldb, hipStream_t st);
int prompt_transpose_v(const float* qkv, float* vt, int B, int T, int ld, int G, int S, hipStream_t st);
int repeat_rows2(const float* src, float* dst, int B, int T, int G, int S, int G2, int S2, hipStream_t st);
int rows_to_btc(const float* rows, const long* len, int len_mul, float* out, int B, int T, int C, int G, int S, hipStream_t st);
int lens_to_i32(const long* a, int* o, int n, int cap, hipStream_t st);   // encops.hip

constexpr int P_G = 8, P_GAP = 8;   // guard rows >= the widest context (look-ahead 3, causal 2, upsampling conv 4)

// Sized on first use for the (B, Tk) asked for and regrown when a larger call arrives: prompts are short and rare next to
// synthesis, so this memory is not reserved at jv_create.
struct PromptWs {
  int B = 0, Tk = 0;
  long rows = 0;
  std::vector<void*> allocs;
  float *x = nullptr, *y = nullptr, *ln = nullptr, *qu = nullptr, *qv = nullptr, *att = nullptr;   // [rows,512]
  float* qkv = nullptr;                                                                         // [rows,1536]
  float* ffn = nullptr;                                                                         // [rows,2048]
  float* o80 = nullptr;                                                                         // [rows,80]
  float *pe = nullptr, *p = nullptr;                                                            // [2*T2,512]
  float *ac = nullptr, *bd = nullptr, *vt = nullptr;
  unsigned char *mask1 = nullptr, *mask2 = nullptr;
  int* lens_i = nullptr;
};

void prompt_ws_destroy(Context& c) {
  if (!c.pws) return;
  for (void* p : c.pws->allocs) (void)hipFree(p);
  delete c.pws;
  c.pws = nullptr;
}

namespace {

int ensure_ws(Context& c, int B, int Tk) {
  if (c.pws && c.pws->B >= B && c.pws->Tk >= Tk) return JV_OK;
  const int nb = c.pws ? (B > c.pws->B ? B : c.pws->B) : B;
  const int nt = c.pws ? (Tk > c.pws->Tk ? Tk : c.pws->Tk) : Tk;
  JV_HIP(hipDeviceSynchronize());     // nothing may still be using the buffers about to be freed
  prompt_ws_destroy(c);
  PromptWs* w = new PromptWs();
  c.pws = w;
  w->B = nb; w->Tk = nt;
  const int T2 = 2 * nt;
  w->rows = round_up(P_G + nb * (T2 + P_GAP), 128) + 256;
  const size_t R = (size_t)w->rows;
  const size_t ld = (size_t)round_up(T2, 32), ldb = (size_t)round_up(2 * T2 - 1, 32);
  auto A = [&](void** p, size_t bytes) -> int {
    JV_HIP(hipMalloc(p, bytes));
    w->allocs.push_back(*p);
    JV_HIP(hipMemset(*p, 0, bytes));
    return JV_OK;
  };
  auto F = [&](float** p, size_t floats) { return A(reinterpret_cast<void**>(p), floats * sizeof(float)); };
  JV_TRY(F(&w->x, R * 512));
  JV_TRY(F(&w->y, R * 512));
  JV_TRY(F(&w->ln, R * 512));
  JV_TRY(F(&w->qu, R * 512));
  JV_TRY(F(&w->qv, R * 512));
  JV_TRY(F(&w->att, R * 512));
  JV_TRY(F(&w->qkv, R * 1536));
  JV_TRY(F(&w->ffn, R * 2048));
  JV_TRY(F(&w->o80, R * 80));
  JV_TRY(F(&w->pe, (size_t)ldb * 512));
  JV_TRY(F(&w->p, (size_t)(ldb + 128) * 512));
  JV_TRY(F(&w->ac, (size_t)nb * 8 * T2 * ld));
  JV_TRY(F(&w->bd, (size_t)nb * 8 * T2 * ldb));
  JV_TRY(F(&w->vt, (size_t)nb * 8 * 64 * ld));
  JV_TRY(A(reinterpret_cast<void**>(&w->mask1), R));
  JV_TRY(A(reinterpret_cast<void**>(&w->mask2), R));
  JV_TRY(A(reinterpret_cast<void**>(&w->lens_i), sizeof(int) * nb));
  return JV_OK;
}

ConvGemmArgs lin_args(const float* A, int lda, long a_rows, long M, const GemmW& w, float* out, int ldo) {
  ConvGemmArgs a;
  conv_gemm_defaults(a);
  a.A = A; a.lda = lda; a.a_rows = a_rows; a.M = (int)M;
  a.Cin = w.Cin; a.ntaps = w.ntaps; a.tap_row0 = 0; a.tap_dil = 1;
  a.W = w.w; a.ldw = w.ldw; a.n_rows_w = w.n_rows; a.N = w.N; a.bias = w.bias;
  a.W3 = w.w3; a.w3_plane = (long)w.n_rows * w.ldw;
  a.out = out; a.ldo = ldo;
  return a;
}

// one pre-LN block on T rows per utterance (geometry G, S): x += MHA_rel(LN(x)); x += W2 silu(W1 LN(x))
int conformer_block(Context& c, const ConfBlockW& k, int B, int T, int G, int S, long M, const long* len, int len_mul,
                    hipStream_t st) {
  PromptWs& w = *c.pws;
  const long AR = w.rows;
  const int ld = round_up(T, 32), npos = 2 * T - 1, ldb = round_up(npos, 32);
  JV_TRY(layernorm_rows(w.x, nullptr, w.ln, k.n_mha.g, k.n_mha.b, 1e-5f, M, 512, nullptr, st));
  ConvGemmArgs a = lin_args(w.ln, 512, AR, M, k.qkv, w.qkv, 1536);
  JV_TRY(conv_gemm(a, 1, st));
  a = lin_args(w.pe, 512, npos, npos, k.pos, w.p, 512);                       // p = linear_pos(pos_emb), shared by the batch
  JV_TRY(conv_gemm(a, 1, st));
  JV_TRY(add_pos_bias(w.qkv, k.u, k.v, w.qu, w.qv, M, st));
  // ac[b,h] = (q + u) K^T : the K rows are the K-contiguous "weight" operand
  conv_gemm_defaults(a);
  a.A = w.qu + (long)G * 512; a.lda = 512; a.a_rows = T; a.M = T; a.Cin = 64; a.ntaps = 1;
  a.W = w.qkv + (long)G * 1536 + 512; a.ldw = 1536; a.n_rows_w = T; a.N = T;
  a.out = w.ac; a.ldo = ld;
  a.nb2 = 8;
  a.sA1 = (long)S * 512; a.sA2 = 64; a.sW1 = (long)S * 1536; a.sW2 = 64; a.sO1 = 8L * T * ld; a.sO2 = (long)T * ld;
  JV_TRY(conv_gemm(a, B * 8, st));
  // bd[b,h] = (q + v) P_h^T over all 2T-1 relative positions (rel_shift happens inside the softmax's indexing)
  conv_gemm_defaults(a);
  a.A = w.qv + (long)G * 512; a.lda = 512; a.a_rows = T; a.M = T; a.Cin = 64; a.ntaps = 1;
  a.W = w.p; a.ldw = 512; a.n_rows_w = npos; a.N = npos;
  a.out = w.bd; a.ldo = ldb;
  a.nb2 = 8;
  a.sA1 = (long)S * 512; a.sA2 = 64; a.sW1 = 0; a.sW2 = 64; a.sO1 = 8L * T * ldb; a.sO2 = (long)T * ldb;
  JV_TRY(conv_gemm(a, B * 8, st));
  JV_TRY(rel_softmax(w.ac, w.bd, len, len_mul, B, 8, T, ld, ldb, st));
  JV_TRY(prompt_transpose_v(w.qkv, w.vt, B, T, ld, G, S, st));
  // att[b, :, h*64:(h+1)*64] = P_bh V_bh
  conv_gemm_defaults(a);
  a.A = w.ac; a.lda = ld; a.a_rows = T; a.M = T; a.Cin = ld; a.ntaps = 1;
  a.W = w.vt; a.ldw = ld; a.n_rows_w = 64; a.N = 64;
  a.out = w.att + (long)G * 512; a.ldo = 512;
  a.nb2 = 8;
  a.sA1 = 8L * T * ld; a.sA2 = (long)T * ld; a.sW1 = 8L * 64 * ld; a.sW2 = 64L * ld; a.sO1 = (long)S * 512; a.sO2 = 64;
  JV_TRY(conv_gemm(a, B * 8, st));
  a = lin_args(w.att, 512, AR, M, k.out, w.x, 512);
  a.res1 = w.x; a.ldr1 = 512;
  JV_TRY(conv_gemm(a, 1, st));
  JV_TRY(layernorm_rows(w.x, nullptr, w.ln, k.n_ff.g, k.n_ff.b, 1e-5f, M, 512, nullptr, st));
  a = lin_args(w.ln, 512, AR, M, k.w1, w.ffn, 2048);
  a.act = ACT_SILU;
  JV_TRY(conv_gemm(a, 1, st));
  a = lin_args(w.ffn, 2048, AR, M, k.w2, w.x, 512);
  a.res1 = w.x; a.ldr1 = 512;
  return conv_gemm(a, 1, st);
}

}  // namespace

int prompt_encoder_fwd(Context& c, const long* tok, const long* len, int B, int Tk, float* h_out, hipStream_t st) {
  if (!c.ready[MODEL_PROMPT]) return fail(JV_ERR_STATE, "prompt encoder weights not finalized");
  if (B < 1 || Tk < 1) return fail(JV_ERR_ARG, "batch and token count must be positive");
  if (B > c.max_batch || 2 * Tk > c.max_frames || Tk > 2048)
    return fail(JV_ERR_SHAPE, "prompt batch/tokens exceed the capacity given to jv_create (2*tokens <= max_frames, tokens <= 2048)");
  JV_TRY(ensure_ws(c, B, Tk));
  PromptWs& w = *c.pws;
  const PromptW& e = c.prompt;
  const long AR = w.rows;
  const int T1 = Tk, S1 = T1 + P_GAP, T2 = 2 * Tk, S2 = T2 + P_GAP;
  const long M1 = P_G + (long)B * S1, M2 = P_G + (long)B * S2;

  JV_TRY(lens_to_i32(len, w.lens_i, B, Tk, st));
  JV_TRY(row_meta(w.mask1, nullptr, w.lens_i, B, 1, P_G, S1, T1, AR, 1, 0, st));
  JV_TRY(row_meta(w.mask2, nullptr, w.lens_i, B, 1, P_G, S2, T2, AR, 2, 0, st));

  // ---- stage 1: embedding -> Linear + LayerNorm (* sqrt 512) -> look-ahead convs -> 6 blocks ---------------------------
  JV_TRY(fill(w.y, 0.f, M1 * 512, st));                     // gap rows of the embedding buffer must read as zero tokens
  JV_TRY(prompt_embed(tok, len, e.emb, w.y, B, T1, P_G, S1, PR_VOCAB, st));
  ConvGemmArgs a = lin_args(w.y, 512, AR, M1, e.emb_lin, w.ln, 512);
  JV_TRY(conv_gemm(a, 1, st));
  JV_TRY(layernorm_rows(w.ln, nullptr, w.x, e.emb_ln.g, e.emb_ln.b, 1e-5f, M1, 512, nullptr, st));
  JV_TRY(rel_pos_table(w.pe, e.div, T1, st));
  // conv1: rows t .. t+3 of the valid frames (zeros beyond the end), bias; LeakyReLU(0.01) is conv2's prologue
  a = lin_args(w.x, 512, AR, M1, e.look1, w.y, 512);
  a.tap_row0 = 0;
  a.rowmask_in = w.mask1;
  JV_TRY(conv_gemm(a, 1, st));
  // conv2: rows t-2 .. t of leaky(conv1) over the valid frames (zeros before the start), + bias + x
  a = lin_args(w.y, 512, AR, M1, e.look2, w.ln, 512);
  a.tap_row0 = -2;
  a.rowmask_in = w.mask1;
  a.pro = PRO_LRELU; a.pro_slope = 0.01f;
  a.res1 = w.x; a.ldr1 = 512;
  JV_TRY(conv_gemm(a, 1, st));
  JV_HIP(hipMemcpyAsync(w.x, w.ln, sizeof(float) * M1 * 512, hipMemcpyDeviceToDevice, st));
  for (int i = 0; i < PR_BLOCKS; ++i) JV_TRY(conformer_block(c, e.blk[i], B, T1, P_G, S1, M1, len, 1, st));

  // ---- stage 2: nearest x2 -> conv k5 over rows u-4 .. u -> Linear + LayerNorm (* sqrt 512) -> 4 blocks -> LN -> proj ----
  JV_TRY(fill(w.y, 0.f, M2 * 512, st));
  JV_TRY(repeat_rows2(w.x, w.y, B, T1, P_G, S1, P_G, S2, st));
  a = lin_args(w.y, 512, AR, M2, e.up_conv, w.ln, 512);
  a.tap_row0 = -4;
  a.rowmask_in = w.mask2;
  JV_TRY(conv_gemm(a, 1, st));
  a = lin_args(w.ln, 512, AR, M2, e.up_emb_lin, w.y, 512);
  JV_TRY(conv_gemm(a, 1, st));
  JV_TRY(layernorm_rows(w.y, nullptr, w.x, e.up_emb_ln.g, e.up_emb_ln.b, 1e-5f, M2, 512, nullptr, st));
  JV_TRY(rel_pos_table(w.pe, e.div, T2, st));
  for (int i = 0; i < PR_UP_BLOCKS; ++i) JV_TRY(conformer_block(c, e.up[i], B, T2, P_G, S2, M2, len, 2, st));
  JV_TRY(layernorm_rows(w.x, nullptr, w.ln, e.after.g, e.after.b, 1e-5f, M2, 512, nullptr, st));
  a = lin_args(w.ln, 512, AR, M2, e.proj, w.o80, 80);
  JV_TRY(conv_gemm(a, 1, st));
  return rows_to_btc(w.o80, len, 2, h_out, B, T2, 80, P_G, S2, st);
}

}  // namespace jv

extern "C" {

int jv_prompt_encoder_fwd(jv_context* ctx, const int64_t* tokens, const int64_t* token_len, int B, int Tk, float* prompt_h,
                          void* stream) {
  if (!ctx || !tokens || !token_len || !prompt_h) return jv::fail(JV_ERR_ARG, "jv_prompt_encoder_fwd: null argument");
  JV_HIP(hipSetDevice(ctx->c.device));
  return jv::prompt_encoder_fwd(ctx->c, reinterpret_cast<const long*>(tokens), reinterpret_cast<const long*>(token_len), B, Tk,
                                prompt_h, static_cast<hipStream_t>(stream));
}

}  // extern "C"
