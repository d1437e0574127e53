// Text encoder + duration predictor + length regulation stage on row buffers.
// jyutvoice/models/text_encoder.py:406-451 (TextEncoder.forward), :75-82 (prenet), :326-337 (6 post-LN layers),
// :216-248 (2-head RoPE attention), :276-281 (conv FFN); jyutvoice/models/duration_predictor.py:48-60;
// jyutvoice/models/jyutvoice_tts.py:175-176 (speaker projection), :184-203 (length regulation).
//
// ~1-2 % of the path's FLOPs: every contraction reuses the conv_gemm kernel -- the 288-wide attention heads as two
// batched GEMMs (q k^T with the K rows as the "weight" operand; P V with a transposed V) around a masked-softmax
// kernel -- so no second attention kernel has to be tuned for a stage this small.
#include "../../include/jyutvoice_hip.h"
#include "jv_model.h"
#include "jv_ops.h"

namespace jv {

int embed_sum(const long* x, const long* tone, const long* wp, const long* sp, const float* e0, const float* e1,
              const float* e2, const float* e3, float* out, int B, int Tt, int G, int S, hipStream_t st);
int concat_fill(const float* spk, const long* lang, const float* lang_emb, const long* xlen, float* h, int B, int Tt, int G,
                int S, hipStream_t st);
int lens_to_i32(const long* a, int* o, int n, int cap, hipStream_t st);
int rope_qk(float* qkv, int B, int Tt, int G, int S, hipStream_t st);
int enc_softmax(float* sc, const long* xlen, int B, int H, int Tt, int ld, hipStream_t st);
int transpose_v(const float* qkv, float* vt, int B, int Tt, int ld, int G, int S, hipStream_t st);
int add_rowvec(const float* x, const float* vec, float* out, int B, int Tt, int G, int S, int C, hipStream_t st);
int l2_normalize(const float* x, float* out, int B, int C, hipStream_t st);
int durations(const float* logw, const long* xlen, float scale, float* w_ceil, float* cum, long* ylen, int B, int Tt,
              hipStream_t st);
int paths(const float* cum, const long* xlen, const long* ylen, const float* mu_x, float* attn, float* mu_y, int B, int Tt,
          int Ty, hipStream_t st);

constexpr int E_G = 4, E_GAP = 4;

struct EncWs {
  long rows_alloc = 0;
  float *e0 = nullptr, *e1 = nullptr, *e2 = nullptr;   // [rows,192] prenet ping-pong
  float *h = nullptr, *y = nullptr, *xd = nullptr;     // [rows,576]
  float* qkv = nullptr;                                // [rows,1728]
  float* att = nullptr;                                // [rows,576]
  float* ffn = nullptr;                                // [rows,768]
  float *d1 = nullptr, *d2 = nullptr;                  // [rows,256]
  float* mu = nullptr;                                 // [rows,80]
  float* lw = nullptr;                                 // [rows,1]
  float* scores = nullptr;                             // [B,2,Tt,ld]
  float* vt = nullptr;                                 // [B,2,288,ld]
  float *spkn = nullptr, *cond = nullptr;              // [B,192], [B,576]
  float* cum = nullptr;                                // [B,Tt]
  unsigned char* mask = nullptr;
  int* lens = nullptr;
};

int enc_ws_create(Context& c) {
  EncWs* w = new EncWs();
  c.ews = w;
  const long rows = E_G + (long)c.max_batch * (c.max_tokens + E_GAP);
  w->rows_alloc = round_up((int)rows, 128) + 128;
  const size_t R = (size_t)w->rows_alloc;
  const size_t ld = (size_t)round_up(c.max_tokens, 32);
  auto F = [&](float** p, size_t floats) { return ws_alloc(c, floats * sizeof(float), reinterpret_cast<void**>(p)); };
  JV_TRY(F(&w->e0, R * 192));
  JV_TRY(F(&w->e1, R * 192));
  JV_TRY(F(&w->e2, R * 192));
  JV_TRY(F(&w->h, R * 576));
  JV_TRY(F(&w->y, R * 576));
  JV_TRY(F(&w->xd, R * 576));
  JV_TRY(F(&w->qkv, R * 1728));
  JV_TRY(F(&w->att, R * 576));
  JV_TRY(F(&w->ffn, R * 768));
  JV_TRY(F(&w->d1, R * 256));
  JV_TRY(F(&w->d2, R * 256));
  JV_TRY(F(&w->mu, R * 80));
  JV_TRY(F(&w->lw, R * 4));
  JV_TRY(F(&w->scores, (size_t)c.max_batch * 2 * c.max_tokens * ld));
  JV_TRY(F(&w->vt, (size_t)c.max_batch * 2 * 288 * ld));
  JV_TRY(F(&w->spkn, (size_t)c.max_batch * 192));
  JV_TRY(F(&w->cond, (size_t)c.max_batch * 576));
  JV_TRY(F(&w->cum, (size_t)c.max_batch * c.max_tokens));
  JV_TRY(ws_alloc(c, R, reinterpret_cast<void**>(&w->mask)));
  JV_TRY(ws_alloc(c, sizeof(int) * c.max_batch, reinterpret_cast<void**>(&w->lens)));
  return JV_OK;
}

void enc_ws_destroy(Context& c) {
  delete c.ews;
  c.ews = nullptr;
}

namespace {

ConvGemmArgs gemm_args(const float* A, int lda, long a_rows, long M, const GemmW& w, float* out, int ldo) {
  ConvGemmArgs a;
  conv_gemm_defaults(a);
  a.A = A; a.lda = lda; a.a_rows = a_rows; a.M = (int)M;
  a.Cin = w.Cin; a.ntaps = w.ntaps; a.tap_row0 = -(w.ntaps / 2); a.tap_dil = 1;   // "same" padding k//2
  a.W = w.w; a.ldw = w.ldw; a.n_rows_w = w.n_rows; a.N = w.N; a.bias = w.bias;
  a.W3 = w.w3; a.w3_plane = (long)w.n_rows * w.ldw;
  a.out = out; a.ldo = ldo;
  return a;
}

}  // namespace

int encoder_fwd(Context& c, const long* phone, const long* lang, const long* tone, const long* wpos, const long* spos,
                const long* xlen, const float* spk, int B, int Tt, float* x_out, float* mu_out, float* logw_out,
                float* spks_out, hipStream_t st) {
  if (!c.ready[MODEL_TTS]) return fail(JV_ERR_STATE, "tts weights not finalized");
  if (B < 1 || Tt < 1) return fail(JV_ERR_ARG, "batch and token count must be positive");
  if (B > c.max_batch || Tt > c.max_tokens || Tt > 512)
    return fail(JV_ERR_SHAPE, "batch/tokens exceed the capacity given to jv_create (and the 512-token attention limit)");
  EncWs& w = *c.ews;
  const EncoderW& e = c.enc;
  const int S = Tt + E_GAP;
  const long M = E_G + (long)B * S;
  const long AR = w.rows_alloc;

  JV_TRY(lens_to_i32(xlen, w.lens, B, Tt, st));
  JV_TRY(row_meta(w.mask, nullptr, w.lens, B, 1, E_G, S, Tt, AR, 1, 0, st));

  // speaker projection for the flow decoder: Linear(192->80)(normalize(spk))  (jyutvoice_tts.py:175-176)
  JV_TRY(l2_normalize(spk, w.spkn, B, 192, st));
  {
    ConvGemmArgs a = gemm_args(w.spkn, 192, B, B, e.spk_affine, spks_out, 80);
    JV_TRY(conv_gemm(a, 1, st));
  }

  // embeddings -> prenet (3 x conv k5 + channel-LN(eps 1e-4) + ReLU, 1x1 proj, residual, mask)
  JV_TRY(embed_sum(phone, tone, wpos, spos, e.emb, e.tone_emb, e.wpos_emb, e.spos_emb, w.e0, B, Tt, E_G, S, st));
  const float* cur = w.e0;
  for (int i = 0; i < 3; ++i) {
    ConvGemmArgs a = gemm_args(cur, 192, AR, M, e.pre_conv[i], w.e1, 192);
    a.rowmask_in = w.mask;
    JV_TRY(conv_gemm(a, 1, st));
    JV_TRY(layernorm_rows(w.e1, nullptr, w.e2, e.pre_ln[i].g, e.pre_ln[i].b, 1e-4f, M, 192, nullptr, st, 1));
    cur = w.e2;
  }
  {
    ConvGemmArgs a = gemm_args(w.e2, 192, AR, M, e.pre_proj, w.h, 576);   // columns [0,192) of the 576-wide concat
    a.res1 = w.e0; a.ldr1 = 192;
    a.rowmask_out = w.mask;
    JV_TRY(conv_gemm(a, 1, st));
  }
  JV_TRY(concat_fill(spk, lang, e.lang_emb, xlen, w.h, B, Tt, E_G, S, st));

  // 6 post-LN layers; every LayerNorm writes masked rows so "x * x_mask" never needs its own pass
  const int ld = round_up(Tt, 32);
  for (int i = 0; i < ENC_LAYERS; ++i) {
    const EncLayerW& L = e.layer[i];
    ConvGemmArgs a = gemm_args(w.h, 576, AR, M, L.qkv, w.qkv, 1728);
    JV_TRY(conv_gemm(a, 1, st));
    JV_TRY(rope_qk(w.qkv, B, Tt, E_G, S, st));
    // scores[b,h] = Q_bh K_bh^T : the K rows are the K-contiguous "weight" operand
    conv_gemm_defaults(a);
    a.A = w.qkv + (long)E_G * 1728; a.lda = 1728; a.a_rows = Tt; a.M = Tt; a.Cin = 288; a.ntaps = 1;
    a.W = w.qkv + (long)E_G * 1728 + 576; a.ldw = 1728; a.n_rows_w = Tt; a.N = Tt;
    a.out = w.scores; a.ldo = ld;
    a.nb2 = 2;
    a.sA1 = (long)S * 1728; a.sA2 = 288; a.sW1 = (long)S * 1728; a.sW2 = 288; a.sO1 = 2L * Tt * ld; a.sO2 = (long)Tt * ld;
    JV_TRY(conv_gemm(a, B * 2, st));
    JV_TRY(enc_softmax(w.scores, xlen, B, 2, Tt, ld, st));
    JV_TRY(transpose_v(w.qkv, w.vt, B, Tt, ld, E_G, S, st));
    // att[b, :, h*288:(h+1)*288] = P_bh V_bh
    conv_gemm_defaults(a);
    a.A = w.scores; a.lda = ld; a.a_rows = Tt; a.M = Tt; a.Cin = ld; a.ntaps = 1;
    a.W = w.vt; a.ldw = ld; a.n_rows_w = 288; a.N = 288;
    a.out = w.att + (long)E_G * 576; a.ldo = 576;
    a.nb2 = 2;
    a.sA1 = 2L * Tt * ld; a.sA2 = (long)Tt * ld; a.sW1 = 2L * 288 * ld; a.sW2 = 288L * ld; a.sO1 = (long)S * 576; a.sO2 = 288;
    JV_TRY(conv_gemm(a, B * 2, st));
    a = gemm_args(w.att, 576, AR, M, L.o, w.y, 576);
    JV_TRY(conv_gemm(a, 1, st));
    JV_TRY(layernorm_rows(w.h, w.y, w.h, L.n1.g, L.n1.b, 1e-4f, M, 576, w.mask, st));
    a = gemm_args(w.h, 576, AR, M, L.ffn1, w.ffn, 768);
    a.rowmask_in = w.mask;
    a.act = ACT_RELU;
    JV_TRY(conv_gemm(a, 1, st));
    a = gemm_args(w.ffn, 768, AR, M, L.ffn2, w.y, 576);
    a.rowmask_in = w.mask;
    a.rowmask_out = w.mask;
    JV_TRY(conv_gemm(a, 1, st));
    JV_TRY(layernorm_rows(w.h, w.y, w.h, L.n2.g, L.n2.b, 1e-4f, M, 576, w.mask, st));
  }
  // outputs: x (masked), mu = proj(x) * mask
  JV_TRY(rows_to_cf(w.h, 576, 0, E_G, S, x_out, 576L * Tt, B, 576, Tt, w.lens, st));
  {
    ConvGemmArgs a = gemm_args(w.h, 576, AR, M, e.proj, w.mu, 80);
    a.rowmask_out = w.mask;
    JV_TRY(conv_gemm(a, 1, st));
    JV_TRY(rows_to_cf(w.mu, 80, 0, E_G, S, mu_out, 80L * Tt, B, 80, Tt, w.lens, st));
  }
  // duration predictor on x + cond(spk_raw): conv k3 -> ReLU -> LN, twice, then 1x1 -> logw * mask
  {
    ConvGemmArgs a = gemm_args(spk, 192, B, B, e.dp_cond, w.cond, 576);
    JV_TRY(conv_gemm(a, 1, st));
    JV_TRY(add_rowvec(w.h, w.cond, w.xd, B, Tt, E_G, S, 576, st));
    a = gemm_args(w.xd, 576, AR, M, e.dp_conv1, w.d1, 256);
    a.rowmask_in = w.mask; a.act = ACT_RELU;
    JV_TRY(conv_gemm(a, 1, st));
    JV_TRY(layernorm_rows(w.d1, nullptr, w.d2, e.dp_ln1.g, e.dp_ln1.b, 1e-4f, M, 256, nullptr, st));
    a = gemm_args(w.d2, 256, AR, M, e.dp_conv2, w.d1, 256);
    a.rowmask_in = w.mask; a.act = ACT_RELU;
    JV_TRY(conv_gemm(a, 1, st));
    JV_TRY(layernorm_rows(w.d1, nullptr, w.d2, e.dp_ln2.g, e.dp_ln2.b, 1e-4f, M, 256, nullptr, st));
    a = gemm_args(w.d2, 256, AR, M, e.dp_proj, w.lw, 1);
    a.rowmask_in = w.mask; a.rowmask_out = w.mask;
    JV_TRY(conv_gemm(a, 1, st));
    JV_TRY(rows_to_cf(w.lw, 1, 0, E_G, S, logw_out, (long)Tt, B, 1, Tt, w.lens, st));
  }
  return JV_OK;
}

int length_regulate(Context& c, const float* logw, const long* xlen, const float* mu_x, int B, int Tt, float scale,
                    float* w_ceil, long* ylen, int Ty, float* attn, float* mu_y, hipStream_t st) {
  if (B < 1 || Tt < 1 || B > c.max_batch || Tt > c.max_tokens) return fail(JV_ERR_SHAPE, "length_regulate: bad batch/tokens");
  EncWs& w = *c.ews;
  if (!attn) return durations(logw, xlen, scale, w_ceil, w.cum, ylen, B, Tt, st);
  if (!mu_y || Ty < 1) return fail(JV_ERR_ARG, "length_regulate: phase 2 needs attn, mu_y and Ty >= 1");
  // recompute the cumulative durations (phase 2 may follow a different phase-1 call on the same context)
  JV_TRY(durations(logw, xlen, scale, w_ceil, w.cum, ylen, B, Tt, st));
  return paths(w.cum, xlen, ylen, mu_x, attn, mu_y, B, Tt, Ty, st);
}

}  // namespace jv


extern "C" {

int jv_encoder_fwd(jv_context* ctx, const int64_t* phone, const int64_t* lang, const int64_t* tone, const int64_t* word_pos,
                   const int64_t* syllable_pos, const int64_t* x_lengths, const float* spk, int B, int Tt, float* x,
                   float* mu_x, float* logw, float* spks_proj, void* stream) {
  if (!ctx || !phone || !lang || !tone || !word_pos || !syllable_pos || !x_lengths || !spk || !x || !mu_x || !logw || !spks_proj)
    return jv::fail(JV_ERR_ARG, "jv_encoder_fwd: null argument");
  JV_HIP(hipSetDevice(ctx->c.device));
  return jv::encoder_fwd(ctx->c, reinterpret_cast<const long*>(phone), reinterpret_cast<const long*>(lang),
                         reinterpret_cast<const long*>(tone), reinterpret_cast<const long*>(word_pos),
                         reinterpret_cast<const long*>(syllable_pos), reinterpret_cast<const long*>(x_lengths), spk, B, Tt, x,
                         mu_x, logw, spks_proj, static_cast<hipStream_t>(stream));
}

int jv_length_regulate(jv_context* ctx, const float* logw, const int64_t* x_lengths, const float* mu_x, int B, int Tt,
                       float length_scale, float* w_ceil, int64_t* y_lengths, int Ty, float* attn, float* mu_y,
                       void* stream) {
  if (!ctx || !logw || !x_lengths || !w_ceil || !y_lengths) return jv::fail(JV_ERR_ARG, "jv_length_regulate: null argument");
  JV_HIP(hipSetDevice(ctx->c.device));
  return jv::length_regulate(ctx->c, logw, reinterpret_cast<const long*>(x_lengths), mu_x, B, Tt, length_scale, w_ceil,
                             reinterpret_cast<long*>(y_lengths), Ty, attn, mu_y, static_cast<hipStream_t>(stream));
}

}  // extern "C"
