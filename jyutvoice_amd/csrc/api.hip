// extern "C" surface of libjyutvoice_hip.so (include/jyutvoice_hip.h).  Nothing throws across it.
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <new>

#include "../../include/jyutvoice_hip.h"
#include "jv_model.h"
#include "jv_ops.h"
#include "hiftconv_kernel.h"
#include "rowconv_kernel.h"

namespace jv {

int hiftconv(const HiftConvArgs& a, int C, hipStream_t st);   // hiftconv.hip
int rowgemm(const RowGemmArgs& a, int epi, hipStream_t st);   // rowgemm.hip
int rowconv(const RowConvArgs& a, hipStream_t st);

int split3_planes(const float* src, unsigned short* dst, long n, hipStream_t st);   // registry.hip
int split2h_planes(const float* src, int rows, int ld, float* stats, unsigned short* dst, float* colscale, hipStream_t st);
float h3_scale_for_bound(float bound);
int split2h_rows(const float* x, long ld, unsigned short* dst, long plane, long rows, int C, float scale, hipStream_t st,
                 long ldd = 0);   // rowops.hip

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

int ws_alloc(Context& c, size_t bytes, void** out) {
  void* p = nullptr;
  bytes = (bytes + 255) & ~(size_t)255;
  JV_HIP(hipMalloc(&p, bytes));
  JV_HIP(hipMemset(p, 0, bytes));
  c.ws_allocs.push_back(p);
  *out = p;
  return JV_OK;
}

int hift_ws_create(Context& c);   // hift.hip
int enc_ws_create(Context& c);    // encoder.hip
void flow_ws_destroy(Context& c);
void hift_ws_destroy(Context& c);
void enc_ws_destroy(Context& c);

}  // namespace jv

using jv::Context;


#define CTX_GUARD(ctx)                                                       \
  if (!(ctx)) return jv::fail(JV_ERR_ARG, "null context");                   \
  if ((ctx)->c.broken) return jv::fail(JV_ERR_STATE, "context unusable: a workspace allocation failed and the previous capacities could not be restored (jv_reserve); destroy it"); \
  {                                                                          \
    hipError_t _e = hipSetDevice((ctx)->c.device);                           \
    if (_e != hipSuccess) return jv::fail(JV_ERR_HIP, hipGetErrorString(_e)); \
  }

// Everything whose size follows the capacities (max_batch, max_frames, max_tokens).  Weights (raw + packed arenas), the
// noise tensor and the mel filterbank do not: jv_reserve re-creates only this, never re-uploads or re-packs a weight.
static int workspaces_create(Context& c) {
  int rc = jv::flow_ws_create(c);
  if (rc == JV_OK) rc = jv::hift_ws_create(c);
  if (rc == JV_OK) rc = jv::enc_ws_create(c);
  return rc;
}
static void workspaces_destroy(Context& c) {
  for (void* p : c.ws_allocs) (void)hipFree(p);
  c.ws_allocs.clear();
  jv::flow_ws_destroy(c);      // also drops captured Euler-step graphs: they point into the old buffers
  jv::hift_ws_destroy(c);
  jv::enc_ws_destroy(c);
  jv::prompt_ws_destroy(c);    // sized on demand by its own calls (prompt.hip); capped by max_frames
}

// Scratch of the operator hooks (jv_op_*: test and tuning aids, but exported): one growing buffer per (device, hook slot),
// looked up under a lock, so a buffer allocated on device A is never handed out after hipSetDevice(B) and two threads cannot
// race the grow.  Calls of ONE hook on ONE device still share the buffer: the caller serialises those (stream order does).
namespace {
enum OpSlot { OPS_X6 = 0, OPS_H3, OPS_H3_A, OPS_CONV, OPS_RG, OPS_RG_A, OPS_RC, OPS_HIFT, OPS_ATTN, OPS_COUNT };
int op_scratch(int slot, size_t need, void** out) {
  struct Buf { void* p = nullptr; size_t cap = 0; };
  static std::mutex mu;
  static Buf bufs[64][OPS_COUNT];
  int dev = 0;
  JV_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  Buf& b = bufs[dev & 63][slot];
  if (need > b.cap) {
    if (b.p) (void)hipFree(b.p);      // (synchronises the device: nothing still reads the old buffer)
    b.p = nullptr; b.cap = 0;
    JV_HIP(hipMalloc(&b.p, need));
    b.cap = need;
  }
  *out = b.p;
  return JV_OK;
}
}  // namespace

extern "C" {

const char* jv_last_error(void) { return jv::g_last_error.c_str(); }

int jv_create(jv_context** out, int device, int max_batch, int max_frames, int max_tokens) {
  if (!out) return jv::fail(JV_ERR_ARG, "jv_create: null out pointer");
  *out = nullptr;
  if (max_batch < 1 || max_frames < 1 || max_tokens < 1) return jv::fail(JV_ERR_ARG, "jv_create: capacities must be >= 1");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return jv::fail(JV_ERR_HIP, "jv_create: no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return jv::fail(JV_ERR_ARG, "jv_create: bad device index");
  JV_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  JV_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return jv::fail(JV_ERR_HIP, std::string("jv_create: built for gfx950 only, found ") + prop.gcnArchName);
  jv_context* ctx = new (std::nothrow) jv_context();
  if (!ctx) return jv::fail(JV_ERR_HIP, "jv_create: out of host memory");
  Context& c = ctx->c;
  c.device = device;
  c.max_batch = max_batch;
  c.step_graphs = getenv("JV_STEP_GRAPH") != nullptr;
  c.exact_range = getenv("JV_EXACT_RANGE") != nullptr;
  c.dma_a = getenv("JV_DMA_A") != nullptr;
  c.no_rowgemm = getenv("JV_NO_ROWGEMM") != nullptr;
  c.no_splitk = getenv("JV_NO_SPLITK") != nullptr;
  c.rg_ff1 = getenv("JV_TILE_FF1") == nullptr;
  c.no_ffn_fuse = getenv("JV_NO_FFN_FUSE") != nullptr;
  c.no_block_fuse = getenv("JV_NO_BLOCK_FUSE") != nullptr;
  c.no_qkv_split = getenv("JV_NO_QKV_SPLIT") != nullptr;
  c.no_compact = getenv("JV_NO_COMPACT") != nullptr;
  c.no_res_fold = getenv("JV_NO_RES_FOLD") != nullptr;
  c.no_res_pair = getenv("JV_NO_RES_PAIR") != nullptr;
  c.no_res_qkv = getenv("JV_NO_RES_QKV") != nullptr;
  c.no_ln_fold = getenv("JV_NO_LN_FOLD") != nullptr;
  c.no_temb_pre = getenv("JV_NO_TEMB_PRE") != nullptr;
  c.no_attn_planes = getenv("JV_NO_ATTN_PLANES") != nullptr;
  c.no_hiftconv = getenv("JV_NO_HIFTCONV") != nullptr;
  c.no_hift_pair = getenv("JV_NO_HIFT_PAIR") != nullptr;
  c.attn_rows = getenv("JV_ATTN_ROWS") != nullptr;
  c.attn_single = getenv("JV_NO_ATTN_SINGLE") == nullptr;
  c.max_frames = max_frames;
  c.max_tokens = max_tokens;
  jv::build_registry(c);
  int rc = jv::conv_gemm_init();
  if (rc == JV_OK) rc = workspaces_create(c);
  // the noise tensor is a weight-like constant: it lives outside the workspace and survives jv_reserve
  if (rc == JV_OK && hipMalloc(reinterpret_cast<void**>(&c.noise), sizeof(float) * jv::N_FEATS * jv::NOISE_FRAMES) != hipSuccess)
    rc = jv::fail(JV_ERR_HIP, "jv_create: out of device memory (noise tensor)");
  if (rc != JV_OK) {
    jv_destroy(ctx);
    return rc;
  }
  *out = ctx;
  return JV_OK;
}

int jv_reserve(jv_context* ctx, int max_batch, int max_frames, int max_tokens) {
  CTX_GUARD(ctx);
  Context& c = ctx->c;
  if (max_batch < 1 || max_frames < 1 || max_tokens < 1) return jv::fail(JV_ERR_ARG, "jv_reserve: capacities must be >= 1");
  if (max_batch == c.max_batch && max_frames == c.max_frames && max_tokens == c.max_tokens) return JV_OK;
  JV_HIP(hipDeviceSynchronize());      // nothing queued may still use the old workspace (or replay a graph that points into it)
  const int old_b = c.max_batch, old_f = c.max_frames, old_t = c.max_tokens;
  workspaces_destroy(c);
  c.max_batch = max_batch;
  c.max_frames = max_frames;
  c.max_tokens = max_tokens;
  int rc = workspaces_create(c);
  if (rc == JV_OK) return JV_OK;
  // Failure-atomic: a create that stopped part-way leaves workspace structs with null buffers behind.  Drop them and
  // come back at the capacities the caller had (they fitted before); if even that fails the context refuses every call.
  const std::string why = jv::g_last_error;
  (void)hipGetLastError();
  workspaces_destroy(c);
  c.max_batch = old_b;
  c.max_frames = old_f;
  c.max_tokens = old_t;
  if (workspaces_create(c) != JV_OK) {
    workspaces_destroy(c);
    c.broken = true;
    return jv::fail(rc, "jv_reserve: " + why + " -- and the previous capacities could not be restored: context unusable");
  }
  return jv::fail(rc, "jv_reserve: " + why + " (capacities unchanged)");
}

int jv_usable(const jv_context* ctx) { return ctx && !ctx->c.broken ? 1 : 0; }

void jv_destroy(jv_context* ctx) {
  if (!ctx) return;
  Context& c = ctx->c;
  (void)hipSetDevice(c.device);
  (void)hipDeviceSynchronize();
  workspaces_destroy(c);
  if (c.noise) (void)hipFree(c.noise);
  c.raw_arena.release();
  c.packed.release();
  jv::audio_ws_destroy(c);
  delete ctx;
}

int jv_num_tensors(const jv_context* ctx) { return ctx ? (int)ctx->c.raw.size() : 0; }
const char* jv_tensor_name(const jv_context* ctx, int i) {
  return (ctx && i >= 0 && i < (int)ctx->c.raw.size()) ? ctx->c.raw[i].name.c_str() : nullptr;
}
int jv_tensor_model(const jv_context* ctx, int i) {
  return (ctx && i >= 0 && i < (int)ctx->c.raw.size()) ? ctx->c.raw[i].model : -1;
}
int jv_tensor_ndim(const jv_context* ctx, int i) {
  return (ctx && i >= 0 && i < (int)ctx->c.raw.size()) ? (int)ctx->c.raw[i].shape.size() : -1;
}
int64_t jv_tensor_dim(const jv_context* ctx, int i, int d) {
  if (!ctx || i < 0 || i >= (int)ctx->c.raw.size()) return -1;
  const auto& s = ctx->c.raw[i].shape;
  return (d >= 0 && d < (int)s.size()) ? s[d] : -1;
}

int jv_load_tensor(jv_context* ctx, const char* name, const float* data, const int64_t* shape, int ndim, int on_device,
                   void* stream) {
  CTX_GUARD(ctx);
  if (!name || !data || !shape) return jv::fail(JV_ERR_ARG, "jv_load_tensor: null argument");
  Context& c = ctx->c;
  auto it = c.index.find(name);
  if (it == c.index.end()) return jv::fail(JV_ERR_NAME, std::string("unexpected key: ") + name);
  jv::RawTensor& t = c.raw[it->second];
  bool same = ndim == (int)t.shape.size();
  for (int d = 0; same && d < ndim; ++d) same = shape[d] == t.shape[d];
  if (!same) {
    std::string want, got;
    for (auto s : t.shape) want += std::to_string(s) + ",";
    for (int d = 0; d < ndim; ++d) got += std::to_string(shape[d]) + ",";
    return jv::fail(JV_ERR_SHAPE, std::string("size mismatch for ") + name + ": expected [" + want + "] got [" + got + "]");
  }
  if (c.ready[t.model]) return jv::fail(JV_ERR_STATE, "model already finalized; create a new context to reload");
  if (!t.dev) JV_TRY(c.raw_arena.alloc((size_t)t.numel, &t.dev));
  hipStream_t st = static_cast<hipStream_t>(stream);
  JV_HIP(hipMemcpyAsync(t.dev, data, sizeof(float) * t.numel, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
  if (!on_device) JV_HIP(hipStreamSynchronize(st));
  t.loaded = true;
  return JV_OK;
}

int jv_load_noise(jv_context* ctx, const float* data, int64_t numel, int on_device, void* stream) {
  CTX_GUARD(ctx);
  if (!data || numel != (int64_t)jv::N_FEATS * jv::NOISE_FRAMES)
    return jv::fail(JV_ERR_SHAPE, "jv_load_noise: expected 80*15000 floats");
  hipStream_t st = static_cast<hipStream_t>(stream);
  JV_HIP(hipMemcpyAsync(ctx->c.noise, data, sizeof(float) * numel, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                        st));
  if (!on_device) JV_HIP(hipStreamSynchronize(st));
  ctx->c.noise_loaded = true;
  return JV_OK;
}

int jv_finalize(jv_context* ctx, int model, void* stream) {
  CTX_GUARD(ctx);
  if (model != JV_MODEL_TTS && model != JV_MODEL_HIFT && model != JV_MODEL_PROMPT)
    return jv::fail(JV_ERR_ARG, "jv_finalize: unknown model id");
  if (ctx->c.ready[model]) return JV_OK;
  if (model == JV_MODEL_TTS) jv::flow_graphs_drop(ctx->c);
  return jv::finalize_model(ctx->c, model, static_cast<hipStream_t>(stream));
}

int jv_flow_estimator_step(jv_context* ctx, const float* x, const int32_t* lens, const float* mu, const float* t,
                           const float* spks, const float* cond, int B2, int T, float* out, void* stream) {
  CTX_GUARD(ctx);
  if (!x || !mu || !t || !spks || !cond || !out) return jv::fail(JV_ERR_ARG, "jv_flow_estimator_step: null tensor");
  return jv::flow_estimator(ctx->c, x, lens, mu, t, spks, cond, B2, T, out, static_cast<hipStream_t>(stream));
}

int jv_flow_estimator_masked(jv_context* ctx, const float* x, const float* mask, const float* mu, const float* t,
                             const float* spks, const float* cond, int B2, int T, float* out, void* stream) {
  CTX_GUARD(ctx);
  if (!x || !mask || !mu || !t || !spks || !cond || !out) return jv::fail(JV_ERR_ARG, "jv_flow_estimator_masked: null tensor");
  return jv::flow_estimator(ctx->c, x, nullptr, mu, t, spks, cond, B2, T, out, static_cast<hipStream_t>(stream), mask);
}

int jv_flow_set_streaming(jv_context* ctx, int chunk_frames) {
  CTX_GUARD(ctx);
  if (chunk_frames < 0) return jv::fail(JV_ERR_ARG, "jv_flow_set_streaming: chunk must be >= 0");
  ctx->c.attn_chunk = chunk_frames;
  return JV_OK;
}

int jv_flow_set_graph(jv_context* ctx, int on) {
  CTX_GUARD(ctx);
  ctx->c.step_graphs = on != 0;
  if (!on && jv::flow_has_graphs(ctx->c)) {
    JV_HIP(hipDeviceSynchronize());      // a replay may still be executing on the private stream
    jv::flow_graphs_drop(ctx->c);
  }
  return JV_OK;
}

// A mode switch is a configuration call, not a hot-path one: it waits for everything this device has queued (solves on
// any stream, graph replays) before it touches the workspace, so nothing in flight sees half of each mode.
int jv_flow_set_contraction(jv_context* ctx, int exact_range) {
  CTX_GUARD(ctx);
  if (ctx->c.exact_range != (exact_range != 0)) {
    JV_HIP(hipDeviceSynchronize());
    jv::flow_graphs_drop(ctx->c);                  // captured steps hold the old kernels
    jv::flow_ws_forget_attention(ctx->c, nullptr);   // that buffer holds fp32 rows in one mode, fp16 planes in the other
    JV_HIP(hipDeviceSynchronize());
  }
  ctx->c.exact_range = exact_range != 0;
  return JV_OK;
}

int jv_flow_contraction_info(const jv_context* ctx, int32_t* out, int n) {
  if (!ctx || !out || n < 4) return jv::fail(JV_ERR_ARG, "jv_flow_contraction_info: null argument or fewer than 4 slots");
  if (!ctx->c.ready[jv::MODEL_TTS]) return jv::fail(JV_ERR_STATE, "tts weights not finalized");
  int blocks = 0, all = 0, lin = 0, att = 0;
  for (int i = 0; i < jv::EST_NRES; ++i)
    for (int j = 0; j < jv::EST_NBLK; ++j) {
      const jv::BtbW& b = ctx->c.est.blk[i][j];
      const int l = (b.qkv.w2 && b.qkv.a_scale > 0.f) + (b.out.w2 && b.out.a_scale > 0.f) + (b.ff1.w2 && b.ff1.a_scale > 0.f) +
                    (b.ff2.w2 && b.ff2.a_scale > 0.f);
      const bool a = b.q_scale > 0.f && b.k_scale > 0.f && b.v_scale > 0.f;
      ++blocks; lin += l; att += a; all += (l == 4 && a);
    }
  out[0] = blocks; out[1] = all; out[2] = lin; out[3] = att;
  return JV_OK;
}

int jv_cfm_solve(jv_context* ctx, const float* mu, const int32_t* lens, const float* spks, const float* cond, int B, int T,
                 int n_timesteps, float temperature, const float* t_span_host, float* mel, void* stream) {
  CTX_GUARD(ctx);
  if (!mu || !spks || !cond || !mel) return jv::fail(JV_ERR_ARG, "jv_cfm_solve: null tensor");
  return jv::cfm_solve(ctx->c, mu, lens, spks, cond, B, T, n_timesteps, temperature, t_span_host, mel,
                       static_cast<hipStream_t>(stream));
}

// ---- operator-level entry points -------------------------------------------------------------------
int jv_op_conv_gemm(const float* A, int64_t a_rows, int M, int Cin, int ntaps, int tap_row0, int dil, const float* W, int N,
                    const float* bias, int act, int prologue, const float* alpha, float slope, const float* ln_g,
                    const float* ln_b, float ln_eps, const uint8_t* rowmask, const float* res, float* out, void* stream) {
  static bool inited = false;
  if (!inited) {
    JV_TRY(jv::conv_gemm_init());
    inited = true;
  }
  jv::ConvGemmArgs a;
  jv::conv_gemm_defaults(a);
  a.A = A; a.lda = Cin; a.a_rows = a_rows; a.M = M; a.Cin = Cin; a.ntaps = ntaps; a.tap_row0 = tap_row0; a.tap_dil = dil;
  a.W = W; a.ldw = ntaps * Cin; a.n_rows_w = N; a.N = N; a.bias = bias; a.out = out; a.ldo = N;
  a.act = act; a.pro = prologue; a.pro_alpha = alpha; a.pro_slope = slope;
  if (ln_g) { a.ln = 1; a.ln_g = ln_g; a.ln_b = ln_b; a.ln_eps = ln_eps; }
  a.rowmask_in = rowmask; a.rowmask_out = rowmask;
  a.res1 = res; a.ldr1 = N;
  if (jv::dyn_env("JV_OP_X6")) {   // exercise the bf16x6 main loop: split W into three bf16 planes on the fly
    unsigned short* scratch = nullptr;
    const size_t n = (size_t)N * ntaps * Cin;
    JV_TRY(op_scratch(OPS_X6, 3 * n * sizeof(unsigned short) + 64, reinterpret_cast<void**>(&scratch)));
    JV_TRY(jv::split3_planes(W, scratch, (long)n, static_cast<hipStream_t>(stream)));
    a.W3 = scratch;
    a.w3_plane = (long)n;
  }
  return jv::conv_gemm(a, 1, static_cast<hipStream_t>(stream));
}

// host logic of the fp16x3 scale choice (no device work): largest power of two s with bound * s <= 60000, 0 = unusable
float jv_h3_scale_for_bound(float bound) { return jv::h3_scale_for_bound(bound); }

// y = act(A W^T + bias) (+ res) through the fp16x3 main loop; a_bound: the caller's proven bound on |A| (test hook)
int jv_op_linear_h3(const float* A, int64_t rows, int M, int K, const float* W, int N, const float* bias, int act,
                    const float* res, float a_bound, int presplit, float* out, void* stream) {
  static bool inited = false;
  if (!inited) {
    JV_TRY(jv::conv_gemm_init());
    inited = true;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (K % 32 || N <= 0 || M <= 0) return jv::fail(JV_ERR_ARG, "jv_op_linear_h3: K must be a multiple of 32");
  const float sc = jv::h3_scale_for_bound(a_bound);
  if (!(sc > 0.f)) return jv::fail(JV_ERR_ARG, "jv_op_linear_h3: unusable bound");
  void* scratch = nullptr;
  const size_t n = (size_t)N * K;
  const size_t need = n * 8 + (size_t)N * 12 + 512;
  JV_TRY(op_scratch(OPS_H3, need, &scratch));
  unsigned short* planes = static_cast<unsigned short*>(scratch);
  float* cs = reinterpret_cast<float*>(static_cast<char*>(scratch) + ((n * 4 + 63) & ~(size_t)63));
  JV_TRY(jv::split2h_planes(W, N, K, cs + N, planes, cs, st));
  jv::ConvGemmArgs a;
  jv::conv_gemm_defaults(a);
  a.A = A; a.lda = K; a.a_rows = rows; a.M = M; a.Cin = K; a.ntaps = 1; a.tap_row0 = 0; a.tap_dil = 1;
  a.W = W; a.ldw = K; a.n_rows_w = N; a.N = N; a.bias = bias; a.out = out; a.ldo = N; a.act = act;
  a.res1 = res; a.ldr1 = N;
  a.W2 = planes; a.w2_plane = (long)n; a.colscale = cs; a.a_scale = sc;
  if (presplit) {      // A as fp16 planes written by a producer kernel; 2 = reuse the planes of the previous call (timing)
    unsigned short* ap = nullptr;
    const size_t an = (size_t)rows * K;
    JV_TRY(op_scratch(OPS_H3_A, an * 4, reinterpret_cast<void**>(&ap)));
    if (presplit == 1) JV_TRY(jv::split2h_rows(A, K, ap, (long)an, rows, K, sc, st));
    a.A2 = ap; a.a2_plane = (long)an; a.lda2 = K;
  }
  return jv::conv_gemm(a, 1, st);
}

int jv_op_attention(const float* qkv, const int32_t* lens, int B, int G, int S, int L, float* out, void* stream) {
  jv::AttnArgs at{};
  at.qkv = qkv; at.ld = 1536; at.k_off = 512; at.v_off = 1024; at.out = out; at.ldo = 512;
  at.B = B; at.H = 8; at.G = G; at.S = S; at.L = L; at.lens = lens; at.chunk = 0;
  return jv::attention64(at, static_cast<hipStream_t>(stream));
}

// conv_gemm through the fp16x3 main loop with a MEASURED bound: amax_in = device float >= max |A| (e.g. the amax_out of
// the launch that produced A), a_extra = what the prologue can add; amax_out (optional) receives max |out| (test hook)
int jv_op_conv_h3_measured(const float* A, int64_t a_rows, int M, int Cin, int ntaps, int tap_row0, int dil, const float* W,
                           int N, const float* bias, int act, int prologue, const float* alpha, float slope,
                           const uint8_t* rowmask, const float* res, const float* amax_in, float a_extra, float* amax_out,
                           float* out, void* stream) {
  static bool inited = false;
  if (!inited) {
    JV_TRY(jv::conv_gemm_init());
    inited = true;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t n = (size_t)N * ntaps * Cin;
  void* scratch = nullptr;
  const size_t need = n * 10 + (size_t)N * 12 + 512;      // 2 fp16 + 3 bf16 planes, colscale, row stats
  JV_TRY(op_scratch(OPS_CONV, need, &scratch));
  unsigned short* planes2 = static_cast<unsigned short*>(scratch);
  unsigned short* planes3 = planes2 + 2 * n;
  float* cs = reinterpret_cast<float*>(static_cast<char*>(scratch) + ((n * 10 + 63) & ~(size_t)63));
  jv::ConvGemmArgs a;
  jv::conv_gemm_defaults(a);
  a.A = A; a.lda = Cin; a.a_rows = a_rows; a.M = M; a.Cin = Cin; a.ntaps = ntaps; a.tap_row0 = tap_row0; a.tap_dil = dil;
  a.W = W; a.ldw = ntaps * Cin; a.n_rows_w = N; a.N = N; a.bias = bias; a.out = out; a.ldo = N;
  a.act = act; a.pro = prologue; a.pro_alpha = alpha; a.pro_slope = slope;
  a.rowmask_in = rowmask; a.rowmask_out = rowmask;
  a.res1 = res; a.ldr1 = N;
  a.amax_out = amax_out;
  if (amax_in) {
    JV_TRY(jv::split2h_planes(W, N, ntaps * Cin, cs + N, planes2, cs, st));
    a.W2 = planes2; a.w2_plane = (long)n; a.colscale = cs; a.amax_in = amax_in; a.a_extra = a_extra;
  } else {      // producer only: bf16x6
    JV_TRY(jv::split3_planes(W, planes3, (long)n, st));
    a.W3 = planes3; a.w3_plane = (long)n;
  }
  return jv::conv_gemm(a, 1, st);
}

// the fp16x3 attention kernel; q_bound, k_bound, v_bound: the caller's proven bounds on |q|, |k|, |v| (test hook)
int jv_op_attention_h3(const float* qkv, const int32_t* lens, int B, int G, int S, int L, float q_bound, float k_bound,
                       float v_bound, float* out, void* stream) {
  jv::AttnArgs at{};
  at.qkv = qkv; at.ld = 1536; at.k_off = 512; at.v_off = 1024; at.out = out; at.ldo = 512;
  at.B = B; at.H = 8; at.G = G; at.S = S; at.L = L; at.lens = lens; at.chunk = 0;
  at.q_scale = jv::h3_scale_for_bound(q_bound * 0.1803369f);
  at.k_scale = jv::h3_scale_for_bound(k_bound);
  at.v_scale = jv::h3_scale_for_bound(v_bound);
  if (!(at.q_scale > 0.f && at.k_scale > 0.f && at.v_scale > 0.f)) return jv::fail(JV_ERR_ARG, "jv_op_attention_h3: unusable bound");
  return jv::attention64(at, static_cast<hipStream_t>(stream));
}

// the row-owning fp16x3 GEMM (rowgemm_kernel.h) with each of its epilogues (test / tuning hook): A [rows,K] fp32 is split
// into planes here (presplit = 2: reuse the planes of the previous call, timing only); epi: 0 plain, 1 GELU -> planes,
// 2 + res, 3 + res -> LayerNorm -> planes.  out: fp32 [M,N] (epi 0, 2, 3); out2: planes [2][M][N] (epi 1) / [2][M][256] (epi 3)
int jv_op_rowgemm(const float* A, int64_t rows, int M, int K, const float* W, int N, const float* bias, int epi,
                  const float* res, const float* ln_g, const float* ln_b, float a_bound, float out2_scale, int presplit,
                  float* out, uint16_t* out2, float* amax_out, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float sc = jv::h3_scale_for_bound(a_bound);
  if (!(sc > 0.f)) return jv::fail(JV_ERR_ARG, "jv_op_rowgemm: unusable bound");
  void* scratch = nullptr;
  const size_t n = (size_t)N * K;
  const size_t need = n * 8 + (size_t)N * 12 + 512;
  JV_TRY(op_scratch(OPS_RG, need, &scratch));
  unsigned short* planes = static_cast<unsigned short*>(scratch);
  float* cs = reinterpret_cast<float*>(static_cast<char*>(scratch) + ((n * 4 + 63) & ~(size_t)63));
  JV_TRY(jv::split2h_planes(W, N, K, cs + N, planes, cs, st));
  unsigned short* wf = reinterpret_cast<unsigned short*>(static_cast<char*>(scratch) + ((n * 4 + (size_t)N * 12 + 511) & ~(size_t)255));
  if (!(K & 63)) JV_TRY(jv::pack_wfrag(planes, (long)n, K, N, K, wf, (long)n, st));
  unsigned short* ap = nullptr;
  const size_t an = (size_t)rows * K;
  JV_TRY(op_scratch(OPS_RG_A, an * 4, reinterpret_cast<void**>(&ap)));
  if (presplit != 2) JV_TRY(jv::split2h_rows(A, K, ap, (long)an, rows, K, sc, st));
  jv::RowGemmArgs a{};
  a.A2 = ap; a.a2_plane = (long)an; a.a_rows = rows; a.lda2 = K;
  a.M = M; a.K = K; a.N = N;
  a.W2 = planes; a.w2_plane = (long)n; a.ldw = K; a.colscale = cs; a.a_scale = sc; a.bias = bias;
  if (!(K & 63)) { a.Wf = wf; a.wf_plane = (long)n; }
  a.out = out; a.ldo = N; a.res = res; a.ldr = N;
  a.out2 = out2; a.out2_plane = (long)M * (epi == jv::RG_RES_LN ? 256 : N); a.ldo2 = epi == jv::RG_RES_LN ? 256 : N;
  a.out2_scale = out2_scale;
  a.ln_g = ln_g; a.ln_b = ln_b; a.ln_eps = 1e-5f;
  a.amax_out = amax_out;
  return jv::rowgemm(a, epi, st);
}

// the row-owning causal k = 3 convolution (rowconv_kernel.h) with its fused row-wise tail (test / tuning hook): one
// measured-bound slot (amax_in, a device float >= max |A|), W [256, 3 Cin] packed tap-major like jv_op_conv_gemm
int jv_op_rowconv(const float* A, int64_t rows, int M, int Cin, const float* W, const float* bias, const float* ln_g,
                  const float* ln_b, int act, const uint8_t* rowmask, const float* rowvec, const float* res,
                  const float* amax_in, float* amax_out, float* out, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int N = 256, K = 3 * Cin;
  void* scratch = nullptr;
  const size_t n = (size_t)N * K;
  const size_t head = ((n * 4 + (size_t)N * 12 + 256 + (size_t)rows * sizeof(int)) + 255) & ~(size_t)255;
  const size_t need = head + n * 4;      // + the planes again in fragment order
  JV_TRY(op_scratch(OPS_RC, need, &scratch));
  unsigned short* planes = static_cast<unsigned short*>(scratch);
  float* cs = reinterpret_cast<float*>(static_cast<char*>(scratch) + ((n * 4 + 63) & ~(size_t)63));
  int* slots = reinterpret_cast<int*>(cs + 3 * N + 16);
  unsigned short* wf = reinterpret_cast<unsigned short*>(static_cast<char*>(scratch) + head);
  JV_TRY(jv::split2h_planes(W, N, K, cs + N, planes, cs, st));
  const bool frag = !((Cin >> 5) & 1) && !(Cin & 31);
  if (frag) JV_TRY(jv::pack_wfrag(planes, (long)n, K, N, K, wf, (long)n, st));
  JV_HIP(hipMemsetAsync(slots, 0, (size_t)rows * sizeof(int), st));
  jv::RowConvArgs a{};
  a.A = A; a.lda = Cin; a.a_rows = rows; a.M = M; a.Cin = Cin; a.rowmask_in = rowmask;
  a.W2 = planes; a.w2_plane = (long)n; a.ldw = K; a.colscale = cs; a.amax_in = amax_in; a.row_slot = slots; a.bias = bias;
  if (frag) { a.Wf = wf; a.wf_plane = (long)n; }
  a.out = out; a.ldo = N;
  if (ln_g) { a.ln = 1; a.ln_g = ln_g; a.ln_b = ln_b; a.ln_eps = 1e-5f; }
  a.act = act; a.rowmask_out = rowmask; a.rowvec = rowvec; a.rowvec_ld = N; a.res = res; a.ldr = N;
  a.amax_out = amax_out; a.row_mask = rowmask;
  return jv::rowconv(a, st);
}

// hiftconv (hiftconv_kernel.h): out = ((Conv1d_k,dil(Snake_alpha(A)) + bias) + res1 + res2) * out_scale (+ out when accumulate) on a
// [rows, C] row buffer, C = 64 / 128 / 256; W [C][ntaps * C] tap-major; amax_in / amax_out: one slot (test hook)
int jv_op_hiftconv(const float* A, int64_t rows, int C, int ntaps, int dil, const float* W, const float* bias, const float* alpha,
                   const uint8_t* rowmask, const float* res1, const float* res2, float out_scale, int accumulate,
                   const float* amax_in, float a_extra, float* amax_out, float* out, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int K = ntaps * C;
  void* scratch = nullptr;
  const size_t n = (size_t)C * K;
  const size_t head = ((n * 4 + (size_t)C * 12 + 256) + 255) & ~(size_t)255;
  const size_t need = head + n * 4;
  JV_TRY(op_scratch(OPS_HIFT, need, &scratch));
  unsigned short* planes = static_cast<unsigned short*>(scratch);
  float* cs = reinterpret_cast<float*>(static_cast<char*>(scratch) + ((n * 4 + 63) & ~(size_t)63));
  unsigned short* wf = reinterpret_cast<unsigned short*>(static_cast<char*>(scratch) + head);
  JV_TRY(jv::split2h_planes(W, C, K, cs + C, planes, cs, st));
  JV_TRY(jv::pack_wfrag(planes, (long)n, K, C, K, wf, (long)n, st));
  jv::HiftConvArgs a{};
  a.A = A; a.a_rows = rows; a.M = (int)rows; a.ntaps = ntaps; a.dil = dil; a.tap_row0 = -(dil * (ntaps - 1) / 2); a.rowmask_in = rowmask;
  a.alpha = alpha; a.Wf = wf; a.wf_plane = (long)n; a.colscale = cs; a.bias = bias;
  a.amax_in = amax_in; a.a_extra = a_extra; a.slot_G = 0; a.slot_S = 0; a.slot_nb = 1;
  a.out = out; a.res1 = res1; a.res2 = res2; a.out_scale = out_scale; a.accumulate = accumulate;
  a.amax_out = amax_out; a.amax_mask = rowmask;
  return jv::hiftconv(a, C, st);
}

// attention64_planes (attention_pl.hip) on an fp32 qkv matrix: K and V are split into planes here the way the qkv GEMM's
// epilogue writes them; planes_out = 1: the result as fp16 planes [2][rows][512] of value * out2_scale (test hook)
int jv_op_attention_planes(const float* qkv, int64_t rows, const int32_t* lens, int B, int G, int S, int L, float q_bound,
                           float k_bound, float v_bound, int chunk, float out2_scale, float* out, uint16_t* out2, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  unsigned short* kv = nullptr;
  JV_TRY(op_scratch(OPS_ATTN, (size_t)rows * 1024 * 2 * sizeof(unsigned short), reinterpret_cast<void**>(&kv)));
  jv::AttnArgs at{};
  at.qkv = qkv; at.ld = 1536; at.out = out; at.ldo = 512;
  at.B = B; at.H = 8; at.G = G; at.S = S; at.L = L; at.lens = lens; at.chunk = chunk;
  at.q_scale = jv::h3_scale_for_bound(q_bound * 0.1803369f);
  at.k_scale = jv::h3_scale_for_bound(k_bound);
  at.v_scale = jv::h3_scale_for_bound(v_bound);
  if (!(at.q_scale > 0.f && at.k_scale > 0.f && at.v_scale > 0.f)) return jv::fail(JV_ERR_ARG, "jv_op_attention_planes: unusable bound");
  const long plane = (long)rows * 1024;
  JV_TRY(jv::split2h_rows(qkv + 512, 1536, kv, plane, rows, 512, at.k_scale, st, 1024));
  JV_TRY(jv::split2h_rows(qkv + 1024, 1536, kv + 512, plane, rows, 512, at.v_scale, st, 1024));
  at.kv2 = kv; at.kv2_plane = plane; at.kv_ld = 1024;
  if (out2) { at.out2 = out2; at.out2_plane = (long)rows * 512; at.out2_scale = out2_scale; }
  if (chunk == 0 && jv::dyn_env("JV_OP_ATTN_ROWS")) return jv::attention64_rows(at, st);      // the row-owning form (attention_r.hip)
  if (chunk == 0 && jv::dyn_env("JV_OP_ATTN_SINGLE")) return jv::attention64_single(at, st);  // one wave per SIMD (attention_s.hip)
  return jv::attention64_planes(at, st);
}

int jv_op_layernorm(const float* x, const float* g, const float* b, float eps, int64_t rows, int C, float* out, void* stream) {
  return jv::layernorm_rows(x, nullptr, out, g, b, eps, rows, C, nullptr, static_cast<hipStream_t>(stream));
}

}  // extern "C"
