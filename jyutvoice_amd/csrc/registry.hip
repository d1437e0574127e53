// Weight registry (checkpoint tensor names/shapes the library accepts) and load-time packing.
//
// Names and shapes are exactly the reference's state-dict keys:
//   tts : encoder.* (117), dp.* (12), decoder.estimator.* (910), spk_embed_affine_layer.* (2)
//         -- jyutvoice/models/jyutvoice_tts.py:42-49, configs/base.yaml:50-110
//   hift: 328 keys, weight-norm as parametrizations.weight.original0/1 (generator.py:26) and
//         weight_g/weight_v (f0_predictor.py:16)
// finalize_model() folds weight-norm (w = v * g/||v||, torch._weight_norm), re-lays convolution
// kernels [Cout][Cin][k] as K-contiguous GEMM operands [Cout][k*Cin] (tap-major), fuses q/k/v and the
// 14 per-resnet time projections into single matrices, and rewrites each ConvTranspose1d as a
// polyphase matrix [stride*Cout][3*Cin].
#include <math.h>

#include "jv_model.h"

namespace jv {

// ------------------------------------------------------------------------------------------------
int Arena::alloc(size_t floats, float** out) {
  floats = (floats + 63) & ~(size_t)63;
  if (!cur || used + floats > cap) {
    const size_t n = floats > chunk_floats ? floats : chunk_floats;
    void* p = nullptr;
    JV_HIP(hipMalloc(&p, n * sizeof(float)));
    chunks.push_back(p);
    cur = static_cast<float*>(p);
    cap = n;
    used = 0;
  }
  *out = cur + used;
  used += floats;
  return JV_OK;
}

void Arena::release() {
  for (void* p : chunks) (void)hipFree(p);
  chunks.clear();
  cur = nullptr;
  used = cap = 0;
}

// ------------------------------------------------------------------------------------------------
namespace {

struct Reg {
  Context& c;
  int model;
  void add(const std::string& name, std::vector<int64_t> shape) {
    RawTensor t;
    t.name = name;
    t.shape = std::move(shape);
    t.numel = 1;
    for (auto s : t.shape) t.numel *= s;
    t.model = model;
    c.index[name] = (int)c.raw.size();
    c.raw.push_back(std::move(t));
  }
  void wb(const std::string& p, std::vector<int64_t> wshape) {   // weight + bias
    const int64_t n = wshape[0];
    add(p + "weight", std::move(wshape));
    add(p + "bias", {n});
  }
  void gb(const std::string& p, int64_t n, const char* g = "gamma", const char* b = "beta") {
    add(p + g, {n});
    add(p + b, {n});
  }
};

std::string S(int i) { return std::to_string(i); }

}  // namespace

void build_registry(Context& c) {
  Reg r{c, MODEL_TTS};
  // ---- encoder.* (text_encoder.py:340-404) -----------------------------------------------------
  const std::string e = "encoder.";
  r.add(e + "emb.weight", {97, ENC_CH});
  r.add(e + "lang_emb.weight", {4, ENC_CH});
  r.add(e + "tone_emb.weight", {7, ENC_CH});
  r.add(e + "word_pos_emb.weight", {4, ENC_CH});
  r.add(e + "syllable_pos.weight", {4, ENC_CH});
  for (int i = 0; i < 3; ++i) {
    r.wb(e + "prenet.conv_layers." + S(i) + ".", {ENC_CH, ENC_CH, 5});
    r.gb(e + "prenet.norm_layers." + S(i) + ".", ENC_CH);
  }
  r.wb(e + "prenet.proj.", {ENC_CH, ENC_CH, 1});
  for (int i = 0; i < ENC_LAYERS; ++i) {
    for (const char* n : {"q", "k", "v", "o"})
      r.wb(e + "encoder.attn_layers." + S(i) + ".conv_" + n + ".", {ENC_HID, ENC_HID, 1});
    r.gb(e + "encoder.norm_layers_1." + S(i) + ".", ENC_HID);
    r.wb(e + "encoder.ffn_layers." + S(i) + ".conv_1.", {ENC_FILTER, ENC_HID, 3});
    r.wb(e + "encoder.ffn_layers." + S(i) + ".conv_2.", {ENC_HID, ENC_FILTER, 3});
    r.gb(e + "encoder.norm_layers_2." + S(i) + ".", ENC_HID);
  }
  r.wb(e + "proj.", {N_FEATS, ENC_HID, 1});
  // ---- dp.* (duration_predictor.py:26-46) --------------------------------------------------------
  r.wb("dp.conv_1.", {DP_FILTER, ENC_HID, 3});
  r.gb("dp.norm_1.", DP_FILTER);
  r.wb("dp.conv_2.", {DP_FILTER, DP_FILTER, 3});
  r.gb("dp.norm_2.", DP_FILTER);
  r.wb("dp.proj.", {1, DP_FILTER, 1});
  r.wb("dp.cond.", {ENC_HID, SPK_DIM, 1});
  // ---- decoder.estimator.* (decoder.py:798-915) ----------------------------------------------------
  const std::string p = "decoder.estimator.";
  r.wb(p + "time_mlp.linear_1.", {EST_TIME, EST_IN});
  r.wb(p + "time_mlp.linear_2.", {EST_TIME, EST_TIME});
  auto resnet = [&](const std::string& q, int cin) {
    r.wb(q + "mlp.1.", {EST_CH, EST_TIME});
    r.wb(q + "block1.block.0.", {EST_CH, cin, 3});
    r.gb(q + "block1.block.2.", EST_CH, "weight", "bias");
    r.wb(q + "block2.block.0.", {EST_CH, EST_CH, 3});
    r.gb(q + "block2.block.2.", EST_CH, "weight", "bias");
    r.wb(q + "res_conv.", {EST_CH, cin, 1});
  };
  auto btb = [&](const std::string& q) {
    r.gb(q + "norm1.", EST_CH, "weight", "bias");
    r.add(q + "attn1.to_q.weight", {EST_INNER, EST_CH});
    r.add(q + "attn1.to_k.weight", {EST_INNER, EST_CH});
    r.add(q + "attn1.to_v.weight", {EST_INNER, EST_CH});
    r.wb(q + "attn1.to_out.0.", {EST_CH, EST_INNER});
    r.gb(q + "norm3.", EST_CH, "weight", "bias");
    r.wb(q + "ff.net.0.proj.", {EST_FF, EST_CH});
    r.wb(q + "ff.net.2.", {EST_CH, EST_FF});
  };
  resnet(p + "down_blocks.0.0.", EST_IN);
  for (int j = 0; j < EST_NBLK; ++j) btb(p + "down_blocks.0.1." + S(j) + ".");
  r.wb(p + "down_blocks.0.2.", {EST_CH, EST_CH, 3});
  for (int i = 0; i < EST_NMID; ++i) {
    resnet(p + "mid_blocks." + S(i) + ".0.", EST_CH);
    for (int j = 0; j < EST_NBLK; ++j) btb(p + "mid_blocks." + S(i) + ".1." + S(j) + ".");
  }
  resnet(p + "up_blocks.0.0.", 2 * EST_CH);
  for (int j = 0; j < EST_NBLK; ++j) btb(p + "up_blocks.0.1." + S(j) + ".");
  r.wb(p + "up_blocks.0.2.", {EST_CH, EST_CH, 3});
  r.wb(p + "final_block.block.0.", {EST_CH, EST_CH, 3});
  r.gb(p + "final_block.block.2.", EST_CH, "weight", "bias");
  r.wb(p + "final_proj.", {N_FEATS, EST_CH, 1});
  r.wb("spk_embed_affine_layer.", {N_FEATS, SPK_DIM});

  // ---- hift (generator.py:245-355, f0_predictor.py:8-50) ----------------------------------------------
  Reg h{c, MODEL_HIFT};
  auto wn = [&](const std::string& q, std::vector<int64_t> shape, int64_t nbias) {
    h.add(q + "bias", {nbias});
    h.add(q + "parametrizations.weight.original0", {shape[0], 1, 1});
    h.add(q + "parametrizations.weight.original1", std::move(shape));
  };
  h.wb("m_source.l_linear.", {1, HIFT_HARM});
  wn("conv_pre.", {HIFT_CH, N_FEATS, 7}, HIFT_CH);
  const int up_k[3] = {16, 11, 7};
  for (int i = 0; i < 3; ++i) wn("ups." + S(i) + ".", {HIFT_CH >> i, HIFT_CH >> (i + 1), up_k[i]}, HIFT_CH >> (i + 1));
  const int sd_k[3] = {30, 6, 1};
  for (int i = 0; i < 3; ++i) h.wb("source_downs." + S(i) + ".", {HIFT_CH >> (i + 1), HIFT_NFFT + 2, sd_k[i]});
  auto resblock = [&](const std::string& q, int ch, int k) {
    for (const char* which : {"convs1.", "convs2."})
      for (int j = 0; j < 3; ++j) wn(q + which + S(j) + ".", {ch, ch, k}, ch);
    for (const char* which : {"activations1.", "activations2."})
      for (int j = 0; j < 3; ++j) h.add(q + which + S(j) + ".alpha", {ch});
  };
  const int src_k[3] = {7, 7, 11}, rb_k[3] = {3, 7, 11};
  for (int i = 0; i < 3; ++i) resblock("source_resblocks." + S(i) + ".", HIFT_CH >> (i + 1), src_k[i]);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) resblock("resblocks." + S(3 * i + j) + ".", HIFT_CH >> (i + 1), rb_k[j]);
  wn("conv_post.", {HIFT_NFFT + 2, HIFT_CH >> 3, 7}, HIFT_NFFT + 2);
  for (int n = 0; n < 5; ++n) {
    const std::string q = "f0_predictor.condnet." + S(2 * n) + ".";
    h.add(q + "bias", {HIFT_F0_CH});
    h.add(q + "weight_g", {HIFT_F0_CH, 1, 1});
    h.add(q + "weight_v", {HIFT_F0_CH, n == 0 ? N_FEATS : HIFT_F0_CH, 3});
  }
  h.wb("f0_predictor.classifier.", {1, HIFT_F0_CH});

  // ---- prompt encoder (infer.py:35-83; upsample_encoder.py:137-300) -----------------------------------
  Reg q{c, MODEL_PROMPT};
  q.add("input_embedding.weight", {PR_VOCAB, PR_DIM});
  auto embed = [&](const std::string& s) {
    q.wb(s + "out.0.", {PR_DIM, PR_DIM});
    q.gb(s + "out.1.", PR_DIM, "weight", "bias");
  };
  auto block = [&](const std::string& s) {
    q.add(s + "self_attn.pos_bias_u", {PR_HEADS, PR_DIM / PR_HEADS});
    q.add(s + "self_attn.pos_bias_v", {PR_HEADS, PR_DIM / PR_HEADS});
    for (const char* n : {"q", "k", "v", "out"}) q.wb(s + "self_attn.linear_" + n + ".", {PR_DIM, PR_DIM});
    q.add(s + "self_attn.linear_pos.weight", {PR_DIM, PR_DIM});
    q.wb(s + "feed_forward.w_1.", {PR_FFN, PR_DIM});
    q.wb(s + "feed_forward.w_2.", {PR_DIM, PR_FFN});
    q.gb(s + "norm_ff.", PR_DIM, "weight", "bias");
    q.gb(s + "norm_mha.", PR_DIM, "weight", "bias");
  };
  const std::string pe = "encoder.";
  embed(pe + "embed.");
  q.gb(pe + "after_norm.", PR_DIM, "weight", "bias");
  q.wb(pe + "pre_lookahead_layer.conv1.", {PR_DIM, PR_DIM, 4});
  q.wb(pe + "pre_lookahead_layer.conv2.", {PR_DIM, PR_DIM, 3});
  for (int i = 0; i < PR_BLOCKS; ++i) block(pe + "encoders." + S(i) + ".");
  q.wb(pe + "up_layer.conv.", {PR_DIM, PR_DIM, 5});
  embed(pe + "up_embed.");
  for (int i = 0; i < PR_UP_BLOCKS; ++i) block(pe + "up_encoders." + S(i) + ".");
  q.wb("encoder_proj.", {N_FEATS, PR_DIM});
  // not a state-dict entry: the positional encoding's 256 frequencies, supplied by the host mirror
  q.add("pos_enc.div_term", {PR_DIM / 2});
}

// ------------------------------------------------------------------------------------------------
namespace {

// dst[(row0+n)*ldw + j*cinp + ci] = src[n][ci][j]   (zero for ci >= cin)
__global__ void pack_conv_kernel(const float* __restrict__ src, float* __restrict__ dst, int cout, int cin, int k, int cinp,
                                 int ldw, int row0) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long per_n = (long)k * cinp;
  if (idx >= per_n * cout) return;
  const int n = (int)(idx / per_n);
  const int rem = (int)(idx - (long)n * per_n);
  const int j = rem / cinp, ci = rem - j * cinp;
  dst[(long)(row0 + n) * ldw + rem] = ci < cin ? src[((long)n * cin + ci) * k + j] : 0.f;
}

// ConvTranspose1d [cin][cout][k], stride s, padding pad -> polyphase rows n = r*cout + co, 3 taps (rows q-1, q, q+1):
//   tap jj reads input row q + jj - 1 = q - m with m = 1 - jj and kernel index kk = r + pad + m*s
__global__ void pack_convT_kernel(const float* __restrict__ src, float* __restrict__ dst, int cin, int cout, int k, int s,
                                  int pad) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long per_n = 3L * cin;
  const long total = per_n * s * cout;
  if (idx >= total) return;
  const int n = (int)(idx / per_n);
  const int rem = (int)(idx - (long)n * per_n);
  const int jj = rem / cin, ci = rem - jj * cin;
  const int r = n / cout, co = n - r * cout;
  const int kk = r + pad + (1 - jj) * s;
  dst[idx] = (kk >= 0 && kk < k) ? src[((long)ci * cout + co) * k + kk] : 0.f;
}

__global__ void tile_bias_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int reps) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n * reps) dst[i] = src[i % n];
}

// fp32 -> three bf16 planes (h, m, l) with x ~= h + m + l to 24 bits (conv_gemm_x6_kernel.h)
__global__ void split3_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = src[i];
  const __bf16 h = (__bf16)x;
  const float r = x - (float)h;
  const __bf16 m = (__bf16)r;
  const __bf16 l = (__bf16)(r - (float)m);
  dst[i] = __builtin_bit_cast(unsigned short, h);
  dst[n + i] = __builtin_bit_cast(unsigned short, m);
  dst[2 * n + i] = __builtin_bit_cast(unsigned short, l);
}

// per row n of W [rows][ld]: amax[n] = max |w|, l1[n] = sum |w| over the first K columns (one wave per row)
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ w, int rows, int ld, int K,
                                                        float* __restrict__ amax, float* __restrict__ l1) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= rows) return;
  float m = 0.f, s = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float a = fabsf(w[(long)n * ld + k]);
    m = fmaxf(m, a);      // NaN-free weights assumed; a NaN sum below disables the layer's fp16 path on the host
    s += a;
  }
  for (int o = 32; o > 0; o >>= 1) {
    m = fmaxf(m, __shfl_xor(m, o));
    s += __shfl_xor(s, o);
  }
  if (lane == 0) { amax[n] = m; l1[n] = s; }
}

// fp32 -> two fp16 planes of w * 2^e_n with max_k |w[n][k]| * 2^e_n in [2^13, 2^14); colscale[n] = 2^-e_n
__global__ void split2h_kernel(const float* __restrict__ src, const float* __restrict__ amax, unsigned short* __restrict__ dst,
                               float* __restrict__ colscale, long total, int ld) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int n = (int)(i / ld);
  int ex = 0;
  const float am = amax[n];
  float sc = 1.f;
  if (am > 0.f && am < INFINITY) {
    (void)frexpf(am, &ex);                       // am = f * 2^ex, f in [0.5, 1)
    sc = ldexpf(1.f, min(max(14 - ex, -60), 60));
  }
  const float x = src[i] * sc;
  const _Float16 h = (_Float16)x;
  const _Float16 l = (_Float16)(x - (float)h);
  dst[i] = __builtin_bit_cast(unsigned short, h);
  dst[total + i] = __builtin_bit_cast(unsigned short, l);
  if (i == (long)n * ld) colscale[n] = 1.0f / sc;
}

// w[r][:] = v[r][:] * (g[r] / ||v[r][:]||)
__global__ __launch_bounds__(256) void fold_wn_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                      float* __restrict__ w, int cols) {
  __shared__ float red[4];
  const int r = blockIdx.x;
  const float* vr = v + (long)r * cols;
  float s = 0.f;
  for (int i = threadIdx.x; i < cols; i += 256) s += vr[i] * vr[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float tot = (red[0] + red[1]) + (red[2] + red[3]);
  const float scale = g[r] / sqrtf(tot);
  for (int i = threadIdx.x; i < cols; i += 256) w[(long)r * cols + i] = vr[i] * scale;
}

}  // namespace

// test hook: fp16x3 planes + column scales of an arbitrary [rows][ld] matrix (jv_op_linear_h3); stats = 2 * rows floats
int split2h_planes(const float* src, int rows, int ld, float* stats, unsigned short* dst, float* colscale, hipStream_t st) {
  hipLaunchKernelGGL(row_stats_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, src, rows, ld, ld, stats, stats + rows);
  const long total = (long)rows * ld;
  hipLaunchKernelGGL(split2h_kernel, dim3((unsigned)cdivl(total, 256)), dim3(256), 0, st, src, stats, dst, colscale, total, ld);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// largest power of two s with bound * s <= 60000 (fp16 max 65504 less a margin for the fp32 roundings on the way), 0 if
// the bound is unusable
float h3_scale_for_bound(float bound) {
  if (!(bound > 0.f) || !(bound < 1e30f)) return 0.f;
  int e = (int)floorf(log2f(60000.f / bound));
  if (e < -24) return 0.f;
  if (e > 24) e = 24;
  float s = ldexpf(1.f, e);
  while (bound * s > 60000.f) s *= 0.5f;
  return s;
}

// test hook: split an arbitrary [n] fp32 matrix into bf16x6 planes (used by jv_op_conv_gemm)
int split3_planes(const float* src, unsigned short* dst, long n, hipStream_t st) {
  hipLaunchKernelGGL(split3_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, st, src, dst, n);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

namespace {

struct Packer {
  Context& c;
  hipStream_t st;
  int rc = JV_OK;

  const RawTensor* get(const std::string& name) {
    auto it = c.index.find(name);
    if (it == c.index.end()) { rc = fail(JV_ERR_NAME, "internal: unknown tensor " + name); return nullptr; }
    const RawTensor& t = c.raw[it->second];
    if (!t.loaded) { rc = fail(JV_ERR_STATE, "missing tensor: " + name); return nullptr; }
    return &t;
  }
  const float* ptr(const std::string& name) {
    const RawTensor* t = get(name);
    return t ? t->dev : nullptr;
  }
  float* alloc(size_t n) {
    float* p = nullptr;
    if (rc == JV_OK) rc = c.packed.alloc(n, &p);
    return p;
  }
  // plain weight of a weight-normalised layer: fold (g, v) into a scratch matrix of the packed arena
  const float* folded(const std::string& gname, const std::string& vname) {
    const RawTensor* g = get(gname);
    const RawTensor* v = get(vname);
    if (!g || !v) return nullptr;
    const int rows = (int)v->shape[0], cols = (int)(v->numel / rows);
    float* w = alloc(v->numel);
    if (!w) return nullptr;
    hipLaunchKernelGGL(fold_wn_kernel, dim3(rows), dim3(256), 0, st, v->dev, g->dev, w, cols);
    return w;
  }
  // attach the bf16x6 planes to a finished GemmW
  void split(GemmW& g) {
    if (!g.w || rc != JV_OK) return;
    const long n = (long)g.n_rows * g.ldw;
    float* d = alloc((size_t)(3 * n + 1) / 2 + 8);
    if (!d) return;
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, st, g.w, reinterpret_cast<unsigned short*>(d), n);
    g.w3 = reinterpret_cast<const unsigned short*>(d);
  }
  // max |x| of a small device vector (load time only: synchronous)
  float host_maxabs(const float* dev, int n, int stride = 1) {
    if (!dev || rc != JV_OK) return NAN;
    std::vector<float> h((size_t)n * stride);
    if (hipMemcpyAsync(h.data(), dev, h.size() * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
      rc = fail(JV_ERR_HIP, "bound readback failed");
      return NAN;
    }
    float m = 0.f;
    for (int i = 0; i < n; ++i) {
      const float a = fabsf(h[(size_t)i * stride]);
      if (a != a) return NAN;
      m = fmaxf(m, a);
    }
    return m;
  }
  // x + sin^2(alpha x) / (alpha + 1e-9) <= |x| + 1 / (alpha + 1e-9): the largest such term over the channels, 0 when some
  // alpha is not positive (then the layer stays on bf16x6)
  float snake_extra(const float* alpha_dev, int n) {
    if (!alpha_dev || rc != JV_OK) return 0.f;
    std::vector<float> h((size_t)n);
    if (hipMemcpyAsync(h.data(), alpha_dev, h.size() * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
      rc = fail(JV_ERR_HIP, "alpha readback failed");
      return 0.f;
    }
    float m = 0.f;
    for (float a : h) {
      if (!(a > 1e-6f) || !(a < 1e30f)) return 0.f;
      m = fmaxf(m, 1.0f / (a + 1e-9f));
    }
    return m;
  }
  // attach the fp16x3 planes; returns the device vector of the rows' L1 norms (for the bound of what the layer produces)
  const float* half3(GemmW& g) {
    if (!g.w || rc != JV_OK) return nullptr;
    const long n = (long)g.n_rows * g.ldw;
    float* d = alloc((size_t)n + 8);             // 2 planes x 2 bytes
    float* cs = alloc((size_t)g.n_rows);
    float* stats = alloc((size_t)2 * g.n_rows);
    if (!d || !cs || !stats) return nullptr;
    if (split2h_planes(g.w, g.n_rows, g.ldw, stats, reinterpret_cast<unsigned short*>(d), cs, st) != JV_OK) {
      rc = JV_ERR_HIP;
      return nullptr;
    }
    g.w2 = reinterpret_cast<const unsigned short*>(d);
    g.colscale = cs;
    return stats + g.n_rows;
  }
  // ... and a second copy of the planes in fragment order for the row-owning GEMM (rowgemm_kernel.h, W-direct)
  // vocoder = true: the ResBlock convolutions (any tap count, 64 / 128 / 256 channels: hiftconv_kernel.h)
  void wfrag(GemmW& g, bool vocoder = false) {
    // linears (K = Cin) and the trunk's k = 3 convolutions (K = 3 Cin, tap-major columns: rowconv_wd_kernel)
    const int K = g.ntaps * g.Cin;
    if (!g.w2 || rc != JV_OK || g.ldw != K) return;
    if (vocoder ? ((g.N != 64 && g.N != 128 && g.N != 256) || g.Cin != g.N || g.n_rows < g.N)
                : ((g.ntaps != 1 && g.ntaps != 3) || (g.N & 255) || (g.Cin & 63))) return;
    const long n = (long)g.N * K;
    float* d = alloc((size_t)n + 8);
    if (!d) return;
    if (pack_wfrag(g.w2, (long)g.n_rows * g.ldw, g.ldw, g.N, K, reinterpret_cast<unsigned short*>(d), n, st) != JV_OK) {
      rc = JV_ERR_HIP;
      return;
    }
    g.wf = reinterpret_cast<const unsigned short*>(d);
  }
  // a vocoder ResBlock's two convolutions (same k, same channel count) as one fragment stream: c1's k C / 32 steps, then c2's
  const unsigned short* wfrag_pair(const GemmW& a, const GemmW& b) {
    const int K = a.ntaps * a.Cin;
    if (rc != JV_OK || !a.w2 || !b.w2 || a.N != b.N || a.Cin != a.N || b.Cin != b.N || a.ntaps != b.ntaps || a.ldw != K || b.ldw != K) return nullptr;
    const long plane = (long)a.N * 2 * K;      // halves
    float* d = alloc((size_t)plane + 8);
    if (!d) return nullptr;
    unsigned short* const wf = reinterpret_cast<unsigned short*>(d);
    const long second = (long)(K >> 5) * (a.N >> 4) * 512;      // k-step major: the second convolution's steps start here
    if (pack_wfrag(a.w2, (long)a.n_rows * a.ldw, a.ldw, a.N, K, wf, plane, st) != JV_OK ||
        pack_wfrag(b.w2, (long)b.n_rows * b.ldw, b.ldw, b.N, K, wf + second, plane, st) != JV_OK) {
      rc = JV_ERR_HIP;
      return nullptr;
    }
    return wf;
  }
  // a resnet's block1 (k = 3) and its 1 x 1 res_conv read the same rows: one fragment stream for both (ResnetW::wf4)
  void wfrag4(ResnetW& r) {
    const GemmW &b = r.block1, &q = r.res;
    if (rc != JV_OK || !b.w2 || !q.w2 || !b.wf || b.ntaps != 3 || q.ntaps != 1 || b.Cin != q.Cin || b.N != 256 || q.N != 256 ||
        (b.Cin & 63) || b.ldw != 3 * b.Cin || q.ldw != q.Cin) return;
    const int Cin = b.Cin, NCH = Cin >> 5;
    const long plane = 256L * 4 * Cin;      // halves
    float* d = alloc((size_t)plane + 8);
    if (!d) return;
    unsigned short* const wf = reinterpret_cast<unsigned short*>(d);
    if (pack_wfrag(b.w2, (long)b.n_rows * b.ldw, b.ldw, 256, 3 * Cin, wf, plane, st) != JV_OK ||
        pack_wfrag(q.w2, (long)q.n_rows * q.ldw, q.ldw, 256, Cin, wf + 3L * NCH * 16 * 512, plane, st) != JV_OK) {
      rc = JV_ERR_HIP;
      return;
    }
    r.wf4 = wf;
  }
  // conv weight [cout][cin][k] (device, plain) -> GemmW with cin padded to cinp
  GemmW conv(const float* w, int cout, int cin, int k, int cinp, const float* bias) {
    GemmW g;
    if (!w) return g;
    const int ldw = k * cinp;
    float* d = alloc((size_t)cout * ldw);
    if (!d) return g;
    const long total = (long)cout * ldw;
    hipLaunchKernelGGL(pack_conv_kernel, dim3((unsigned)cdivl(total, 256)), dim3(256), 0, st, w, d, cout, cin, k, cinp, ldw,
                       0);
    g.w = d; g.ldw = ldw; g.n_rows = cout; g.N = cout; g.Cin = cinp; g.ntaps = k; g.bias = bias;
    split(g);
    return g;
  }
  GemmW conv_named(const std::string& p, int cout, int cin, int k, int cinp = 0) {
    return conv(ptr(p + "weight"), cout, cin, k, cinp ? cinp : cin, ptr(p + "bias"));
  }
  GemmW conv_wn(const std::string& p, int cout, int cin, int k, int cinp = 0) {
    return conv(folded(p + "parametrizations.weight.original0", p + "parametrizations.weight.original1"), cout, cin, k,
                cinp ? cinp : cin, ptr(p + "bias"));
  }
  // [N][K] matrices usable in place (Linear, 1x1 conv)
  GemmW linear(const std::string& wname, const std::string& bname, int n, int k) {
    GemmW g;
    g.w = ptr(wname); g.ldw = k; g.n_rows = n; g.N = n; g.Cin = k; g.ntaps = 1;
    g.bias = bname.empty() ? nullptr : ptr(bname);
    split(g);
    return g;
  }
  // vertical concatenation of [n_i][k] matrices (+ optional biases)
  GemmW concat(const std::vector<std::string>& wnames, const std::vector<std::string>& bnames, int n_each, int k) {
    GemmW g;
    const int parts = (int)wnames.size();
    float* d = alloc((size_t)parts * n_each * k);
    float* b = bnames.empty() ? nullptr : alloc((size_t)parts * n_each);
    if (!d || (!bnames.empty() && !b)) return g;
    for (int i = 0; i < parts; ++i) {
      const float* w = ptr(wnames[i]);
      if (!w) return g;
      if (hipMemcpyAsync(d + (size_t)i * n_each * k, w, (size_t)n_each * k * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
        rc = fail(JV_ERR_HIP, "concat copy failed");
      if (b) {
        const float* bs = ptr(bnames[i]);
        if (!bs) return g;
        if (hipMemcpyAsync(b + (size_t)i * n_each, bs, (size_t)n_each * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
          rc = fail(JV_ERR_HIP, "concat copy failed");
      }
    }
    g.w = d; g.ldw = k; g.n_rows = parts * n_each; g.N = parts * n_each; g.Cin = k; g.ntaps = 1; g.bias = b;
    split(g);
    return g;
  }
  GemmW convT_wn(const std::string& p, int cin, int cout, int k, int s) {
    GemmW g;
    const float* w = folded(p + "parametrizations.weight.original0", p + "parametrizations.weight.original1");
    const float* bias = ptr(p + "bias");
    if (!w || !bias) return g;
    const long total = 3L * cin * s * cout;
    float* d = alloc(total);
    float* b = alloc((size_t)s * cout);
    if (!d || !b) return g;
    hipLaunchKernelGGL(pack_convT_kernel, dim3((unsigned)cdivl(total, 256)), dim3(256), 0, st, w, d, cin, cout, k, s,
                       (k - s) / 2);
    hipLaunchKernelGGL(tile_bias_kernel, dim3(cdiv(s * cout, 256)), dim3(256), 0, st, bias, b, cout, s);
    g.w = d; g.ldw = 3 * cin; g.n_rows = s * cout; g.N = s * cout; g.Cin = cin; g.ntaps = 3; g.bias = b;
    split(g);
    return g;
  }
  LnW ln(const std::string& p, const char* gname, const char* bname) { return LnW{ptr(p + gname), ptr(p + bname)}; }
};

}  // namespace

int scale_copy(const float* x, float* y, float s, int n, hipStream_t st);   // promptops.hip

int finalize_model(Context& c, int model, hipStream_t st) {
  for (const RawTensor& t : c.raw)
    if (t.model == model && !t.loaded) return fail(JV_ERR_STATE, "missing tensor: " + t.name);
  Packer pk{c, st};

  if (model == MODEL_PROMPT) {
    PromptW& w = c.prompt;
    const std::string e = "encoder.";
    w.emb = pk.ptr("input_embedding.weight");
    w.div = pk.ptr("pos_enc.div_term");
    // x * sqrt(512) after the LayerNorm (embedding.py:268) folded into its gain and offset
    auto scaled_ln = [&](const std::string& s) {
      LnW r;
      float* g = pk.alloc(PR_DIM);
      float* b = pk.alloc(PR_DIM);
      const float* g0 = pk.ptr(s + "weight");
      const float* b0 = pk.ptr(s + "bias");
      if (g && b && g0 && b0) {
        const float xs = sqrtf((float)PR_DIM);
        if (scale_copy(g0, g, xs, PR_DIM, st) != JV_OK || scale_copy(b0, b, xs, PR_DIM, st) != JV_OK) pk.rc = JV_ERR_HIP;
        r.g = g; r.b = b;
      }
      return r;
    };
    auto block = [&](const std::string& s, ConfBlockW& k) {
      k.n_mha = pk.ln(s + "norm_mha.", "weight", "bias");
      k.n_ff = pk.ln(s + "norm_ff.", "weight", "bias");
      k.qkv = pk.concat({s + "self_attn.linear_q.weight", s + "self_attn.linear_k.weight", s + "self_attn.linear_v.weight"},
                        {s + "self_attn.linear_q.bias", s + "self_attn.linear_k.bias", s + "self_attn.linear_v.bias"}, PR_DIM,
                        PR_DIM);
      k.pos = pk.linear(s + "self_attn.linear_pos.weight", "", PR_DIM, PR_DIM);
      k.out = pk.linear(s + "self_attn.linear_out.weight", s + "self_attn.linear_out.bias", PR_DIM, PR_DIM);
      k.w1 = pk.linear(s + "feed_forward.w_1.weight", s + "feed_forward.w_1.bias", PR_FFN, PR_DIM);
      k.w2 = pk.linear(s + "feed_forward.w_2.weight", s + "feed_forward.w_2.bias", PR_DIM, PR_FFN);
      k.u = pk.ptr(s + "self_attn.pos_bias_u");
      k.v = pk.ptr(s + "self_attn.pos_bias_v");
    };
    w.emb_lin = pk.linear(e + "embed.out.0.weight", e + "embed.out.0.bias", PR_DIM, PR_DIM);
    w.emb_ln = scaled_ln(e + "embed.out.1.");
    w.up_emb_lin = pk.linear(e + "up_embed.out.0.weight", e + "up_embed.out.0.bias", PR_DIM, PR_DIM);
    w.up_emb_ln = scaled_ln(e + "up_embed.out.1.");
    w.look1 = pk.conv_named(e + "pre_lookahead_layer.conv1.", PR_DIM, PR_DIM, 4);
    w.look2 = pk.conv_named(e + "pre_lookahead_layer.conv2.", PR_DIM, PR_DIM, 3);
    w.up_conv = pk.conv_named(e + "up_layer.conv.", PR_DIM, PR_DIM, 5);
    w.after = pk.ln(e + "after_norm.", "weight", "bias");
    w.proj = pk.linear("encoder_proj.weight", "encoder_proj.bias", N_FEATS, PR_DIM);
    for (int i = 0; i < PR_BLOCKS; ++i) block(e + "encoders." + S(i) + ".", w.blk[i]);
    for (int i = 0; i < PR_UP_BLOCKS; ++i) block(e + "up_encoders." + S(i) + ".", w.up[i]);
    if (pk.rc != JV_OK) return pk.rc;
    JV_HIP(hipStreamSynchronize(st));
    c.ready[MODEL_PROMPT] = true;
    return JV_OK;
  }

  if (model == MODEL_TTS) {
    // ---------------- estimator ----------------
    const std::string p = "decoder.estimator.";
    EstimatorW& e = c.est;
    e.time1 = pk.linear(p + "time_mlp.linear_1.weight", p + "time_mlp.linear_1.bias", EST_TIME, EST_IN);
    e.time2 = pk.linear(p + "time_mlp.linear_2.weight", p + "time_mlp.linear_2.bias", EST_TIME, EST_TIME);
    std::vector<std::string> stage(EST_NRES);
    stage[0] = p + "down_blocks.0.";
    for (int i = 0; i < EST_NMID; ++i) stage[1 + i] = p + "mid_blocks." + S(i) + ".";
    stage[EST_NRES - 1] = p + "up_blocks.0.";
    std::vector<std::string> tw, tb;
    float v_bound = 0.f;
    for (int i = 0; i < EST_NRES; ++i) {
      const std::string q = stage[i] + "0.";
      const int cin = i == 0 ? EST_IN : (i == EST_NRES - 1 ? 2 * EST_CH : EST_CH);
      tw.push_back(q + "mlp.1.weight");
      tb.push_back(q + "mlp.1.bias");
      e.res[i].block1 = pk.conv_named(q + "block1.block.0.", EST_CH, cin, 3);
      e.res[i].ln1 = pk.ln(q + "block1.block.2.", "weight", "bias");
      e.res[i].block2 = pk.conv_named(q + "block2.block.0.", EST_CH, EST_CH, 3);
      e.res[i].ln2 = pk.ln(q + "block2.block.2.", "weight", "bias");
      e.res[i].res = pk.linear(q + "res_conv.weight", q + "res_conv.bias", EST_CH, cin);
      // fp16x3 with the measured bound of the residual stream (flow.hip)
      (void)pk.half3(e.res[i].block1);
      (void)pk.half3(e.res[i].block2);
      pk.wfrag(e.res[i].block1); pk.wfrag(e.res[i].block2);
      (void)pk.half3(e.res[i].res);
      pk.wfrag4(e.res[i]);
      {
        const float gm = pk.host_maxabs(e.res[i].ln1.g, EST_CH), bm = pk.host_maxabs(e.res[i].ln1.b, EST_CH);
        const float hb = sqrtf(255.f) * gm + bm;
        e.res[i].h2_bound = (hb == hb && hb > 0.f && hb < 1e30f) ? hb : 0.f;
      }
      for (int j = 0; j < EST_NBLK; ++j) {
        const std::string b = stage[i] + "1." + S(j) + ".";
        BtbW& w = e.blk[i][j];
        w.n1 = pk.ln(b + "norm1.", "weight", "bias");
        w.qkv = pk.concat({b + "attn1.to_q.weight", b + "attn1.to_k.weight", b + "attn1.to_v.weight"}, {}, EST_INNER, EST_CH);
        w.out = pk.linear(b + "attn1.to_out.0.weight", b + "attn1.to_out.0.bias", EST_CH, EST_INNER);
        w.n3 = pk.ln(b + "norm3.", "weight", "bias");
        w.ff1 = pk.linear(b + "ff.net.0.proj.weight", b + "ff.net.0.proj.bias", EST_FF, EST_CH);
        w.ff2 = pk.linear(b + "ff.net.2.weight", b + "ff.net.2.bias", EST_CH, EST_FF);
        // fp16x3 where the input range is proven at load time (conv_gemm_x6_kernel.h, NP = 2).  |LayerNorm_256(x) g + b| <=
        // sqrt(255) max|g| + max|b|; a Linear of a bounded input is bounded by its largest row L1 norm; attention
        // returns convex combinations of its V rows; |gelu(x)| <= |x|.
        const float b1 = 16.f * pk.host_maxabs(w.n1.g, EST_CH) + pk.host_maxabs(w.n1.b, EST_CH);
        const float b3 = 16.f * pk.host_maxabs(w.n3.g, EST_CH) + pk.host_maxabs(w.n3.b, EST_CH);
        const float* l1_qkv = pk.half3(w.qkv);
        (void)pk.half3(w.out);
        const float* l1_ff1v = pk.half3(w.ff1);
        (void)pk.half3(w.ff2);
        pk.wfrag(w.qkv); pk.wfrag(w.out); pk.wfrag(w.ff1); pk.wfrag(w.ff2);
        const float l1_q = pk.host_maxabs(l1_qkv, EST_INNER), l1_k = pk.host_maxabs(l1_qkv ? l1_qkv + EST_INNER : nullptr, EST_INNER);
        const float l1_v = pk.host_maxabs(l1_qkv ? l1_qkv + 2 * EST_INNER : nullptr, EST_INNER);
        const float l1_ff1 = pk.host_maxabs(l1_ff1v, EST_FF);
        w.qkv.a_scale = h3_scale_for_bound(b1);
        w.ff1.a_scale = h3_scale_for_bound(b3);
        w.ff2.a_scale = h3_scale_for_bound(l1_ff1 * b3 + pk.host_maxabs(w.ff1.bias, EST_FF));
        // attention: q (with the kernel's log2(e)/8 folded in), k, v are rows of the qkv Linear (no bias)
        w.q_scale = h3_scale_for_bound(l1_q * b1 * 0.1803369f);
        w.k_scale = h3_scale_for_bound(l1_k * b1);
        w.v_scale = h3_scale_for_bound(l1_v * b1);
        if (!(w.q_scale > 0.f && w.k_scale > 0.f && w.v_scale > 0.f)) w.q_scale = w.k_scale = w.v_scale = 0.f;
        // a block whose V bound is unusable (non-finite, or beyond 1e30: h3_scale_for_bound) keeps its attention and its
        // to_out on bf16x6 and does not enter the shared scale below: one hostile layer costs that layer (and the row-owning
        // kernels of its stage, flow.hip stage_all_rg), not the engine of the other 55 blocks
        w.out.a_scale = (w.v_scale > 0.f && l1_v * b1 < 1e30f) ? 1.f : 0.f;      // (1 = "usable", replaced below)
        if (w.out.a_scale > 0.f) v_bound = fmaxf(v_bound, l1_v * b1);
      }
    }
    // the attention buffer is shared by all 56 blocks (and keeps rows of earlier calls between the utterances' windows):
    // one scale from the largest usable V bound.  (Every kernel on the path is row-local -- an output row depends on its own
    // input row alone -- and masked rows are never tracked or staged, so whatever an fp32-writing block leaves in the gap
    // rows stays there.)
    for (int i = 0; i < EST_NRES; ++i)
      for (int j = 0; j < EST_NBLK; ++j)
        if (e.blk[i][j].out.a_scale > 0.f) e.blk[i][j].out.a_scale = h3_scale_for_bound(v_bound);
    flow_ws_forget_attention(c, st);
    e.temb_all = pk.concat(tw, tb, EST_CH, EST_TIME);
    e.down_conv = pk.conv_named(p + "down_blocks.0.2.", EST_CH, EST_CH, 3);
    e.up_conv = pk.conv_named(p + "up_blocks.0.2.", EST_CH, EST_CH, 3);
    e.final_conv = pk.conv_named(p + "final_block.block.0.", EST_CH, EST_CH, 3);
    e.final_ln = pk.ln(p + "final_block.block.2.", "weight", "bias");
    e.final_proj = pk.linear(p + "final_proj.weight", p + "final_proj.bias", N_FEATS, EST_CH);
    (void)pk.half3(e.down_conv);
    (void)pk.half3(e.up_conv);
    (void)pk.half3(e.final_conv);
    (void)pk.half3(e.final_proj);
    pk.wfrag(e.down_conv); pk.wfrag(e.up_conv); pk.wfrag(e.final_conv);

    // ---------------- text encoder + duration predictor ----------------
    EncoderW& n = c.enc;
    const std::string q = "encoder.";
    n.emb = pk.ptr(q + "emb.weight");
    n.lang_emb = pk.ptr(q + "lang_emb.weight");
    n.tone_emb = pk.ptr(q + "tone_emb.weight");
    n.wpos_emb = pk.ptr(q + "word_pos_emb.weight");
    n.spos_emb = pk.ptr(q + "syllable_pos.weight");
    for (int i = 0; i < 3; ++i) {
      n.pre_conv[i] = pk.conv_named(q + "prenet.conv_layers." + S(i) + ".", ENC_CH, ENC_CH, 5);
      n.pre_ln[i] = pk.ln(q + "prenet.norm_layers." + S(i) + ".", "gamma", "beta");
    }
    n.pre_proj = pk.linear(q + "prenet.proj.weight", q + "prenet.proj.bias", ENC_CH, ENC_CH);
    for (int i = 0; i < ENC_LAYERS; ++i) {
      const std::string a = q + "encoder.attn_layers." + S(i) + ".";
      EncLayerW& L = n.layer[i];
      L.qkv = pk.concat({a + "conv_q.weight", a + "conv_k.weight", a + "conv_v.weight"},
                        {a + "conv_q.bias", a + "conv_k.bias", a + "conv_v.bias"}, ENC_HID, ENC_HID);
      L.o = pk.linear(a + "conv_o.weight", a + "conv_o.bias", ENC_HID, ENC_HID);
      L.n1 = pk.ln(q + "encoder.norm_layers_1." + S(i) + ".", "gamma", "beta");
      L.ffn1 = pk.conv_named(q + "encoder.ffn_layers." + S(i) + ".conv_1.", ENC_FILTER, ENC_HID, 3);
      L.ffn2 = pk.conv_named(q + "encoder.ffn_layers." + S(i) + ".conv_2.", ENC_HID, ENC_FILTER, 3);
      L.n2 = pk.ln(q + "encoder.norm_layers_2." + S(i) + ".", "gamma", "beta");
    }
    n.proj = pk.linear(q + "proj.weight", q + "proj.bias", N_FEATS, ENC_HID);
    n.dp_cond = pk.linear("dp.cond.weight", "dp.cond.bias", ENC_HID, SPK_DIM);
    n.dp_conv1 = pk.conv_named("dp.conv_1.", DP_FILTER, ENC_HID, 3);
    n.dp_ln1 = pk.ln("dp.norm_1.", "gamma", "beta");
    n.dp_conv2 = pk.conv_named("dp.conv_2.", DP_FILTER, DP_FILTER, 3);
    n.dp_ln2 = pk.ln("dp.norm_2.", "gamma", "beta");
    n.dp_proj = pk.linear("dp.proj.weight", "dp.proj.bias", 1, DP_FILTER);
    n.spk_affine = pk.linear("spk_embed_affine_layer.weight", "spk_embed_affine_layer.bias", N_FEATS, SPK_DIM);
  } else if (model == MODEL_HIFT) {
    HiftW& h = c.hift;
    for (int i = 0; i < 5; ++i) {
      const std::string q = "f0_predictor.condnet." + S(2 * i) + ".";
      const int cin = i == 0 ? N_FEATS : HIFT_F0_CH;
      h.f0_conv[i] = pk.conv(pk.folded(q + "weight_g", q + "weight_v"), HIFT_F0_CH, cin, 3, i == 0 ? 96 : cin, pk.ptr(q + "bias"));
    }
    h.f0_cls_w = pk.ptr("f0_predictor.classifier.weight");
    h.f0_cls_b = pk.ptr("f0_predictor.classifier.bias");
    h.src_lin_w = pk.ptr("m_source.l_linear.weight");
    h.src_lin_b = pk.ptr("m_source.l_linear.bias");
    h.conv_pre = pk.conv_wn("conv_pre.", HIFT_CH, N_FEATS, 7, 96);
    const int up_k[3] = {16, 11, 7}, up_s[3] = {8, 5, 3}, sd_k[3] = {30, 6, 1}, src_k[3] = {7, 7, 11}, rb_k[3] = {3, 7, 11};
    for (int i = 0; i < 3; ++i) {
      h.ups[i] = pk.convT_wn("ups." + S(i) + ".", HIFT_CH >> i, HIFT_CH >> (i + 1), up_k[i], up_s[i]);
      (void)pk.half3(h.ups[i]);
      // strided source conv as a GEMM over k*32 contiguous floats of the 32-column s_stft row buffer
      GemmW sd = pk.conv_named("source_downs." + S(i) + ".", HIFT_CH >> (i + 1), HIFT_NFFT + 2, sd_k[i], 32);
      sd.Cin = sd_k[i] * 32;
      sd.ntaps = 1;
      h.src_down[i] = sd;
    }
    auto resblock = [&](const std::string& q, int ch, int k) {
      ResBlockW w;
      w.k = k;
      for (int j = 0; j < 3; ++j) {
        w.c1[j] = pk.conv_wn(q + "convs1." + S(j) + ".", ch, ch, k);
        w.c2[j] = pk.conv_wn(q + "convs2." + S(j) + ".", ch, ch, k);
        w.a1[j] = pk.ptr(q + "activations1." + S(j) + ".alpha");
        w.a2[j] = pk.ptr(q + "activations2." + S(j) + ".alpha");
        // fp16x3 with a measured input bound (hift.hip): planes + what the Snake prologue can add on top of |x|
        const float* const l1 = pk.half3(w.c1[j]);
        (void)pk.half3(w.c2[j]);
        pk.wfrag(w.c1[j], true); pk.wfrag(w.c2[j], true);
        w.e1[j] = pk.snake_extra(w.a1[j], ch);
        w.e2[j] = pk.snake_extra(w.a2[j], ch);
        // the pair as one launch (64 / 128 channels): one fragment stream, and the intermediate's bound from the weights
        if ((ch == 64 || ch == 128) && w.c1[j].wf && w.c2[j].wf && l1 && w.e1[j] > 0.f && w.e2[j] > 0.f) {
          const float l1m = pk.host_maxabs(l1, ch);
          const float b1m = w.c1[j].bias ? pk.host_maxabs(w.c1[j].bias, ch) : 0.f;
          if (l1m > 0.f && l1m < 1e30f && b1m == b1m && b1m < 1e30f) {
            w.wfp[j] = pk.wfrag_pair(w.c1[j], w.c2[j]);
            w.l1max[j] = l1m;
            w.b1max[j] = b1m;
          }
        }
      }
      return w;
    };
    for (int i = 0; i < 3; ++i) h.src_rb[i] = resblock("source_resblocks." + S(i) + ".", HIFT_CH >> (i + 1), src_k[i]);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) h.rb[3 * i + j] = resblock("resblocks." + S(3 * i + j) + ".", HIFT_CH >> (i + 1), rb_k[j]);
    h.conv_post = pk.conv_wn("conv_post.", HIFT_NFFT + 2, HIFT_CH >> 3, 7);
  } else {
    return fail(JV_ERR_ARG, "finalize: unknown model id");
  }
  if (pk.rc != JV_OK) return pk.rc;
  JV_HIP(hipGetLastError());
  JV_HIP(hipStreamSynchronize(st));
  c.ready[model] = true;
  return JV_OK;
}

}  // namespace jv
