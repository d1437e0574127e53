// Text-encoder / duration / length-regulation glue kernels.
// Reference: jyutvoice/models/text_encoder.py:119-172 (RoPE), :235-246 (masked softmax), :418-445 (embeddings,
// concat); jyutvoice/models/jyutvoice_tts.py:175,184-203; jyutvoice/utils/model.py:29-46 (generate_path).
#include <math.h>

#include "jv_common.h"
#include "jv_ops.h"

namespace jv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// h[row, 0:192] = (emb[x] + tone_emb[tone] + word_pos_emb[wp] + syllable_pos[sp]) * sqrt(192)
__global__ __launch_bounds__(256) void embed_sum_kernel(const long* __restrict__ x, const long* __restrict__ tone,
                                                        const long* __restrict__ wp, const long* __restrict__ sp,
                                                        const float* __restrict__ e0, const float* __restrict__ e1,
                                                        const float* __restrict__ e2, const float* __restrict__ e3,
                                                        float* __restrict__ out, int B, int Tt, int G, int S) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // f32x4 over B*Tt*48
  if (idx >= (long)B * Tt * 48) return;
  const int c4 = (int)(idx % 48);
  const long bt = idx / 48;
  const int b = (int)(bt / Tt), t = (int)(bt - (long)b * Tt);
  const f32x4 a = *reinterpret_cast<const f32x4*>(e0 + x[bt] * 192 + 4 * c4);
  const f32x4 c = *reinterpret_cast<const f32x4*>(e1 + tone[bt] * 192 + 4 * c4);
  const f32x4 d = *reinterpret_cast<const f32x4*>(e2 + wp[bt] * 192 + 4 * c4);
  const f32x4 e = *reinterpret_cast<const f32x4*>(e3 + sp[bt] * 192 + 4 * c4);
  const f32x4 v = (((a + c) + d) + e) * 13.856406460551018f;   // sqrt(192)
  *reinterpret_cast<f32x4*>(out + ((long)G + (long)b * S + t) * 192 + 4 * c4) = v;
}

int embed_sum(const long* x, const long* tone, const long* wp, const long* sp, const float* e0, const float* e1,
              const float* e2, const float* e3, float* out, int B, int Tt, int G, int S, hipStream_t st) {
  hipLaunchKernelGGL(embed_sum_kernel, dim3((unsigned)cdivl((long)B * Tt * 48, 256)), dim3(256), 0, st, x, tone, wp, sp, e0,
                     e1, e2, e3, out, B, Tt, G, S);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// h[row, 192:384] = spk[b] (raw), h[row, 384:576] = lang_emb[lang[b,t]]; zero on padded tokens
__global__ __launch_bounds__(256) void concat_fill_kernel(const float* __restrict__ spk, const long* __restrict__ lang,
                                                          const float* __restrict__ lang_emb, const long* __restrict__ xlen,
                                                          float* __restrict__ h, int B, int Tt, int G, int S) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // f32x4 over B*Tt*96
  if (idx >= (long)B * Tt * 96) return;
  const int c4 = (int)(idx % 96);
  const long bt = idx / 96;
  const int b = (int)(bt / Tt), t = (int)(bt - (long)b * Tt);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (t < xlen[b]) {
    if (c4 < 48) v = *reinterpret_cast<const f32x4*>(spk + b * 192 + 4 * c4);
    else v = *reinterpret_cast<const f32x4*>(lang_emb + lang[bt] * 192 + 4 * (c4 - 48));
  }
  *reinterpret_cast<f32x4*>(h + ((long)G + (long)b * S + t) * 576 + 192 + 4 * c4) = v;
}

int concat_fill(const float* spk, const long* lang, const float* lang_emb, const long* xlen, float* h, int B, int Tt, int G,
                int S, hipStream_t st) {
  hipLaunchKernelGGL(concat_fill_kernel, dim3((unsigned)cdivl((long)B * Tt * 96, 256)), dim3(256), 0, st, spk, lang, lang_emb,
                     xlen, h, B, Tt, G, S);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// int64 lengths -> int32 lengths
__global__ void lens_to_i32_kernel(const long* a, int* o, int n, int cap) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = (int)min((long)cap, max(0L, a[i]));
}
int lens_to_i32(const long* a, int* o, int n, int cap, hipStream_t st) {
  hipLaunchKernelGGL(lens_to_i32_kernel, dim3(cdiv(n, 64)), dim3(64), 0, st, a, o, n, cap);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// RoPE on q and k of a fused [rows,1728] qkv buffer: 2 heads x 288, first 144 dims, pairs (i, i+72), position = t
__global__ __launch_bounds__(256) void rope_kernel(float* __restrict__ qkv, int B, int Tt, int G, int S) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // over B*Tt*(2 tensors * 2 heads * 72)
  if (idx >= (long)B * Tt * 288) return;
  const int i = (int)(idx % 72);
  const int hh = (int)((idx / 72) % 4);                   // 0,1: q heads; 2,3: k heads
  const long bt = idx / 288;
  const int b = (int)(bt / Tt), t = (int)(bt - (long)b * Tt);
  // theta_i = 1 / 10000^(2i/144), in fp32 like torch (text_encoder.py:119)
  const float theta = 1.0f / powf(10000.0f, (float)(2 * i) / 144.0f);
  const float ang = (float)t * theta;
  const float cs = cosf(ang), sn = sinf(ang);
  float* p = qkv + ((long)G + (long)b * S + t) * 1728 + (hh >> 1) * 576 + (hh & 1) * 288;
  const float a = p[i], c = p[i + 72];
  p[i] = a * cs + (-c) * sn;
  p[i + 72] = c * cs + a * sn;
}

int rope_qk(float* qkv, int B, int Tt, int G, int S, hipStream_t st) {
  hipLaunchKernelGGL(rope_kernel, dim3((unsigned)cdivl((long)B * Tt * 288, 256)), dim3(256), 0, st, qkv, B, Tt, G, S);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// scores [B,H,Tt,ld] (raw q.k) -> P = softmax(masked_fill(scores/sqrt(288), mask2d == 0, -1e4)), zero-padded to ld.
// one wave per (b,h,query) row.
__global__ __launch_bounds__(256) void enc_softmax_kernel(float* __restrict__ sc, const long* __restrict__ xlen, int B, int H,
                                                          int Tt, int ld) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)B * H * Tt) return;
  const int q = (int)(row % Tt);
  const int b = (int)(row / ((long)H * Tt));
  const int len = (int)xlen[b];
  float* p = sc + row * ld;
  const float inv = 1.0f / sqrtf(288.0f);
  float v[8];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = lane + 64 * i;
    float s = -INFINITY;
    if (k < Tt) s = (q < len && k < len) ? p[k] * inv : -1e4f;
    v[i] = s;
    m = fmaxf(m, s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i] = (lane + 64 * i < Tt) ? expf(v[i] - m) : 0.f;
    sum += v[i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float r = 1.0f / sum;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = lane + 64 * i;
    if (k < ld) p[k] = v[i] * r;
  }
}

int enc_softmax(float* sc, const long* xlen, int B, int H, int Tt, int ld, hipStream_t st) {
  if (ld > 512) return fail(JV_ERR_SHAPE, "encoder attention supports up to 512 tokens");
  hipLaunchKernelGGL(enc_softmax_kernel, dim3((unsigned)cdivl((long)B * H * Tt, 4)), dim3(256), 0, st, sc, xlen, B, H, Tt, ld);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// Vt[b,h,d,k] = qkv[row(b,k), 1152 + h*288 + d] for k < Tt, zero for Tt <= k < ld
__global__ __launch_bounds__(256) void transpose_v_kernel(const float* __restrict__ qkv, float* __restrict__ vt, int B, int Tt,
                                                          int ld, int G, int S) {
  __shared__ float tile[32][33];
  const int bh = blockIdx.z, b = bh >> 1, h = bh & 1;
  const int k0 = blockIdx.x * 32, d0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = k0 + ty + 8 * i, d = d0 + tx;
    tile[ty + 8 * i][tx] = (k < Tt) ? qkv[((long)G + (long)b * S + k) * 1728 + 1152 + h * 288 + d] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int d = d0 + ty + 8 * i, k = k0 + tx;
    if (k < ld) vt[((long)bh * 288 + d) * ld + k] = tile[tx][ty + 8 * i];
  }
}

int transpose_v(const float* qkv, float* vt, int B, int Tt, int ld, int G, int S, hipStream_t st) {
  hipLaunchKernelGGL(transpose_v_kernel, dim3(cdiv(ld, 32), 9, B * 2), dim3(256), 0, st, qkv, vt, B, Tt, ld, G, S);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// out[row, :] = x[row, :] + vec[b(row), :]   (duration predictor: x + cond(spk), duration_predictor.py:50)
__global__ __launch_bounds__(256) void add_rowvec_kernel(const float* __restrict__ x, const float* __restrict__ vec,
                                                         float* __restrict__ out, int B, int Tt, int G, int S, int C) {
  const int c4n = C >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * Tt * c4n) return;
  const int c4 = (int)(idx % c4n);
  const long bt = idx / c4n;
  const int b = (int)(bt / Tt), t = (int)(bt - (long)b * Tt);
  const long o = ((long)G + (long)b * S + t) * C + 4 * c4;
  *reinterpret_cast<f32x4*>(out + o) = *reinterpret_cast<const f32x4*>(x + o) + *reinterpret_cast<const f32x4*>(vec + (long)b * C + 4 * c4);
}
int add_rowvec(const float* x, const float* vec, float* out, int B, int Tt, int G, int S, int C, hipStream_t st) {
  hipLaunchKernelGGL(add_rowvec_kernel, dim3((unsigned)cdivl((long)B * Tt * (C >> 2), 256)), dim3(256), 0, st, x, vec, out, B,
                     Tt, G, S, C);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// F.normalize(spk, dim=1): x / max(||x||_2, 1e-12), one wave per utterance, 192 dims
__global__ void l2_normalize_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int C) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float s = 0.f;
  for (int i = lane; i < C; i += 64) s += x[b * C + i] * x[b * C + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float d = fmaxf(sqrtf(s), 1e-12f);
  for (int i = lane; i < C; i += 64) out[b * C + i] = x[b * C + i] / d;
}
int l2_normalize(const float* x, float* out, int B, int C, hipStream_t st) {
  hipLaunchKernelGGL(l2_normalize_kernel, dim3(B), dim3(64), 0, st, x, out, B, C);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// ---- length regulation -------------------------------------------------------------------------------------
// w_ceil = ceil(exp(logw) * mask) * length_scale; cum = torch.cumsum semantics on the CPU (sequential fp64
// accumulation, every prefix rounded to fp32); y_len = max(long(sum), 1).  One lane per utterance (Tt <= 512).
__global__ void durations_kernel(const float* __restrict__ logw, const long* __restrict__ xlen, float scale,
                                 float* __restrict__ w_ceil, float* __restrict__ cum, long* __restrict__ ylen, int B, int Tt) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double c = 0.0;
  for (int t = 0; t < Tt; ++t) {
    const float m = t < xlen[b] ? 1.f : 0.f;
    const float w = ceilf(expf(logw[b * Tt + t]) * m) * scale;
    w_ceil[b * Tt + t] = w;
    c += (double)w;
    cum[b * Tt + t] = (float)c;
  }
  ylen[b] = max((long)(float)c, 1L);
}

// attn[b,i,j] = ([j < cum_i] - [j < cum_{i-1}]) * xmask_i * ymask_j   (utils/model.py:36-45)
__global__ __launch_bounds__(256) void path_kernel(const float* __restrict__ cum, const long* __restrict__ xlen,
                                                   const long* __restrict__ ylen, float* __restrict__ attn, int B, int Tt,
                                                   int Ty) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * Tt * Ty) return;
  const int j = (int)(idx % Ty);
  const long bi = idx / Ty;
  const int b = (int)(bi / Tt), i = (int)(bi - (long)b * Tt);
  const float fj = (float)j;
  const float hi = fj < cum[b * Tt + i] ? 1.f : 0.f;
  const float lo = (i > 0 && fj < cum[b * Tt + i - 1]) ? 1.f : 0.f;
  const float m = (i < xlen[b] && j < ylen[b]) ? 1.f : 0.f;
  attn[idx] = (hi - lo) * m;
}

// mu_y[b,c,j] = mu_x[b,c,tok(j)], tok(j) = first i with j < cum_i (the matmul with a 0/1 path is a gather)
__global__ __launch_bounds__(128) void expand_mu_kernel(const float* __restrict__ cum, const long* __restrict__ xlen,
                                                        const long* __restrict__ ylen, const float* __restrict__ mu_x,
                                                        float* __restrict__ mu_y, int B, int Tt, int Ty) {
  const int b = blockIdx.y, j = blockIdx.x;
  __shared__ int tok;
  if (threadIdx.x == 0) {
    int t = -1;
    if (j < ylen[b]) {
      const float fj = (float)j;
      const int n = (int)min((long)Tt, xlen[b]);
      for (int i = 0; i < n; ++i)
        if (fj < cum[b * Tt + i]) { t = i; break; }
    }
    tok = t;
  }
  __syncthreads();
  const int t = tok;
  for (int c = threadIdx.x; c < 80; c += blockDim.x)
    mu_y[((long)b * 80 + c) * Ty + j] = t >= 0 ? mu_x[((long)b * 80 + c) * Tt + t] : 0.f;
}

int durations(const float* logw, const long* xlen, float scale, float* w_ceil, float* cum, long* ylen, int B, int Tt,
              hipStream_t st) {
  hipLaunchKernelGGL(durations_kernel, dim3(cdiv(B, 64)), dim3(64), 0, st, logw, xlen, scale, w_ceil, cum, ylen, B, Tt);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

int paths(const float* cum, const long* xlen, const long* ylen, const float* mu_x, float* attn, float* mu_y, int B, int Tt,
          int Ty, hipStream_t st) {
  hipLaunchKernelGGL(path_kernel, dim3((unsigned)cdivl((long)B * Tt * Ty, 256)), dim3(256), 0, st, cum, xlen, ylen, attn, B, Tt,
                     Ty);
  hipLaunchKernelGGL(expand_mu_kernel, dim3(Ty, B), dim3(128), 0, st, cum, xlen, ylen, mu_x, mu_y, B, Tt, Ty);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

}  // namespace jv
