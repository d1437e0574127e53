// Row-wise kernels of the prompt encoder (prompt.hip): token embedding, relative positional table, (q + u, q + v),
// the rel-shifted masked softmax, V transpose, nearest-neighbour x2 row repeat.
// jyutvoice/transformer/embedding.py:224-254, attention.py:226-246, 290-334, upsample_encoder.py:45-61.
#include <math.h>

#include "jv_common.h"

namespace jv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// rows[G + b*S + t][0:512] = emb[clamp(tok[b][t], 0)] for t < len[b], zero for len[b] <= t < T
__global__ __launch_bounds__(256) void prompt_embed_kernel(const long* __restrict__ tok, const long* __restrict__ len,
                                                           const float* __restrict__ emb, float* __restrict__ rows, int B,
                                                           int T, int G, int S, int vocab) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // f32x4 index over B*T*128
  if (idx >= (long)B * T * 128) return;
  const int c4 = (int)(idx & 127);
  const long bt = idx >> 7;
  const int b = (int)(bt / T), t = (int)(bt - (long)b * T);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (t < (int)len[b]) {
    long id = tok[(long)b * T + t];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    v = *reinterpret_cast<const f32x4*>(emb + id * 512 + 4 * c4);
  }
  *reinterpret_cast<f32x4*>(rows + ((long)G + (long)b * S + t) * 512 + 4 * c4) = v;
}

int prompt_embed(const long* tok, const long* len, const float* emb, float* rows, int B, int T, int G, int S, int vocab,
                 hipStream_t st) {
  hipLaunchKernelGGL(prompt_embed_kernel, dim3((unsigned)cdivl((long)B * T * 128, 256)), dim3(256), 0, st, tok, len, emb, rows,
                     B, T, G, S, vocab);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// pe[m][2i] = sin(r * w_i), pe[m][2i+1] = cos(r * w_i), r = T - 1 - m (relative position i - j), m < 2T - 1.
// w_i = exp(2i * -(ln 1e4 / 512)) comes from the host (`div`, the reference's own torch expression, embedding.py:239-242):
// at |r| ~ 1000 one ulp of w_i moves the angle by 6e-5, so the frequencies are not recomputed here.  The reference
// builds the r < 0 half from sin(-1 * position * div_term) (:245-246); the product has the same magnitude either way.
__global__ __launch_bounds__(256) void rel_pos_table_kernel(float* __restrict__ pe, const float* __restrict__ div, int T) {
  const int m = blockIdx.x, i = threadIdx.x;   // 256 frequency pairs
  if (m >= 2 * T - 1) return;
  const float a = (float)(T - 1 - m) * div[i];
  pe[(long)m * 512 + 2 * i] = sinf(a);
  pe[(long)m * 512 + 2 * i + 1] = cosf(a);
}

int rel_pos_table(float* pe, const float* div, int T, hipStream_t st) {
  hipLaunchKernelGGL(rel_pos_table_kernel, dim3(2 * T - 1), dim3(256), 0, st, pe, div, T);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// qu = q + pos_bias_u[h], qv = q + pos_bias_v[h]   (q = columns [0,512) of the [rows,1536] qkv buffer)
__global__ __launch_bounds__(256) void add_pos_bias_kernel(const float* __restrict__ qkv, const float* __restrict__ u,
                                                           const float* __restrict__ v, float* __restrict__ qu,
                                                           float* __restrict__ qv, long rows) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * 128) return;
  const int c4 = (int)(idx & 127);
  const long r = idx >> 7;
  const f32x4 q = *reinterpret_cast<const f32x4*>(qkv + r * 1536 + 4 * c4);
  *reinterpret_cast<f32x4*>(qu + r * 512 + 4 * c4) = q + *reinterpret_cast<const f32x4*>(u + 4 * c4);
  *reinterpret_cast<f32x4*>(qv + r * 512 + 4 * c4) = q + *reinterpret_cast<const f32x4*>(v + 4 * c4);
}

int add_pos_bias(const float* qkv, const float* u, const float* v, float* qu, float* qv, long rows, hipStream_t st) {
  hipLaunchKernelGGL(add_pos_bias_kernel, dim3((unsigned)cdivl(rows * 128, 256)), dim3(256), 0, st, qkv, u, v, qu, qv, rows);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// P[b,h,i,j] = softmax_j((ac[b,h,i,j] + bd[b,h,i, j - i + T - 1]) / 8) over keys j < len[b]; masked keys and the padding
// columns T <= j < ld are written as 0 (they are the K operand's zero padding of the P V product).  One wave per row.
__global__ __launch_bounds__(256) void rel_softmax_kernel(float* __restrict__ ac, const float* __restrict__ bd,
                                                          const long* __restrict__ len, int len_mul, int B, int H, int T,
                                                          int ld, int ldb) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)B * H * T) return;
  const int i = (int)(row % T);
  const int b = (int)(row / ((long)H * T));
  const int L = min((int)len[b] * len_mul, T);
  float* p = ac + row * ld;
  const float* q = bd + row * ldb + (T - 1 - i);
  float m = -INFINITY;
  for (int j = lane; j < L; j += 64) {
    const float s = (p[j] + q[j]) * 0.125f;
    p[j] = s;
    m = fmaxf(m, s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  float sum = 0.f;
  for (int j = lane; j < L; j += 64) {
    const float e = expf(p[j] - m);
    p[j] = e;
    sum += e;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float r = L > 0 ? 1.0f / sum : 0.f;
  for (int j = lane; j < ld; j += 64) p[j] = j < L ? p[j] * r : 0.f;
}

int rel_softmax(float* ac, const float* bd, const long* len, int len_mul, int B, int H, int T, int ld, int ldb,
                hipStream_t st) {
  hipLaunchKernelGGL(rel_softmax_kernel, dim3((unsigned)cdivl((long)B * H * T, 4)), dim3(256), 0, st, ac, bd, len, len_mul, B,
                     H, T, ld, ldb);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// Vt[b,h,d,k] = qkv[row(b,k), 1024 + h*64 + d] for k < T, zero for T <= k < ld
__global__ __launch_bounds__(256) void prompt_transpose_v_kernel(const float* __restrict__ qkv, float* __restrict__ vt, int T,
                                                                 int ld, int G, int S) {
  __shared__ float tile[32][33];
  const int bh = blockIdx.z, b = bh >> 3, h = bh & 7;
  const int k0 = blockIdx.x * 32, d0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = k0 + ty + 8 * i, d = d0 + tx;
    tile[ty + 8 * i][tx] = (k < T) ? qkv[((long)G + (long)b * S + k) * 1536 + 1024 + h * 64 + d] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int d = d0 + ty + 8 * i, k = k0 + tx;
    if (k < ld) vt[((long)bh * 64 + d) * ld + k] = tile[tx][ty + 8 * i];
  }
}

int prompt_transpose_v(const float* qkv, float* vt, int B, int T, int ld, int G, int S, hipStream_t st) {
  hipLaunchKernelGGL(prompt_transpose_v_kernel, dim3(cdiv(ld, 32), 2, B * 8), dim3(256), 0, st, qkv, vt, T, ld, G, S);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// F.interpolate(scale_factor=2, mode="nearest") on rows: dst[G2 + b*S2 + v] = src[G + b*S + (v >> 1)], v < 2T
__global__ __launch_bounds__(256) void repeat_rows2_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int T,
                                                           int G, int S, int G2, int S2) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * 2 * T * 128) return;
  const int c4 = (int)(idx & 127);
  const long bv = idx >> 7;
  const int b = (int)(bv / (2 * T)), v = (int)(bv - (long)b * 2 * T);
  *reinterpret_cast<f32x4*>(dst + ((long)G2 + (long)b * S2 + v) * 512 + 4 * c4) =
      *reinterpret_cast<const f32x4*>(src + ((long)G + (long)b * S + (v >> 1)) * 512 + 4 * c4);
}

int repeat_rows2(const float* src, float* dst, int B, int T, int G, int S, int G2, int S2, hipStream_t st) {
  hipLaunchKernelGGL(repeat_rows2_kernel, dim3((unsigned)cdivl((long)B * 2 * T * 128, 256)), dim3(256), 0, st, src, dst, B, T,
                     G, S, G2, S2);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// out[b][t][0:80] = rows[G + b*S + t][0:80] for t < len[b]*len_mul, else 0   ([B, T, 80] contiguous: prompt_h)
__global__ __launch_bounds__(256) void rows_to_btc_kernel(const float* __restrict__ rows, const long* __restrict__ len,
                                                          int len_mul, float* __restrict__ out, int B, int T, int C, int G,
                                                          int S) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * T * C) return;
  const int c = (int)(idx % C);
  const long bt = idx / C;
  const int b = (int)(bt / T), t = (int)(bt - (long)b * T);
  out[idx] = t < (int)len[b] * len_mul ? rows[((long)G + (long)b * S + t) * C + c] : 0.f;
}

int rows_to_btc(const float* rows, const long* len, int len_mul, float* out, int B, int T, int C, int G, int S,
                hipStream_t st) {
  hipLaunchKernelGGL(rows_to_btc_kernel, dim3((unsigned)cdivl((long)B * T * C, 256)), dim3(256), 0, st, rows, len, len_mul, out,
                     B, T, C, G, S);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// y = x * s  (LayerNorm gains / offsets pre-multiplied by sqrt(512) at load time)
__global__ void scale_copy_kernel(const float* __restrict__ x, float* __restrict__ y, float s, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = x[i] * s;
}

int scale_copy(const float* x, float* y, float s, int n, hipStream_t st) {
  hipLaunchKernelGGL(scale_copy_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, x, y, s, n);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

}  // namespace jv
