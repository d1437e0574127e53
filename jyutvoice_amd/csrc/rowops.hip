// Bandwidth-bound row-buffer kernels: LayerNorm, layout packing, the flow solver's glue.
// All of them move 16 B per lane, one wave (or a few lanes) per row, rows contiguous in HBM.
#include <math.h>

#include "jv_common.h"
#include "jv_device.h"
#include "jv_ops.h"

namespace jv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- LayerNorm over the channel axis of a row buffer (optionally of x + add) -------------------
// One wave per row; two-pass (mean, then centred sum of squares) entirely in registers for C <= 1024.
// C == 256 fast path: one wave handles 4 rows at once (4 independent 16-byte loads per lane in flight), the rest as below
// PL: the result leaves as two fp16 planes of value * scale (out2[0/1][row][256], the pre-split A operand of the
// fp16x3 GEMM that consumes it -- same bytes as the fp32 row) instead of fp32
template <bool PL>
__global__ __launch_bounds__(256) void layernorm256_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                           const float* __restrict__ g, const float* __restrict__ b, float eps,
                                                           long rows, unsigned short* __restrict__ out2, long plane, float scale) {
  const int lane = threadIdx.x & 63;
  const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
  f32x4 v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    v[r] = (row0 + r < rows) ? *reinterpret_cast<const f32x4*>(x + (row0 + r) * 256 + 4 * lane) : f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 gg = *reinterpret_cast<const f32x4*>(g + 4 * lane);
  const f32x4 bb = *reinterpret_cast<const f32x4*>(b + 4 * lane);
  float sum[4], sq[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) sum[r] = (v[r][0] + v[r][1]) + (v[r][2] + v[r][3]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int r = 0; r < 4; ++r) sum[r] += __shfl_xor(sum[r], o);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const f32x4 d = v[r] - sum[r] * (1.f / 256.f);
    sq[r] = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int r = 0; r < 4; ++r) sq[r] += __shfl_xor(sq[r], o);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (row0 + r < rows) {
      const float mean = sum[r] * (1.f / 256.f);
      const float rstd = 1.0f / sqrtf(sq[r] * (1.f / 256.f) + eps);
      const f32x4 y = (v[r] - mean) * rstd * gg + bb;
      if constexpr (PL) {
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        const Split2 s0 = split2h_pair(y[0] * scale, y[1] * scale), s1 = split2h_pair(y[2] * scale, y[3] * scale);
        unsigned short* o2 = out2 + (row0 + r) * 256 + 4 * lane;
        *reinterpret_cast<u32x2*>(o2) = u32x2{s0.h, s1.h};
        *reinterpret_cast<u32x2*>(o2 + plane) = u32x2{s0.l, s1.l};
      } else {
        *reinterpret_cast<f32x4*>(out + (row0 + r) * 256 + 4 * lane) = y;
      }
    }
  }
}

template <int VPL>   // f32x4 per lane
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                                             float* __restrict__ out, const float* __restrict__ g,
                                                             const float* __restrict__ b, float eps, long rows, int C,
                                                             const unsigned char* __restrict__ rowmask_out, int relu) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int c4n = C >> 2;
  f32x4 v[VPL];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c4 = lane + 64 * i;
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    if (c4 < c4n) {
      t = *reinterpret_cast<const f32x4*>(x + row * C + 4 * c4);
      if (add) t += *reinterpret_cast<const f32x4*>(add + row * C + 4 * c4);
    }
    v[i] = t;
    sum += (t[0] + t[1]) + (t[2] + t[3]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < c4n) {
      const f32x4 d = v[i] - mean;
      sq += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
  const bool keep = !rowmask_out || rowmask_out[row];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < c4n) {
      const f32x4 gg = *reinterpret_cast<const f32x4*>(g + 4 * c4);
      const f32x4 bb = *reinterpret_cast<const f32x4*>(b + 4 * c4);
      f32x4 r = (v[i] - mean) * rstd * gg + bb;
      if (relu) r = f32x4{fmaxf(r[0], 0.f), fmaxf(r[1], 0.f), fmaxf(r[2], 0.f), fmaxf(r[3], 0.f)};
      if (!keep) r = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(out + row * C + 4 * c4) = r;
    }
  }
}

int layernorm_rows(const float* x, const float* add, float* out, const float* g, const float* b, float eps, long rows,
                   int C, const unsigned char* rowmask_out, hipStream_t st, int relu) {
  if (rows <= 0) return JV_OK;
  if (C & 3 || C > 1024) return fail(JV_ERR_ARG, "layernorm_rows: C must be a multiple of 4, <= 1024");
  const dim3 grid((unsigned)cdivl(rows, 4));
  const bool prof = prof_on();
  if (prof) prof_begin(st);
  if (C == 256 && !add && !rowmask_out && !relu)
    hipLaunchKernelGGL(layernorm256_kernel<false>, dim3((unsigned)cdivl(rows, 16)), dim3(256), 0, st, x, out, g, b, eps, rows,
                       nullptr, 0L, 1.f);
  else if (C <= 256)
    hipLaunchKernelGGL((layernorm_rows_kernel<1>), grid, dim3(256), 0, st, x, add, out, g, b, eps, rows, C, rowmask_out, relu);
  else if (C <= 768)
    hipLaunchKernelGGL((layernorm_rows_kernel<3>), grid, dim3(256), 0, st, x, add, out, g, b, eps, rows, C, rowmask_out, relu);
  else
    hipLaunchKernelGGL((layernorm_rows_kernel<4>), grid, dim3(256), 0, st, x, add, out, g, b, eps, rows, C, rowmask_out, relu);
  if (prof) prof_end(st, "layernorm", 0.0, 4.0 * rows * C * (add ? 3 : 2));   // HBM-bound: rows read (+ add) and written once
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// ---- LayerNorm epilogue of a convolution, in place: x = (act(LN(x)) * mask + rowvec[sample] + res) * scale ----------
// (CausalBlock1D's conv -> LayerNorm -> Mish -> mask, the resnet's time-embedding add and residual; decoder.py:784-788,
// 110-115).  One wave per row, C <= 1024, 16 B per lane.
template <int VPL>
__global__ __launch_bounds__(256) void ln_epilogue_kernel(float* __restrict__ x, const float* __restrict__ g,
                                                          const float* __restrict__ b, float eps, long rows, int C, int act,
                                                          const unsigned char* __restrict__ rowmask,
                                                          const float* __restrict__ rowvec, const int* __restrict__ row_sample,
                                                          int rowvec_ld, const float* __restrict__ res, long ldr, float scale,
                                                          float* __restrict__ amax_out, int amax_G, int amax_S, int amax_nb,
                                                          const int* __restrict__ amax_rows) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int c4n = C >> 2;
  f32x4 v[VPL];
  float sum = 0.f;
  unsigned amax = 0u;      // max |value written|, as conv_gemm's epilogue tracks it (ConvGemmArgs::amax_out)
  // the row's utterance owns the slot (ConvGemmArgs::amax_G/S/nb); rows the mask marks as padding are not tracked
  if (amax_out && amax_rows) {      // compact geometry (ConvGemmArgs::amax_rows)
    const int sl = amax_rows[row];
    amax_out += sl < 0 ? 0 : (sl >= amax_nb ? amax_nb - 1 : sl);
  } else if (amax_out && amax_S > 0) {
    const long sl = (row - amax_G) / amax_S;
    amax_out += sl < 0 ? 0 : (sl >= amax_nb ? amax_nb - 1 : sl);
  }
  // what the slot held when this wave started (it only grows, so a stale value is a valid lower bound); read here so
  // that the load's latency hides behind the row loads instead of extending the wave's tail
  // (a plain, cacheable load on purpose: twenty thousand waves polling one L2 line with a volatile load cost 8 us)
  const unsigned seen = amax_out ? *reinterpret_cast<const unsigned*>(amax_out) : 0xffffffffu;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c4 = lane + 64 * i;
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    if (c4 < c4n) t = *reinterpret_cast<const f32x4*>(x + row * C + 4 * c4);
    v[i] = t;
    sum += (t[0] + t[1]) + (t[2] + t[3]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    if (lane + 64 * i < c4n) {
      const f32x4 d = v[i] - mean;
      sq += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
  const bool keep = !rowmask || rowmask[row];
  const float* rv = rowvec ? rowvec + (long)row_sample[row] * rowvec_ld : nullptr;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < c4n) {
      const f32x4 gg = *reinterpret_cast<const f32x4*>(g + 4 * c4);
      const f32x4 bb = *reinterpret_cast<const f32x4*>(b + 4 * c4);
      f32x4 r = (v[i] - mean) * rstd * gg + bb;
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = keep ? act_apply(r[e], act) : 0.f;
      if (rv) r += *reinterpret_cast<const f32x4*>(rv + 4 * c4);
      if (res) r += *reinterpret_cast<const f32x4*>(res + row * ldr + 4 * c4);
      r = r * scale;
      *reinterpret_cast<f32x4*>(x + row * C + 4 * c4) = r;
#pragma unroll
      for (int e = 0; e < 4; ++e) amax = max(amax, __float_as_uint(r[e]) & 0x7fffffffu);
    }
  }
  // wave-uniform: once the slot is warm no lane exceeds it, and the cross-lane reduction (six dependent LDS round trips at
  // the very end of a one-row wave: +50 % on this bandwidth-bound kernel) is not entered at all
  if (amax_out && keep && __builtin_amdgcn_ballot_w64(amax > seen) != 0) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = max(amax, (unsigned)__shfl_xor((int)amax, o));
    if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(amax_out), amax);
  }
}

int ln_epilogue_rows(float* x, const float* g, const float* b, float eps, long rows, int C, int act,
                     const unsigned char* rowmask, const float* rowvec, const int* row_sample, int rowvec_ld, const float* res,
                     long ldr, float scale, hipStream_t st, float* amax_out, int amax_G, int amax_S, int amax_nb,
                     const int* amax_rows) {
  if (rows <= 0) return JV_OK;
  if ((C & 3) || C > 1024 || (rowvec && (rowvec_ld & 3)) || (res && (ldr & 3)))
    return fail(JV_ERR_ARG, "ln_epilogue_rows: C, rowvec_ld and ldr must be multiples of 4 (C <= 1024)");
  const dim3 grid((unsigned)cdivl(rows, 4));
  const bool prof = prof_on();
  if (prof) prof_begin(st);
  if (C <= 256)
    hipLaunchKernelGGL((ln_epilogue_kernel<1>), grid, dim3(256), 0, st, x, g, b, eps, rows, C, act, rowmask, rowvec, row_sample,
                       rowvec_ld, res, ldr, scale, amax_out, amax_G, amax_S, amax_nb, amax_rows);
  else
    hipLaunchKernelGGL((ln_epilogue_kernel<4>), grid, dim3(256), 0, st, x, g, b, eps, rows, C, act, rowmask, rowvec, row_sample,
                       rowvec_ld, res, ldr, scale, amax_out, amax_G, amax_S, amax_nb, amax_rows);
  if (prof) prof_end(st, "ln_epilogue", 0.0, 4.0 * rows * C * (res ? 3 : 2));
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// ---- split-K tail (short M): sum of the partials + everything row-wise that follows the GEMM ---------------------------
// One wave per row of 256 columns.  With few rows (a single utterance) the K = 512 ... 1536 contractions of the estimator are
// 20 workgroups x up to 48 dependent K steps; split over 4 ... 8 workgroups per tile they finish in a few steps each, and this
// kernel does what their epilogues did (bias, LayerNorm -> Mish -> mask, time embedding, residual, bound tracking) AND the
// LayerNorm that feeds the next GEMM -- one launch where round 1 had the long GEMM plus a LayerNorm launch.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const SplitKReduceArgs a) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  const float* src = a.partial + row * 256 + 4 * lane;
  f32x4 v = *reinterpret_cast<const f32x4*>(src);
  for (int z = 1; z < a.ksplit; ++z) v += *reinterpret_cast<const f32x4*>(src + (long)z * a.split_stride);      // fixed order: deterministic
  if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + 4 * lane);
  const int sample = a.row_sample ? a.row_sample[row] : 0;
  float* slot = a.amax_out ? a.amax_out + sample : nullptr;
  const unsigned seen = slot ? *reinterpret_cast<const unsigned*>(slot) : 0xffffffffu;
  const bool keep = !a.rowmask || a.rowmask[row] != 0;
  if (a.ln) {
    const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.f / 256.f);
    const f32x4 d = v - mean;
    const float rstd = 1.0f / sqrtf(wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.f / 256.f) + a.ln_eps);
    v = d * rstd * *reinterpret_cast<const f32x4*>(a.ln_g + 4 * lane) + *reinterpret_cast<const f32x4*>(a.ln_b + 4 * lane);
  }
  if (a.ln || a.act != ACT_NONE || a.rowmask) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = keep ? act_apply(v[e], a.act) : 0.f;
  }
  if (a.rowvec) v += *reinterpret_cast<const f32x4*>(a.rowvec + (long)sample * a.rowvec_ld + 4 * lane);
  if (a.res) v += *reinterpret_cast<const f32x4*>(a.res + row * a.ldr + 4 * lane);
  *reinterpret_cast<f32x4*>(a.out + row * a.ldo + 4 * lane) = v;
  if (slot && (!a.amax_mask || a.amax_mask[row] != 0)) {
    unsigned u = 0u;
#pragma unroll
    for (int e = 0; e < 4; ++e) u = max(u, __float_as_uint(v[e]) & 0x7fffffffu);
    if (__builtin_amdgcn_ballot_w64(u > seen) != 0) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) u = max(u, (unsigned)__shfl_xor((int)u, o));
      if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(slot), u);
    }
  }
  if (a.out2) {
    const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.f / 256.f);
    const f32x4 d = v - mean;
    const float rstd = 1.0f / sqrtf(wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.f / 256.f) + 1e-5f);
    *reinterpret_cast<f32x4*>(a.out2 + row * 256 + 4 * lane) =
        d * rstd * *reinterpret_cast<const f32x4*>(a.ln2_g + 4 * lane) + *reinterpret_cast<const f32x4*>(a.ln2_b + 4 * lane);
  }
}

int splitk_reduce_rows(const SplitKReduceArgs& a, hipStream_t st) {
  if (a.rows <= 0) return JV_OK;
  if (!a.partial || a.ksplit < 1 || !a.out || (a.ldo & 3) || (a.res && (a.ldr & 3)) || (a.rowvec && (a.rowvec_ld & 3)) ||
      (a.ln && (!a.ln_g || !a.ln_b)) || (a.out2 && (!a.ln2_g || !a.ln2_b)))
    return fail(JV_ERR_ARG, "splitk_reduce_rows: bad arguments");
  const bool prof = prof_on();
  if (prof) prof_begin(st);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdivl(a.rows, 4)), dim3(256), 0, st, a);
  if (prof) prof_end(st, "splitk_reduce", 0.0, 4.0 * a.rows * 256 * (a.ksplit + 1 + (a.res ? 1 : 0) + (a.out2 ? 1 : 0)));
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// ---- row metadata: rowmask[r] (1 = real frame) and row_sample[r] (utterance index) -----------------
// Two geometries.  Uniform (uoff == null): utterance b, frame t at row G + b S + t.  COMPACT (uoff != null, ragged batches,
// flow.hip): utterance b starts at row uoff[b] and owns lens[b % nb] rows + the gap up to uoff[b + 1] -- no row is spent on
// padding an utterance to the longest (uoff has nb * reps + 1 entries, the last one = the first row past the batch).
__global__ void row_meta_kernel(unsigned char* rowmask, int* row_sample, const int* lens, int nb, int reps, int G, int S,
                                int L, long rows, int mul, int add, const int* __restrict__ uoff) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const int B = nb * reps;
  if (uoff) {
    // binary search: the last utterance whose first row is <= r (rows ahead of the first utterance: utterance 0, masked)
    int lo = 0, hi = B - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (uoff[mid] <= r) lo = mid; else hi = mid - 1;
    }
    const long t = r - uoff[lo];
    const bool ok = t >= 0 && r < uoff[B] && t < (lens ? (long)lens[lo % nb] * mul + add : (long)L);
    rowmask[r] = ok ? 1 : 0;
    if (row_sample) row_sample[r] = lo;
    return;
  }
  const long rel = r - G;
  int b = rel >= 0 ? (int)(rel / S) : 0;
  const int t = rel >= 0 ? (int)(rel - (long)b * S) : -1;
  bool ok = rel >= 0 && b < B && t < L;
  if (b >= B) b = B - 1;
  if (ok && lens) ok = t < lens[b % nb] * mul + add;
  rowmask[r] = ok ? 1 : 0;
  if (row_sample) row_sample[r] = b;
}

int row_meta(unsigned char* rowmask, int* row_sample, const int* lens, int nb, int reps, int G, int S, int L, long rows,
             int mul, int add, hipStream_t st, const int* uoff) {
  hipLaunchKernelGGL(row_meta_kernel, dim3((unsigned)cdivl(rows, 256)), dim3(256), 0, st, rowmask, row_sample, lens, nb,
                     reps, G, S, L, rows, mul, add, uoff);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// ---- channels-first [B,C,T] <-> row buffer [G + b*S + t][ld] (LDS-tiled transpose) ----------------
__global__ __launch_bounds__(256) void cf_to_rows_kernel(const float* __restrict__ src, long src_bstride, long pitch, int C,
                                                         int T, float* __restrict__ dst, int ld, int col0, int G, int S,
                                                         float scale, const int* __restrict__ lens, const int* __restrict__ uoff) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const int tmax = lens ? min(lens[b], T) : T;
  // compact geometry: an utterance owns its own frames only (the rows behind them are the next utterance's)
  const long row0 = uoff ? (long)uoff[b] : (long)G + (long)b * S;
  const int twr = uoff ? tmax : T;
  if (t0 >= twr) return;      // (uniform over the block)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, t = t0 + tx;
    tile[ty + 8 * i][tx] = (c < C && t < tmax) ? src[b * src_bstride + (long)c * pitch + t] * scale : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = t0 + ty + 8 * i, c = c0 + tx;
    if (t < twr && c < C) dst[(row0 + t) * ld + col0 + c] = tile[tx][ty + 8 * i];
  }
}

int cf_to_rows(const float* src, long src_bstride, long pitch, int B, int C, int T, float* dst, int ld, int col0, int G,
               int S, float scale, const int* lens, hipStream_t st, const int* uoff) {
  if (B <= 0 || T <= 0) return JV_OK;
  if (uoff && !lens) return fail(JV_ERR_ARG, "cf_to_rows: the compact geometry needs the lengths");
  hipLaunchKernelGGL(cf_to_rows_kernel, dim3(cdiv(T, 32), cdiv(C, 32), B), dim3(256), 0, st, src, src_bstride, pitch, C, T,
                     dst, ld, col0, G, S, scale, lens, uoff);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

__global__ __launch_bounds__(256) void rows_to_cf_kernel(const float* __restrict__ src, int ld, int col0, int G, int S,
                                                         float* __restrict__ dst, long dst_bstride, int C, int T,
                                                         const int* __restrict__ lens, const int* __restrict__ uoff) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int tmax = lens ? min(lens[b], T) : T;
  const long row0 = uoff ? (long)uoff[b] : (long)G + (long)b * S;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = t0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (t < tmax && c < C) ? src[(row0 + t) * ld + col0 + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, t = t0 + tx;
    if (c < C && t < T) dst[b * dst_bstride + (long)c * T + t] = tile[tx][ty + 8 * i];
  }
}

int rows_to_cf(const float* src, int ld, int col0, int G, int S, float* dst, long dst_bstride, int B, int C, int T,
               const int* lens, hipStream_t st, const int* uoff) {
  if (B <= 0 || T <= 0) return JV_OK;
  if (uoff && !lens) return fail(JV_ERR_ARG, "rows_to_cf: the compact geometry needs the lengths");
  hipLaunchKernelGGL(rows_to_cf_kernel, dim3(cdiv(T, 32), cdiv(C, 32), B), dim3(256), 0, st, src, ld, col0, G, S, dst,
                     dst_bstride, C, T, lens, uoff);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// ---- flow estimator input: [x | mu | spks | cond] per row, CFG rows appended ---------------------------
// rows of utterance b' < B are conditional; b' >= B are the unconditional twin of b' - B (mu = spks = cond = 0),
// jyutvoice/flow/flow_matching.py:246-251.  x/mu/cond are row buffers of 80 columns with geometry (G,S).
__global__ __launch_bounds__(256) void assemble_xin_kernel(const float* __restrict__ x, const float* __restrict__ mu,
                                                           const float* __restrict__ spks, const float* __restrict__ cond,
                                                           float* __restrict__ xin, int B, int G, int S, int L, long rows2,
                                                           const int* __restrict__ uoff, const int* __restrict__ row_sample,
                                                           const unsigned char* __restrict__ rowmask) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // one f32x4 (4 of 320 columns) per thread
  const long r = idx / 80;
  const int c4 = (int)(idx - r * 80);
  if (r >= rows2) return;
  const long rel = r - G;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (rel >= 0) {
    // compact geometry (row_meta): the row's utterance and validity come from the row tables, its frame from uoff
    const int b2 = uoff ? row_sample[r] : (int)(rel / S);
    const int t = uoff ? (rowmask[r] ? (int)(r - uoff[b2]) : L) : (int)(rel - (long)b2 * S);
    if (b2 < 2 * B && t < L) {
      const bool un = b2 >= B;
      const int b = un ? b2 - B : b2;
      const long sr = uoff ? (long)uoff[b] + t : (long)G + (long)b * S + t;
      const int part = c4 / 20, cc = (c4 % 20) * 4;
      if (part == 0) v = *reinterpret_cast<const f32x4*>(x + sr * 80 + cc);
      else if (!un) {
        if (part == 1) v = *reinterpret_cast<const f32x4*>(mu + sr * 80 + cc);
        else if (part == 2) v = *reinterpret_cast<const f32x4*>(spks + b * 80 + cc);
        else v = *reinterpret_cast<const f32x4*>(cond + sr * 80 + cc);
      }
    }
  }
  *reinterpret_cast<f32x4*>(xin + r * 320 + 4 * c4) = v;
}

int assemble_xin(const float* x, const float* mu, const float* spks, const float* cond, float* xin, int B, int G, int S,
                 int L, long rows2, hipStream_t st, const int* uoff, const int* row_sample, const unsigned char* rowmask) {
  if (uoff && (!row_sample || !rowmask)) return fail(JV_ERR_ARG, "assemble_xin: the compact geometry needs the row tables");
  hipLaunchKernelGGL(assemble_xin_kernel, dim3((unsigned)cdivl(rows2 * 80, 256)), dim3(256), 0, st, x, mu, spks, cond, xin,
                     B, G, S, L, rows2, uoff, row_sample, rowmask);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// generic variant for jv_flow_estimator_step (caller supplies all 2B rows, nothing implied zero)
__global__ __launch_bounds__(256) void assemble_xin_plain_kernel(const float* __restrict__ x, const float* __restrict__ mu,
                                                                 const float* __restrict__ spks,
                                                                 const float* __restrict__ cond, float* __restrict__ xin,
                                                                 int B, int G, int S, int L, long rows) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long r = idx / 80;
  const int c4 = (int)(idx - r * 80);
  if (r >= rows) return;
  const long rel = r - G;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (rel >= 0) {
    const int b = (int)(rel / S);
    const int t = (int)(rel - (long)b * S);
    if (b < B && t < L) {
      const int part = c4 / 20, cc = (c4 % 20) * 4;
      if (part == 0) v = *reinterpret_cast<const f32x4*>(x + r * 80 + cc);
      else if (part == 1) v = *reinterpret_cast<const f32x4*>(mu + r * 80 + cc);
      else if (part == 2) v = *reinterpret_cast<const f32x4*>(spks + b * 80 + cc);
      else v = *reinterpret_cast<const f32x4*>(cond + r * 80 + cc);
    }
  }
  *reinterpret_cast<f32x4*>(xin + r * 320 + 4 * c4) = v;
}

int assemble_xin_plain(const float* x, const float* mu, const float* spks, const float* cond, float* xin, int B, int G,
                       int S, int L, long rows, hipStream_t st) {
  hipLaunchKernelGGL(assemble_xin_plain_kernel, dim3((unsigned)cdivl(rows * 80, 256)), dim3(256), 0, st, x, mu, spks, cond,
                     xin, B, G, S, L, rows);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// ---- sinusoidal timestep embedding (jyutvoice/flow/decoder.py:21-30), 320 = 160 sin | 160 cos ------------
__global__ void time_sinusoid_kernel(const float* __restrict__ t, int t_stride, float* __restrict__ out, int B) {
  const int b = blockIdx.x, j = threadIdx.x;   // 160 threads
  if (b >= B || j >= 160) return;
  const float c = (float)(-9.210340371976184 / 159.0);     // -(ln 10000)/(half-1), rounded to f32 like torch
  const float f = expf((float)j * c);
  const float a = (1000.0f * t[b * t_stride]) * f;
  out[b * 320 + j] = sinf(a);
  out[b * 320 + 160 + j] = cosf(a);
}

int time_sinusoid(const float* t, int t_stride, float* out, int B, hipStream_t st) {
  hipLaunchKernelGGL(time_sinusoid_kernel, dim3(B), dim3(192), 0, st, t, t_stride, out, B);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// ---- Euler step with classifier-free guidance: x += dt * ((1+r) d_cond - r d_uncond) ----------------------
// d holds 2B utterances (geometry G,S); step scalars come from a device table so the loop never syncs.
__global__ __launch_bounds__(256) void euler_cfg_kernel(float* __restrict__ x, const float* __restrict__ d, int B, int G, int S,
                                                        int L, const float* __restrict__ dt_table, int step, float rate,
                                                        const int* __restrict__ uoff, const int* __restrict__ lens) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // f32x4 index over B*L*20
  const long per_b = (long)L * 20;
  if (idx >= per_b * B) return;
  const int b = (int)(idx / per_b);
  const long rem = idx - (long)b * per_b;
  const long t = rem / 20;
  if (uoff && t >= lens[b]) return;      // compact geometry: the rows behind an utterance's frames are the next utterance's
  const long r = uoff ? (long)uoff[b] + t : (long)G + (long)b * S + t;
  const long ru = uoff ? (long)uoff[b + B] + t : r + (long)B * S;      // the unconditional twin's row
  const int cc = (int)(rem % 20) * 4;
  const float dt = dt_table[step];
  const f32x4 dc = *reinterpret_cast<const f32x4*>(d + r * 80 + cc);
  const f32x4 du = *reinterpret_cast<const f32x4*>(d + ru * 80 + cc);
  f32x4 xv = *reinterpret_cast<f32x4*>(x + r * 80 + cc);
  const f32x4 g = (1.0f + rate) * dc - rate * du;
  xv = xv + dt * g;
  *reinterpret_cast<f32x4*>(x + r * 80 + cc) = xv;
}

int euler_cfg(float* x, const float* d, int B, int G, int S, int L, const float* dt_table, int step, float rate,
              hipStream_t st, const int* uoff, const int* lens) {
  if (uoff && !lens) return fail(JV_ERR_ARG, "euler_cfg: the compact geometry needs the lengths");
  hipLaunchKernelGGL(euler_cfg_kernel, dim3((unsigned)cdivl((long)B * L * 20, 256)), dim3(256), 0, st, x, d, B, G, S, L,
                     dt_table, step, rate, uoff, lens);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

__global__ void fill_int_kernel(int* p, int v, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
int fill_int(int* p, int v, long n, hipStream_t st) {
  if (n <= 0) return JV_OK;
  hipLaunchKernelGGL(fill_int_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, st, p, v, n);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

__global__ void fill_kernel(float* p, float v, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
int fill(float* p, float v, long n, hipStream_t st) {
  if (n <= 0) return JV_OK;
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, st, p, v, n);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// LayerNorm_256 whose result is written as the pre-split fp16x3 operand (see layernorm256_kernel)
int layernorm256_planes(const float* x, unsigned short* out2, long plane, float scale, const float* g, const float* b, float eps,
                        long rows, hipStream_t st) {
  if (rows <= 0) return JV_OK;
  const bool prof = prof_on();
  if (prof) prof_begin(st);
  hipLaunchKernelGGL(layernorm256_kernel<true>, dim3((unsigned)cdivl(rows, 16)), dim3(256), 0, st, x, nullptr, g, b, eps, rows,
                     out2, plane, scale);
  if (prof) prof_end(st, "layernorm", 0.0, 4.0 * rows * 256 * 2);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// fp32 rows [rows][C] (row stride ld) -> two fp16 planes [2][rows][C] of x * scale (the A operand of the fp16x3 GEMM)
__global__ __launch_bounds__(256) void split2h_rows_kernel(const float* __restrict__ x, long ld, unsigned short* __restrict__ dst,
                                                           long plane, long rows, int C, float scale, long ldd) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;      // pair index
  const int c2 = C >> 1;
  if (i >= rows * c2) return;
  const long r = i / c2;
  const int c = (int)(i - r * c2) * 2;
  const float* s = x + r * ld + c;
  const Split2 t = split2h_pair(s[0] * scale, s[1] * scale);
  *reinterpret_cast<unsigned*>(dst + r * ldd + c) = t.h;
  *reinterpret_cast<unsigned*>(dst + plane + r * ldd + c) = t.l;
}

int split2h_rows(const float* x, long ld, unsigned short* dst, long plane, long rows, int C, float scale, hipStream_t st, long ldd) {
  hipLaunchKernelGGL(split2h_rows_kernel, dim3((unsigned)cdivl(rows * (C >> 1), 256)), dim3(256), 0, st, x, ld, dst, plane, rows,
                     C, scale, ldd > 0 ? ldd : (long)C);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

}  // namespace jv
