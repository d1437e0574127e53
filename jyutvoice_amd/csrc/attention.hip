// Self-attention for the flow estimator's BasicTransformerBlock (8 heads x 64, fp32, key-padding
// mask), flash-style on the fp32 matrix cores.  Replaces: q k^T / 8 + bias(0 | -1e10), softmax, . v
// of diffusers' AttnProcessor2_0 as called from jyutvoice/flow/transformer.py:380-389 with the
// [B',T,T] bias of jyutvoice/flow/decoder.py:951-959 (a pure key mask: exp(-1e10 - max) == 0 in fp32,
// so masked keys are simply skipped here).
//
// One wave owns 32 queries.  Scores are computed TRANSPOSED, S^T = K . Q^T (A = K rows from LDS,
// B = Q held in registers), so the 32x32 accumulator has the query on the lane and 16 keys in the
// registers: the softmax row statistics are per-lane scalars, and the probabilities are already laid
// out as the B operand of the second product O^T = V^T . P^T -- no LDS round trip, no shuffles except
// one lane^32 exchange per reduction.  K/V tiles of 32 keys are shared by the workgroup's waves through
// LDS (row stride 68 floats: conflict-free ds_read_b128) and register-prefetched one tile ahead.
#include <math.h>

#include "jv_common.h"

namespace jv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KV_STRIDE = 68;

template <int NW>
__global__ __launch_bounds__(64 * NW) void attn64_kernel(const AttnArgs p) {
  __shared__ __attribute__((aligned(16))) float ldsK[32 * KV_STRIDE];
  __shared__ __attribute__((aligned(16))) float ldsV[32 * KV_STRIDE];
  constexpr int NT = 64 * NW;
  constexpr int NLD = (32 * 16) / NT;   // f32x4 per thread per tile, each of K and V

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, half = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * 32 * NW + wave * 32;
  const int len = p.lens ? min(p.lens[b], p.L) : p.L;
  const long rowbase = (long)p.G + (long)b * p.S;
  const bool active = q0 < p.L;

  // Q fragment: this lane's query row, dims [32*half, 32*half+32), pre-scaled by log2(e)/sqrt(64): scores come out in
  // base-2 units so the softmax needs one v_exp_f32 per element (exp(x) == exp2(x log2 e))
  float q[32];
  {
    const int qi = q0 + r32;
    const float* src = p.qkv + (rowbase + qi) * p.ld + h * 64 + 32 * half;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      if (qi < p.L) t = *reinterpret_cast<const f32x4*>(src + 4 * i);
#pragma unroll
      for (int e = 0; e < 4; ++e) q[4 * i + e] = t[e] * (0.125f * 1.44269504088896340736f);
    }
  }

  f32x16 o0, o1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;

  f32x4 pk[NLD], pv[NLD];
  auto prefetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * NT;
      const int key = idx >> 4, c4 = idx & 15;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      if (k0 + key < len) {
        const float* src = p.qkv + (rowbase + k0 + key) * p.ld + h * 64 + 4 * c4;
        pk[i] = *reinterpret_cast<const f32x4*>(src + p.k_off);
        pv[i] = *reinterpret_cast<const f32x4*>(src + p.v_off);
      } else {
        pk[i] = z;
        pv[i] = z;
      }
    }
  };

  const int nkt = (len + 31) >> 5;
  if (nkt > 0) prefetch(0);
  for (int kt = 0; kt < nkt; ++kt) {
    const int k0 = kt * 32;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * NT;
      const int key = idx >> 4, c4 = idx & 15;
      *reinterpret_cast<f32x4*>(ldsK + key * KV_STRIDE + 4 * c4) = pk[i];
      *reinterpret_cast<f32x4*>(ldsV + key * KV_STRIDE + 4 * c4) = pv[i];
    }
    __syncthreads();
    if (kt + 1 < nkt) prefetch(k0 + 32);
    if (!active) continue;

    // S^T[key][query] = sum_d K[key][d] * Q[query][d]
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
    const float* kr = ldsK + r32 * KV_STRIDE + 32 * half;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(kr + 4 * i);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, q[4 * i + 0], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, q[4 * i + 1], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, q[4 * i + 2], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, q[4 * i + 3], s, 0, 0, 0);
    }
    // key mask + online softmax (statistics are per query = per lane column)
    float mt = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = k0 + (e & 3) + 8 * (e >> 2) + 4 * half;
      s[e] = key < len ? s[e] : -INFINITY;
      mt = fmaxf(mt, s[e]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32));
    const float m_new = fmaxf(m_run, mt);
    const float alpha = exp2f(m_run - m_new);
    float lt = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      s[e] = exp2f(s[e] - m_new);
      lt += s[e];
    }
    lt += __shfl_xor(lt, 32);
    l_run = l_run * alpha + lt;
    m_run = m_new;
#pragma unroll
    for (int e = 0; e < 16; ++e) { o0[e] *= alpha; o1[e] *= alpha; }
    // O^T[d][query] += sum_key V[key][d] * P[query][key]; register e of s holds keys (e&3)+8(e>>2)+4*half
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float* vr = ldsV + ((e & 3) + 8 * (e >> 2) + 4 * half) * KV_STRIDE + r32;
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[0], s[e], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[32], s[e], o1, 0, 0, 0);
    }
  }

  const int qi = q0 + r32;
  if (active && qi < p.L) {
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
    float* dst = p.out + (rowbase + qi) * p.ldo + h * 64 + 4 * half;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 a = {o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv};
      const f32x4 c = {o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv};
      *reinterpret_cast<f32x4*>(dst + 8 * g) = a;
      *reinterpret_cast<f32x4*>(dst + 32 + 8 * g) = c;
    }
  }
}

int attention64(const AttnArgs& a, hipStream_t st) {
  if (a.B <= 0 || a.L <= 0) return JV_OK;
  if ((a.ld & 3) || (a.ldo & 3) || (a.k_off & 3) || (a.v_off & 3))
    return fail(JV_ERR_ARG, "attention64: strides/offsets must be multiples of 4 floats");
  // 64-query workgroups waste least on T = 300 (5 x 64); 128-query ones halve K/V staging at T = 512
  const int waste2 = round_up(a.L, 64) - a.L, waste4 = round_up(a.L, 128) - a.L;
  const bool prof = prof_on();
  if (prof) prof_begin(st);
  if (waste4 <= waste2) {
    hipLaunchKernelGGL((attn64_kernel<4>), dim3(cdiv(a.L, 128), a.H, a.B), dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL((attn64_kernel<2>), dim3(cdiv(a.L, 64), a.H, a.B), dim3(128), 0, st, a);
  }
  if (prof) {
    // algorithmic (full-length) figure of SURVEY.md 8(d): QK^T + PV = 4*L*L*64 per head; q,k,v,o once
    const double bh = (double)a.B * a.H;
    prof_end(st, waste4 <= waste2 ? "attn64<4 waves>" : "attn64<2 waves>", 4.0 * bh * a.L * a.L * 64.0, 4.0 * bh * a.L * 64.0 * 4.0);
  }
  JV_HIP(hipGetLastError());
  return JV_OK;
}

}  // namespace jv
