// Device-side helpers shared by the kernels.
#pragma once
#include <math.h>

#include <type_traits>

#include "jv_common.h"

namespace jv {

// Exact (erf) GELU, 0.5 v (1 + erf(v / sqrt 2)), in ONE branch-free chain of 14 operations:
//     gelu(v) = relu(v) - u E(u),   u = |v|,   E(u) = erfc(u / sqrt 2) / 2 = 2^Q(u)
// with Q a degree-9 polynomial (weighted minimax fit of log2 E on [0, 6.5], weight u E ln 2 = the sensitivity of the result;
// tools/fit_gelu.py holds the fit and the error measurement) and the raw v_exp_f32.  Beyond u = 6.5 the correction term is
// frozen at 6.5 E(6.5) = 1e-9 (the true one is smaller still); relu is written (v + |v|) / 2 so that a NaN propagates (v_max
// would return the other operand).  Measured against fp64 over [-12, 12] and at +-30, 100, 1e4: absolute error <= 4.4e-8
// for v < 0, relative error <= 1.9e-7 for v > 0.05 -- the accuracy of the two-branch form it replaces (rounds 1-2: N. Juffa's
// erff pair evaluated both ways and selected, 27 operations: 8.3e-8 / 1.8e-7) and tighter than torch's own fp32 gelu
// (1.1e-6 / 3.8e-7), at half the operations: the GELU pass between
// the two feed-forward GEMMs is pure VALU work with the matrix pipe idle (20 % of the fused block launch in the phase stamps).
__device__ __forceinline__ float gelu_erf(const float v) {
  const float u = fminf(fabsf(v), 6.5f);
  float q = 3.9286530295612465e-07f;
  q = fmaf(q, u, -7.801393621775787e-06f);
  q = fmaf(q, u, 6.441200821427628e-05f);
  q = fmaf(q, u, -0.00025067193200811744f);
  q = fmaf(q, u, -4.676209937315434e-05f);
  q = fmaf(q, u, 0.0069969831965863705f);
  q = fmaf(q, u, -0.0524740107357502f);
  q = fmaf(q, u, -0.45920976996421814f);
  q = fmaf(q, u, -1.151105523109436f);
  q = fmaf(q, u, -1.0f);
  const float w = u * __builtin_amdgcn_exp2f(q);
  return fmaf(0.5f, v + fabsf(v), -w);
}

// The same chain on TWO values per instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 are full-rate on gfx950: an fma per
// lane and component is the same IEEE operation, so the results are gelu_erf's bit for bit).  For the fused block's GELU pass,
// which is pure vector issue with the matrix pipe idle: nine of the chain's fourteen operations are the polynomial.
typedef float jv_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ jv_f32x2 gelu_erf2(const jv_f32x2 v) {
  const jv_f32x2 a = {fabsf(v[0]), fabsf(v[1])};
  const jv_f32x2 u = {fminf(a[0], 6.5f), fminf(a[1], 6.5f)};
  auto c2 = [](const float c) { return jv_f32x2{c, c}; };
  jv_f32x2 q = c2(3.9286530295612465e-07f);
  q = __builtin_elementwise_fma(q, u, c2(-7.801393621775787e-06f));
  q = __builtin_elementwise_fma(q, u, c2(6.441200821427628e-05f));
  q = __builtin_elementwise_fma(q, u, c2(-0.00025067193200811744f));
  q = __builtin_elementwise_fma(q, u, c2(-4.676209937315434e-05f));
  q = __builtin_elementwise_fma(q, u, c2(0.0069969831965863705f));
  q = __builtin_elementwise_fma(q, u, c2(-0.0524740107357502f));
  q = __builtin_elementwise_fma(q, u, c2(-0.45920976996421814f));
  q = __builtin_elementwise_fma(q, u, c2(-1.151105523109436f));
  q = __builtin_elementwise_fma(q, u, c2(-1.0f));
  const jv_f32x2 w = u * jv_f32x2{__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
  return __builtin_elementwise_fma(c2(0.5f), v + a, -w);
}

// sum over the 64 lanes of a wave, returned in every lane: four DPP butterflies inside each row of 16 lanes, then the four
// row totals R0..R3 combined as (R0 + R1) + (R2 + R3) with gfx950's row / half exchanges (v_permlane16_swap: rows 1 and 3 of
// the first operand trade places with rows 0 and 2 of the second; v_permlane32_swap: the same for the two halves) -- all
// VALU, no LDS round trips (__shfl_xor is ds_bpermute) and no trip through SGPRs.  (Round 2 read the row totals with four
// v_readlane and added them as wave-uniform values: the same association, hence the same bits, but each readlane result
// costs hazard wait states and a v_mov before a VALU add can take two of them -- measured in the row-owning kernels'
// LayerNorm epilogues, where nothing hides latency: 20 reductions per wave.)
__device__ __forceinline__ float wave_sum(float x) {
  auto dpp = [](float v, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  x += dpp(x, std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
  x += dpp(x, std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
  x += dpp(x, std::integral_constant<int, 0x141>{});     // row_half_mirror
  x += dpp(x, std::integral_constant<int, 0x140>{});     // row_mirror
  const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);      // even row + odd row, in every lane of both
  const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r32[0]) + __uint_as_float(r32[1]);   // lower half + upper half
}

// activations as the reference's PyTorch ops compute them (exact erf GELU, ...).  Mish x tanh(softplus x) is evaluated
// as x n / (n + 2), n = e^x (e^x + 2) -- the same function (tanh(log(1 + e)) = ((1 + e)^2 - 1) / ((1 + e)^2 + 1)) with one
// transcendental instead of three; against fp64 its fp32 error (3.6e-7 relative) matches torch's own mish (2.6e-7).
__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case ACT_RELU: return fmaxf(v, 0.f);
    case ACT_GELU: return gelu_erf(v);
    case ACT_MISH: {
      const float e = expf(fminf(v, 20.f));
      const float n = e * (e + 2.f);
      return v > 20.f ? v : v * (n / (n + 2.f));   // softplus threshold 20 of F.mish: tanh is 1 to fp32 there
    }
    case ACT_ELU: return v > 0.f ? v : expm1f(v);
    case ACT_SILU: return v / (1.f + expf(-v));
    case ACT_LOGCLIP: return logf(fmaxf(v, 1e-5f));   // dynamic_range_compression of utils/audio.py:9-10
    default: return v;
  }
}

// Mish with the raw v_exp_f32 and v_rcp_f32 (each ~1 ulp) instead of expf and an IEEE division: the same x n / (n + 2)
// form as act_apply, ~9 instructions instead of ~25.  For epilogues that run on the few waves of a row-owning workgroup,
// where nothing hides VALU latency (rowconv_kernel.h); relative error vs fp64 <= 5e-7, the same order as act_apply's 3.6e-7.
__device__ __forceinline__ float mish_fast(const float v) {
  const float e = __builtin_amdgcn_exp2f(fminf(v, 20.f) * 1.44269504088896340736f);
  const float n = e * (e + 2.f);
  return v > 20.f ? v : v * (n * __builtin_amdgcn_rcpf(n + 2.f));
}

// sin^2(a) for the Snake activation x + sin^2(alpha x) / alpha (jyutvoice/hifigan/generator.py Snake).  sinf() costs ~130
// VALU instructions per value on gfx950 (its Payne-Hanek path is compiled in line), which made the Snake-prologue
// convolutions VALU-bound.  Here: a = k pi + r with a two-constant Cody-Waite reduction (exact products through fma; the
// sign of sin(r) is irrelevant once squared), then an odd degree-11 polynomial on [-pi/2, pi/2].  |a| <= 2^15: absolute
// error <= 2.3e-7 against fp64 (a correctly rounded sinf squared: 0.9e-7).  Callers route larger or non-finite arguments
// to sinf (snake_args_small).
__device__ __forceinline__ float sin2_small(const float a) {
  const float k = rintf(a * 0.31830988618379067154f);
  float r = fmaf(-k, 3.1415927410125732f, a);
  r = fmaf(-k, -8.742277657347586e-08f, r);
  const float u = r * r;
  float q = fmaf(-4.054625790672617e-08f, u, 2.843463789758971e-06f);
  q = fmaf(q, u, -1.9857272855006158e-04f);
  q = fmaf(q, u, 8.333439007401466e-03f);
  q = fmaf(q, u, -1.666666865348816e-01f);
  const float sn = fmaf(r * u, q, r);
  return sn * sn;
}
// wave-uniform: true when every lane's argument is in sin2_small's range (NaN compares false and propagates through it)
__device__ __forceinline__ bool snake_args_small(const bool lane_has_big) { return __builtin_amdgcn_ballot_w64(lane_has_big) == 0; }

// bf16x3 split of two fp32 values at once: x ~= h + m + l to 24 bits (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m),
// round to nearest even).  Returns each plane as one dword holding the pair (element 0 in the low half), the form
// v_cvt_pk_bf16_f32 produces and the MFMA operands consume.
typedef float jv_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 jv_bf16x2 __attribute__((ext_vector_type(2)));
struct Split3 { unsigned h, m, l; };
__device__ __forceinline__ Split3 split3_pair(const float x0, const float x1) {
  unsigned h, m, l;
  const jv_f32x2 x = {x0, x1};
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(x, jv_bf16x2));
  // scalar subtractions on purpose: beside MFMAs a packed v_pk_add_f32 issues slower than two v_sub_f32
  // (MI355X_MICROARCH.md, 'price of one filler'; measured here: attention 111 -> 104 us), and the build passes
  // -fno-slp-vectorize so the compiler does not re-pack them
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  const jv_f32x2 r = {r0, r1};
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(r, jv_bf16x2));
  const float q0 = r0 - __uint_as_float(m << 16), q1 = r1 - __uint_as_float(m & 0xffff0000u);
  const jv_f32x2 q = {q0, q1};
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(q, jv_bf16x2));
  return {h, m, l};
}

// fp16x2 split of two fp32 values: x ~= h + l with h = fp16(x), l = fp16(x - h): 22 significant bits while l is a normal
// fp16 (|x| >= 2^-3), an absolute error of at most 2^-25 below that (fp16 subnormals are kept by the conversion and by
// v_mfma_f32_32x32x16_f16 on gfx950 -- tools/probes/f16_denorm.hip).  |x| must stay below 65504: callers scale by a power
// of two chosen from a proven bound.
typedef _Float16 jv_f16x2 __attribute__((ext_vector_type(2)));
struct Split2 { unsigned h, l; };
// largest power of two s with bound * s <= 60000 (the host's h3_scale_for_bound, registry.hip), for a bound measured on
// the device: the exponent of 60000 / bound with the mantissa cleared.  A bound that is not finite (the producer already
// wrote inf or NaN: the integer max of bit patterns propagates both into the slot) gives scale 0 ON PURPOSE: every output
// of the utterance that owns the slot then comes out NaN (acc * colscale / 0), i.e. the fault stays visible and stays
// inside that utterance -- fmaxf(NaN, x) would have returned x and scaled the finite rows into fp16 overflow silently.
__device__ __forceinline__ float h3_scale_dev(const float bound) {
  if (!(bound <= 3.0e38f)) return 0.f;
  const float q = 60000.f / fmaxf(bound, 1e-30f);
  return fminf(__uint_as_float(__float_as_uint(q) & 0xff800000u), 16777216.f);
}
__device__ __forceinline__ Split2 split2h_pair(const float x0, const float x1) {
  const jv_f32x2 x = {x0, x1};
  const jv_f16x2 h = __builtin_convertvector(x, jv_f16x2);
  // x * 1 - h with a 1.0 the optimiser cannot see through: the residual then is ONE v_fma_mix_f32 per element (h read as
  // fp16 in place, same exactly-rounded x - h) instead of v_cvt_f32_f16 + v_sub_f32
  float one = 1.0f;
  asm("" : "+v"(one));      // (not volatile: a volatile asm per pair chains the pairs of an unrolled loop in program order)
  const float r0 = fmaf(x0, one, -(float)h[0]), r1 = fmaf(x1, one, -(float)h[1]);
  const jv_f32x2 r = {r0, r1};
  const jv_f16x2 l = __builtin_convertvector(r, jv_f16x2);
  return {__builtin_bit_cast(unsigned, h), __builtin_bit_cast(unsigned, l)};
}

// Attention's running softmax (attention_pl.hip, attention_s.hip -- the same rule in both, so an utterance's result does not
// depend on which of them its batch selects): the exponent's reference maximum moves only when a key tile's maximum exceeds it
// by more than ATTN_LAZY (base-2 units).  P = p 2^10 can then reach 2^(10 + ATTN_LAZY) = 2^14 < 65504, and the accumulated
// output is rescaled a few times per head instead of in nearly every tile (a record high among a tile's queries is the rule,
// one 16x higher is not).  Exact in exact arithmetic: the reference cancels in O / l.
constexpr float ATTN_LAZY = 4.0f;
__device__ __forceinline__ float attn_lazy_max(const float m_run, const float mt) { return mt > m_run + ATTN_LAZY ? mt : m_run; }

}  // namespace jv
