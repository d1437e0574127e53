// Device-side helpers shared by the kernels.
#pragma once
#include <math.h>

#include <type_traits>

#include "jv_common.h"

namespace jv {

// erf(a) in fp32 without branches: two minimax fits (|a| <= 0.927734375: odd polynomial; beyond: 1 - exp(-poly(|a|))) whose
// coefficients are N. Juffa's published single-precision erff (< 1 ulp); here both are evaluated and selected, and the
// exponential is the raw v_exp_f32.  Measured against fp64 over [-6, 6]: 5.8e-8 absolute, 8.8e-8 relative -- the accuracy of
// libm's erff, at ~20 VALU instructions instead of the device library's branchy ~60 (the GELU epilogue of ff.net.0 was 30 %
// of that GEMM's time).
__device__ __forceinline__ float erf_fast(const float a) {
  const float t = fabsf(a), s = a * a;
  float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
  const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
  r = fmaf(r, s, u);
  r = fmaf(r, t, -1.06777877e-1f);
  r = fmaf(r, t, -6.34846687e-1f);
  r = fmaf(r, t, -1.28717512e-1f);
  r = fmaf(r, t, -t);
  r = 1.0f - __builtin_amdgcn_exp2f(r * 1.44269504088896340736f);
  const float big = copysignf(r, a);
  float q = -5.96761703e-4f;
  q = fmaf(q, s, 4.99119423e-3f);
  q = fmaf(q, s, -2.67681349e-2f);
  q = fmaf(q, s, 1.12819925e-1f);
  q = fmaf(q, s, -3.76125336e-1f);
  q = fmaf(q, s, 1.28379166e-1f);
  const float small = fmaf(q, a, a);
  return t > 0.927734375f ? big : small;      // NaN: the comparison is false, `small` propagates it
}
__device__ __forceinline__ float gelu_erf(const float v) { return 0.5f * v * (1.f + erf_fast(v * 0.70710678118654752440f)); }

// Two values at once with packed fp32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32: two IEEE operations per lane and instruction):
// the same operations in the same order as erf_fast / gelu_erf, so the same bits, in ~30 VALU instructions per pair instead
// of ~44.  For phases in which every wave of the workgroup runs the activation and the matrix pipe is idle anyway (the GELU
// pass between the two feed-forward GEMMs, rowblock_kernel.h); beside MFMAs the packed forms issue slower than scalar pairs.
typedef float jv_pk2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ jv_pk2 erf_fast2(const jv_pk2 a) {
  auto bc = [](const float x) { return jv_pk2{x, x}; };
  auto fma2 = [](const jv_pk2 x, const jv_pk2 y, const jv_pk2 z) { return __builtin_elementwise_fma(x, y, z); };
  const jv_pk2 t = __builtin_elementwise_abs(a), s = a * a;
  jv_pk2 r = fma2(bc(-1.72853470e-5f), t, bc(3.83197126e-4f));
  const jv_pk2 u = fma2(bc(-3.88396438e-3f), t, bc(2.42546219e-2f));
  r = fma2(r, s, u);
  r = fma2(r, t, bc(-1.06777877e-1f));
  r = fma2(r, t, bc(-6.34846687e-1f));
  r = fma2(r, t, bc(-1.28717512e-1f));
  r = fma2(r, t, -t);
  r = r * 1.44269504088896340736f;
  jv_pk2 ex = {__builtin_amdgcn_exp2f(r[0]), __builtin_amdgcn_exp2f(r[1])};
  ex = 1.0f - ex;
  const jv_pk2 big = {copysignf(ex[0], a[0]), copysignf(ex[1], a[1])};
  jv_pk2 q = bc(-5.96761703e-4f);
  q = fma2(q, s, bc(4.99119423e-3f));
  q = fma2(q, s, bc(-2.67681349e-2f));
  q = fma2(q, s, bc(1.12819925e-1f));
  q = fma2(q, s, bc(-3.76125336e-1f));
  q = fma2(q, s, bc(1.28379166e-1f));
  const jv_pk2 small = fma2(q, a, a);
  return jv_pk2{t[0] > 0.927734375f ? big[0] : small[0], t[1] > 0.927734375f ? big[1] : small[1]};
}
__device__ __forceinline__ jv_pk2 gelu_erf2(const jv_pk2 v) { return 0.5f * v * (1.f + erf_fast2(v * 0.70710678118654752440f)); }

// sum over the 64 lanes of a wave, returned in every lane: four DPP butterflies inside each row of 16 lanes (VALU only),
// then the four row totals through v_readlane -- no LDS round trips (__shfl_xor is ds_bpermute: six dependent ones)
__device__ __forceinline__ float wave_sum(float x) {
  auto dpp = [](float v, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  x += dpp(x, std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
  x += dpp(x, std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
  x += dpp(x, std::integral_constant<int, 0x141>{});     // row_half_mirror
  x += dpp(x, std::integral_constant<int, 0x140>{});     // row_mirror
  const int b = __builtin_bit_cast(int, x);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return (r0 + r1) + (r2 + r3);
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
  asm volatile("" : "+v"(one));
  const float r0 = fmaf(x0, one, -(float)h[0]), r1 = fmaf(x1, one, -(float)h[1]);
  const jv_f32x2 r = {r0, r1};
  const jv_f16x2 l = __builtin_convertvector(r, jv_f16x2);
  return {__builtin_bit_cast(unsigned, h), __builtin_bit_cast(unsigned, l)};
}

}  // namespace jv
