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
#include <stdio.h>
#include <stdlib.h>

#include "jv_common.h"
#include "jv_device.h"

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
  // flat grid, remapped so that each XCD gets a contiguous run of (b, h, q-tile) triples with the q-tile fastest: the
  // q-tiles of one (b, h) then share that head's K/V in one L2 (round 1: FETCH_SIZE was 2.3x the algorithmic bytes)
  int b, h, qt;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
    const int lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    const int nqt = (p.L + 32 * NW - 1) / (32 * NW);
    qt = lid % nqt;
    h = (lid / nqt) % p.H;
    b = lid / (nqt * p.H);
  }
  const int q0 = qt * 32 * NW + wave * 32;
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

  // chunk-causal (streaming) mask: this lane's query sees keys < kend; the workgroup only walks key tiles that some
  // query of it can see (jyutvoice/utils/mask.py:91-126: static chunks, all left chunks)
  int kend = len, kmax = len;
  if (p.chunk > 0) {
    kend = min(len, ((q0 + r32) / p.chunk + 1) * p.chunk);
    kmax = min(len, ((qt * 32 * NW + 32 * NW - 1) / p.chunk + 1) * p.chunk);
  }
  const int nkt = (kmax + 31) >> 5;
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
      s[e] = key < kend ? s[e] : -INFINITY;
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

// ---- split-plane variant: same algorithm, contractions as six bf16 MFMA products of 3-plane splits (NP = 3) or, where the
// caller has proven the range of q, k, v (AttnArgs::q_scale), three fp16 products of 2-plane splits (NP = 2) ------------
// K is split while it is staged ([NP][32 keys][64 d]); V is split AND transposed to [NP][64 d][32 keys] so that the PV
// product's A operand is 8 consecutive keys of one d; the key order inside a tile is the
// one the S^T accumulator already has (register e of lane-half h holds key (e&3)+8(e>>2)+4h), i.e. position p in the V^T
// row holds key swap_bits23(p): P goes from the softmax to the MFMA with three conversions and no data movement.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// unpadded tiles [NP planes][32 keys][128 B] and [NP][64 d][64 B]; slot keys: a ds_read_b128 lane group
// holds every key parity twice per value of (key >> 1) & 7, and every 64-byte quadrant once per value of (row >> 2) & 3
constexpr int XK_PLANE = 32 * 128, XV_PLANE = 64 * 64;
__device__ __forceinline__ int xk_swz(int key) { return (key >> 1) & 7; }
__device__ __forceinline__ int xv_swz(int row) { return (row >> 2) & 3; }

// combine a value with its partner lane's (lane ^ 32: the two lanes that hold one query's keys) in one VALU op:
// v_permlane32_swap_b32 (gfx950) returns {own half | partner's low half, partner's high half | own}; ds_bpermute, what
// __shfl_xor compiles to, is an LDS round trip
__device__ __forceinline__ float half_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// NP = 3: six bf16 products of (h, m, l) planes; NP = 2: three fp16 products of (h, l) planes (conv_gemm_x6.hip)
template <int NP>
__device__ __forceinline__ f32x16 mfma_planes(const u32x4 (&a)[NP], const u32x4 (&b)[NP], f32x16 c) {
  if constexpr (NP == 2) {
    auto mm = [&](const u32x4& x, const u32x4& y) {
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0);
    };
    mm(a[1], b[0]);
    mm(a[0], b[1]);
    mm(a[0], b[0]);
  } else {
    auto mm = [&](const u32x4& x, const u32x4& y) {
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0);
    };
    mm(a[2], b[0]);
    mm(a[0], b[2]);
    mm(a[1], b[1]);
    mm(a[1], b[0]);
    mm(a[0], b[1]);
    mm(a[0], b[0]);
  }
  return c;
}

// one pair of fp32 values -> NP plane dwords (bf16 h, m, l or fp16 h, l)
template <int NP>
__device__ __forceinline__ void split_pair(const float x0, const float x1, unsigned (&o)[NP]) {
  if constexpr (NP == 2) {
    const Split2 t = split2h_pair(x0, x1);
    o[0] = t.h; o[1] = t.l;
  } else {
    const Split3 t = split3_pair(x0, x1);
    o[0] = t.h; o[1] = t.m; o[2] = t.l;
  }
}

// NP = 2 (fp16x3): q, k, v are multiplied by the exact powers of two p.q_scale, p.k_scale, p.v_scale, which the caller
// chose from PROVEN bounds so that nothing reaches fp16's 65504; the scores are scaled back inside the exp2 argument, the
// probabilities are kept as p * 2^10 (<= 1024; the factor cancels in the normalisation), and v_scale leaves with 1 / l.
template <int NW, int WPE, int NP>
__global__ __launch_bounds__(64 * NW, WPE) void attn64_x6_kernel(const AttnArgs p) {
  constexpr int XK_TILE = NP * XK_PLANE, XV_TILE = NP * XV_PLANE;
  // Two buffers per operand, one barrier per key tile: tile kt + 1 is split and stored into the other buffer while tile kt
  // is being used.  Rows are unpadded (K: 128 B = 64 d, V^T: 64 B = 32 keys) with XOR-swizzled 16-byte slots, which keeps
  // the ds_read_b128 fragment fetches conflict-free and the two buffers within 48 KiB (three workgroups per CU).
  __shared__ __attribute__((aligned(256))) unsigned char ldsKV[2 * (XK_TILE + XV_TILE)];
  constexpr int NT = 64 * NW;
  constexpr int NKL = (32 * 16 + NT - 1) / NT;   // f32x4 pieces of K per thread per tile (512 pieces over the workgroup)
  constexpr int NVG = (8 + NW - 1) / NW;         // groups of 4 consecutive keys of V per thread per tile (8 groups over the waves)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, half = lane >> 5;
  // flat grid, remapped so that each XCD gets a contiguous run of (b, h, q-tile) triples with the q-tile fastest: the
  // q-tiles of one (b, h) then share that head's K/V in one L2 (round 1: FETCH_SIZE was 2.3x the algorithmic bytes)
  int b, h, qt;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
    const int lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    const int nqt = (p.L + 32 * NW - 1) / (32 * NW);
    qt = lid % nqt;
    h = (lid / nqt) % p.H;
    b = lid / (nqt * p.H);
  }
  const int q0 = qt * 32 * NW + wave * 32;
  const int len = p.lens ? min(p.lens[b], p.L) : p.L;
  const long rowbase = (long)p.G + (long)b * p.S;
  const bool active = q0 < p.L;

  // Q planes: lane (query, half) holds d = 16 s + 8 half + j for k-step s, pre-scaled by log2(e)/8
  u32x4 q[4][NP];
  const float qsc = 0.125f * 1.44269504088896340736f * (NP == 2 ? p.q_scale : 1.f);
  const float ksc = NP == 2 ? p.k_scale : 1.f, vsc = NP == 2 ? p.v_scale : 1.f;
  const float sinv = NP == 2 ? 1.0f / (p.q_scale * p.k_scale) : 1.f;      // powers of two: exact
  constexpr float pshift = NP == 2 ? 10.f : 0.f;
  {
    const int qi = q0 + r32;
    const float* src = p.qkv + (rowbase + qi) * p.ld + h * 64 + 8 * half;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0;
      if (qi < p.L) {
        t0 = *reinterpret_cast<const f32x4*>(src + 16 * s);
        t1 = *reinterpret_cast<const f32x4*>(src + 16 * s + 4);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x0 = (e < 2 ? t0[2 * e] : t1[2 * e - 4]) * qsc, x1 = (e < 2 ? t0[2 * e + 1] : t1[2 * e - 3]) * qsc;
        unsigned o[NP];
        split_pair<NP>(x0, x1, o);
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) q[s][pl][e] = o[pl];
      }
    }
  }

  f32x16 o0, o1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;

  f32x4 pk[NKL];
  float pv[NVG][4];
  const int vd = tid & 63, vkg = tid >> 6;
  // Rows past the last valid key are clamped to it instead of branched around: their scores are masked to -inf below, so
  // P is exactly 0 there and whatever finite K/V values were staged do not matter.
  const float* const kbase = p.qkv + rowbase * p.ld + p.k_off + h * 64;    // wave-uniform bases, 32-bit offsets per lane
  const float* const vbase = p.qkv + rowbase * p.ld + p.v_off + h * 64 + vd;
  const unsigned uld = (unsigned)p.ld;
  auto prefetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NKL; ++i) {
      const int idx = tid + i * NT;
      if ((512 % NT) == 0 || idx < 512) {
        const unsigned key = (unsigned)min(k0 + (idx >> 4), len - 1);
        pk[i] = *reinterpret_cast<const f32x4*>(kbase + (key * uld + 4u * (idx & 15)));
      }
    }
#pragma unroll
    for (int j = 0; j < NVG; ++j) {
      const int g = vkg + j * NW;
      if ((8 % NW) == 0 || g < 8) {
#pragma unroll
        for (int e = 0; e < 4; ++e) pv[j][e] = vbase[(unsigned)min(k0 + 4 * g + e, len - 1) * uld];
      }
    }
  };

  // chunk-causal (streaming) mask: this lane's query sees keys < kend; the workgroup only walks key tiles that some
  // query of it can see (jyutvoice/utils/mask.py:91-126: static chunks, all left chunks)
  int kend = len, kmax = len;
  if (p.chunk > 0) {
    kend = min(len, ((q0 + r32) / p.chunk + 1) * p.chunk);
    kmax = min(len, ((qt * 32 * NW + 32 * NW - 1) / p.chunk + 1) * p.chunk);
  }
  const int nkt = (kmax + 31) >> 5;
  auto stage = [&](int buf) {      // split the prefetched K / V tile into the three bf16 planes of buffer `buf`
    unsigned char* const bK = ldsKV + buf * (XK_TILE + XV_TILE);
    unsigned char* const bV = bK + XK_TILE;
#pragma unroll
    for (int i = 0; i < NKL; ++i) {
      const int idx = tid + i * NT;
      if ((512 % NT) != 0 && idx >= 512) continue;
      const int key = idx >> 4, c4 = idx & 15;
      unsigned o0_[NP], o1_[NP];
      if constexpr (NP == 2) {
        split_pair<NP>(pk[i][0] * ksc, pk[i][1] * ksc, o0_);
        split_pair<NP>(pk[i][2] * ksc, pk[i][3] * ksc, o1_);
      } else {
        split_pair<NP>(pk[i][0], pk[i][1], o0_);
        split_pair<NP>(pk[i][2], pk[i][3], o1_);
      }
      unsigned char* dst = bK + key * 128 + ((((c4 >> 1) ^ xk_swz(key)) << 4) | ((c4 & 1) << 3));
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<u32x2*>(dst + pl * XK_PLANE) = u32x2{o0_[pl], o1_[pl]};
    }
#pragma unroll
    for (int j = 0; j < NVG; ++j) {             // 4 consecutive keys stay consecutive under the bit swap
      const int g = vkg + j * NW;
      if ((8 % NW) != 0 && g >= 8) continue;
      const int key = 4 * g;
      const int pos = (key & 0x13) | ((key & 4) << 1) | ((key & 8) >> 1);       // bf16 index inside the 32-key row
      unsigned o0_[NP], o1_[NP];
      if constexpr (NP == 2) {
        split_pair<NP>(pv[j][0] * vsc, pv[j][1] * vsc, o0_);
        split_pair<NP>(pv[j][2] * vsc, pv[j][3] * vsc, o1_);
      } else {
        split_pair<NP>(pv[j][0], pv[j][1], o0_);
        split_pair<NP>(pv[j][2], pv[j][3], o1_);
      }
      unsigned char* dst = bV + vd * 64 + ((((pos >> 3) ^ xv_swz(vd)) << 4) | (((pos >> 2) & 1) << 3));
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<u32x2*>(dst + pl * XV_PLANE) = u32x2{o0_[pl], o1_[pl]};
    }
  };
  if (nkt > 0) {
    prefetch(0);
    stage(0);
    if (nkt > 1) prefetch(32);
    __syncthreads();
  }
  for (int kt = 0; kt < nkt; ++kt) {
    const int k0 = kt * 32;
    // buffer kt & 1 holds tile kt (stored before the previous barrier); registers hold tile kt + 1
    if (kt + 1 < nkt) {
      stage((kt + 1) & 1);                      // that buffer was last read in iteration kt - 1, before the barrier
      if (kt + 2 < nkt) prefetch(k0 + 64);
    }
    const unsigned char* const ldsK = ldsKV + (kt & 1) * (XK_TILE + XV_TILE);
    const unsigned char* const ldsV = ldsK + XK_TILE;
    if (active) {
    // S^T[key][query] = sum_d K[key][d] * Q[query][d]
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
    const unsigned char* kr = ldsK + r32 * 128;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      u32x4 a[NP];
      const int ko = ((2 * st + half) ^ xk_swz(r32)) << 4;
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) a[pl] = *reinterpret_cast<const u32x4*>(kr + pl * XK_PLANE + ko);
      if (!JV_ABLATE(p, 16)) s = mfma_planes<NP>(a, q[st], s);
      else s[st] += __uint_as_float(a[0][0] ^ a[1][1]);
    }
    if (!JV_ABLATE(p, 32)) {
    if (__builtin_amdgcn_ballot_w64(k0 + 32 > kend) != 0) {     // wave-uniform: only tiles that straddle a mask edge
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = k0 + (e & 3) + 8 * (e >> 2) + 4 * half;
        s[e] = key < kend ? s[e] : -INFINITY;
      }
    }
    float mt = s[0];
#pragma unroll
    for (int e = 1; e < 16; ++e) mt = fmaxf(mt, s[e]);
    mt = half_max(mt) * sinv;
    const float m_new = fmaxf(m_run, mt);
    // v_exp_f32 directly: exp2f() wraps it in denormal-range scaling (6 more VALU ops per score); weights below 2^-126
    // flush to zero instead, far under the fp32 rounding of the row sum
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float lt = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      s[e] = __builtin_amdgcn_exp2f(NP == 2 ? fmaf(s[e], sinv, pshift - m_new) : s[e] - m_new);
      lt += s[e];
    }
    lt = half_sum(lt);
    l_run = l_run * alpha + lt;
    m_run = m_new;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {        // the running maximum moved for some query of this wave
#pragma unroll
      for (int e = 0; e < 16; ++e) { o0[e] *= alpha; o1[e] *= alpha; }
    }
    }
    // O^T[d][query] += sum_key V[key][d] * P[query][key]
    const unsigned char* vr = ldsV + r32 * 64;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      u32x4 pb[NP];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned o[NP];
        split_pair<NP>(s[8 * st + 2 * e], s[8 * st + 2 * e + 1], o);
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) pb[pl][e] = o[pl];
      }
      u32x4 a[NP];
      const int vo = ((2 * st + half) ^ xv_swz(r32)) << 4;       // rows r32 and r32 + 32 share the key
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) a[pl] = *reinterpret_cast<const u32x4*>(vr + pl * XV_PLANE + vo);
      if (!JV_ABLATE(p, 8)) o0 = mfma_planes<NP>(a, pb, o0);
      else o0[st] += __uint_as_float(a[0][0] ^ a[1][1] ^ pb[0][0] ^ pb[1][1]);
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) a[pl] = *reinterpret_cast<const u32x4*>(vr + (pl * 64 + 32) * 64 + vo);
      if (!JV_ABLATE(p, 8)) o1 = mfma_planes<NP>(a, pb, o1);
      else o1[st] += __uint_as_float(a[0][0] ^ a[1][1]);
    }
    }
    // the one barrier of the tile: tile kt + 1 is complete in the other buffer, and nobody reads this one any more
    if (!JV_ABLATE(p, 4)) __syncthreads();
  }

  const int qi = q0 + r32;
  if (active && qi < p.L) {
    const float inv = l_run > 0.f ? (1.0f / vsc) / l_run : 0.f;
    float* dst = p.out + (rowbase + qi) * p.ldo + h * 64 + 4 * half;
    if (p.out2) {      // wave-uniform: the consumer is an fp16x3 GEMM that takes its A operand pre-split
      unsigned short* d2 = p.out2 + (rowbase + qi) * p.ldo + h * 64 + 4 * half;
      const float sc = p.out2_scale;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const Split2 a0 = split2h_pair(o0[4 * g] * inv * sc, o0[4 * g + 1] * inv * sc);
        const Split2 a1 = split2h_pair(o0[4 * g + 2] * inv * sc, o0[4 * g + 3] * inv * sc);
        const Split2 c0 = split2h_pair(o1[4 * g] * inv * sc, o1[4 * g + 1] * inv * sc);
        const Split2 c1 = split2h_pair(o1[4 * g + 2] * inv * sc, o1[4 * g + 3] * inv * sc);
        *reinterpret_cast<u32x2*>(d2 + 8 * g) = u32x2{a0.h, a1.h};
        *reinterpret_cast<u32x2*>(d2 + 8 * g + p.out2_plane) = u32x2{a0.l, a1.l};
        *reinterpret_cast<u32x2*>(d2 + 32 + 8 * g) = u32x2{c0.h, c1.h};
        *reinterpret_cast<u32x2*>(d2 + 32 + 8 * g + p.out2_plane) = u32x2{c0.l, c1.l};
      }
      return;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 a = {o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv};
      const f32x4 c = {o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv};
      *reinterpret_cast<f32x4*>(dst + 8 * g) = a;
      *reinterpret_cast<f32x4*>(dst + 32 + 8 * g) = c;
    }
  }
}

namespace {

template <int NW>
void launch_x6(const AttnArgs& a, hipStream_t st) {
  const dim3 grid(cdiv(a.L, 32 * NW) * a.H * a.B);
  if (a.q_scale > 0.f) hipLaunchKernelGGL((attn64_x6_kernel<NW, (NW == 2 ? 2 : 3), 2>), grid, dim3(64 * NW), 0, st, a);
  else hipLaunchKernelGGL((attn64_x6_kernel<NW, (NW == 2 ? 2 : 3), 3>), grid, dim3(64 * NW), 0, st, a);
}

}  // namespace

int attention64(const AttnArgs& a, hipStream_t st) {
  if (a.B <= 0 || a.L <= 0) return JV_OK;
  if ((a.ld & 3) || (a.ldo & 3) || (a.k_off & 3) || (a.v_off & 3))
    return fail(JV_ERR_ARG, "attention64: strides/offsets must be multiples of 4 floats");
  static const bool fp32_path = getenv("JV_ATTN_FP32") != nullptr;
  if (a.q_scale > 0.f && !(a.k_scale > 0.f && a.v_scale > 0.f)) return fail(JV_ERR_ARG, "attention64: fp16x3 needs all three scales");
  if (const char* ab = tuning_env("JV_ABLATE")) const_cast<AttnArgs&>(a).ablate = atoi(ab);
  const bool prof = prof_on();
  if (prof) prof_begin(st);
  const char* name = "";
  if (fp32_path) {
    // 64-query workgroups waste least on T = 300 (5 x 64); 128-query ones halve K/V staging at T = 512
    const int waste2 = round_up(a.L, 64) - a.L, waste4 = round_up(a.L, 128) - a.L;
    if (waste4 <= waste2) hipLaunchKernelGGL((attn64_kernel<4>), dim3(cdiv(a.L, 128) * a.H * a.B), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((attn64_kernel<2>), dim3(cdiv(a.L, 64) * a.H * a.B), dim3(128), 0, st, a);
    name = waste4 <= waste2 ? "attn64<4 waves>" : "attn64<2 waves>";
  } else {
    // One wave per 32 queries; every workgroup stages (splits, transposes) the K/V tiles of its head itself, so larger
    // workgroups repeat less staging -- but only wave counts that fill the CU's four SIMDs evenly pay (measured at
    // T = 300: 5-wave workgroups 174 us against 110 us for 4-wave ones; T = 512: 8 waves 61 us, 4 waves 67 us).
    int nw = a.L <= 64 ? 2 : (round_up(a.L, 256) == round_up(a.L, 128) ? 8 : 4);
    if (const char* f = tuning_env("JV_ATTN_NW")) nw = atoi(f);
    switch (nw) {
      case 2: launch_x6<2>(a, st); break;
      case 4: launch_x6<4>(a, st); break;
      case 8: launch_x6<8>(a, st); break;
      default: return fail(JV_ERR_ARG, "attention64: bad wave count");
    }
    static const char* const names[9] = {"", "", "attn64_x6<2 waves>", "attn64_x6<3 waves>", "attn64_x6<4 waves>", "attn64_x6<5 waves>",
                                         "attn64_x6<6 waves>", "attn64_x6<7 waves>", "attn64_x6<8 waves>"};
    static const char* const names_h3[9] = {"", "", "attn64_h3<2 waves>", "", "attn64_h3<4 waves>", "", "", "", "attn64_h3<8 waves>"};
    name = a.q_scale > 0.f ? names_h3[nw] : names[nw];
  }
  if (prof) {
    // algorithmic (full-length) figure of SURVEY.md 8(d): QK^T + PV = 4*L*L*64 per head; q,k,v,o once
    const double bh = (double)a.B * a.H;
    prof_end(st, name, 4.0 * bh * a.L * a.L * 64.0, 4.0 * bh * a.L * 64.0 * 4.0);
  }
  JV_HIP(hipGetLastError());
  return JV_OK;
}

}  // namespace jv
