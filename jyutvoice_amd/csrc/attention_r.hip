// fp16x3 self-attention of the estimator, third form: ONE workgroup per (utterance, head, 64 QT queries), each of its four
// waves owning 16 QT queries (QT = 5: 80 queries per wave, 320 per workgroup -- a whole 300-frame head) against every key.
// jyutvoice/flow/transformer.py:380-389 -> diffusers AttnProcessor2_0 as restated in oracle/flow.py:46-58.
//
// Why a third kernel.  attn64_pl_kernel (attention_pl.hip) gives a wave ONE 32-query tile: per 32-key tile it reads the whole
// K / V tile from LDS for 24 MFMAs, passes a barrier, issues its share of the next tile's DMA, and runs a softmax whose
// instructions cannot overlap the wave's own MFMAs -- measured (profiles/r02_pmc_attention_pl.md) 10.7 vector instructions
// per MFMA, the matrix pipe 27 % busy; and a 300-frame head is 10 such tiles = 2.5 four-wave workgroups, so a launch is 1.5
// rounds of workgroups with the last one per head mostly idle.  Here:
//   * a wave's K fragment (16 keys x 32 d) and V^T fragment (16 d x 32 keys) are read ONCE per key tile and used for all QT
//     query tiles: LDS reads, barriers, DMA issues and loop overhead per MFMA drop QT-fold;
//   * v_mfma_f32_16x16x32_f16: 16-query granularity is what lets 80-query waves exist (300 = 3.75 x 80); S^T[key][query]
//     again, so softmax statistics are per lane (query = lane & 15) and P is directly the B operand of O^T = V^T P^T:
//     lane group kq = lane >> 4 holds keys 4 kq + e of the two 16-key blocks, which is the key order the transposing
//     ds_read_b64_tr_b16 delivers V^T in;
//   * the row sum of P stays a per-lane partial (this lane's keys) until the end: one cross-lane reduction per query and
//     launch instead of one per key tile; only the running maximum crosses lanes per tile (two row / half exchanges);
//   * 512 workgroups at 32 x 300 frames = two per CU, one round.
// Arithmetic: attention_pl.hip's (scales, base-2 softmax on raw v_exp_f32, P kept as p * 2^10, products hh' + hl' + lh'
// smallest first); the contraction over d runs in two 32-deep MFMAs instead of four 16-deep ones and the row sums are
// grouped per lane, so results agree with that kernel to rounding, not bit for bit.
#include <math.h>
#include <stdlib.h>

#include "jv_common.h"
#include "jv_device.h"

namespace jv {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int AR_PLANE = 32 * 128;                 // one plane of a 32-key tile: 128 bytes (64 d) per key
constexpr int AR_STAGE = 4 * AR_PLANE;             // K h, K l, V h, V l
constexpr int AR_NW = 4;

// 16-byte slot keys of a key row (128 B = 8 slots), brute-forced against the LDS lane groups of MI355X_MICROARCH.md:
// K is read by rows (ds_read_b128: 16 keys x 4 consecutive slots per instruction), V transposed (ds_read_b64_tr_b16: a
// 32-lane half takes 8 keys x 32 contiguous bytes) -- both conflict-free with these
__device__ __forceinline__ int ark_swz(int key) { return (key >> 1) & 7; }
__device__ __forceinline__ int arv_swz(int key) { return ((key >> 1) & 3) << 1; }

__device__ __forceinline__ f32x4 ar_mfma3(const u32x4 (&a)[2], const u32x4 (&b)[2], f32x4 c) {
  auto mm = [&](const u32x4& x, const u32x4& y) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0);
  };
  mm(a[1], b[0]);      // smallest terms first, as everywhere
  mm(a[0], b[1]);
  mm(a[0], b[0]);
  return c;
}
// max / sum over the four lanes that share a query (lane, lane ^ 16, lane ^ 32, lane ^ 48), in every one of them
__device__ __forceinline__ float ar_q_max(float v) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float ar_q_sum(float v) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
template <int N>
__device__ __forceinline__ void ar_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void ar_barrier() { asm volatile("s_barrier" ::: "memory"); }
__device__ __forceinline__ void ar_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int QT, int NST>
__global__ __launch_bounds__(64 * AR_NW, 2) void attn64_r_kernel(const AttnArgs p) {
  // the K / V ring; after the key loop each wave's [2 planes][16 queries][128 B] output patch (4 KB per wave)
  __shared__ __attribute__((aligned(256))) unsigned char lds[NST * AR_STAGE];
  static_assert(NST * AR_STAGE >= AR_NW * 2 * 16 * 128, "the output patches fit the ring");
  constexpr int PPW = 16 / AR_NW;      // DMA pieces per wave and key tile
  constexpr int QW = 16 * QT;          // queries per wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
  const int nqt = (p.L + AR_NW * QW - 1) / (AR_NW * QW);      // workgroups per head
  const int qt = blockIdx.x % nqt;
  const int h = (blockIdx.x / nqt) % p.H;
  const int b = blockIdx.x / (nqt * p.H);
  const int q0 = qt * AR_NW * QW + wave * QW;
  const int len = p.lens ? min(p.lens[b], p.L) : p.L;
  const long rowbase = (long)p.G + (long)b * p.S;
  const bool active = q0 < p.L;

  // ---- this wave's DMA pieces of a key tile (attention_pl.hip): piece pc = wave + 4 i -> operand pc >> 3 (K, V), plane
  // (pc >> 2) & 1, 8-key group pc & 3; lane L lands on key 8 g + (L >> 3), slot L & 7 and fetches the slot the read side maps there
  const unsigned short* src[PPW];
  int dst[PPW], krel[PPW];
  const int kstride = p.kv_ld;
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int pc = wave + AR_NW * i;
    const int opv = pc >> 3, pl = (pc >> 2) & 1, g = pc & 3;
    const int key = 8 * g + (lane >> 3);
    const int slot = (lane & 7) ^ (opv ? arv_swz(key) : ark_swz(key));
    krel[i] = key;
    src[i] = p.kv2 + (long)pl * p.kv2_plane + rowbase * kstride + opv * 512 + h * 64 + 8 * slot;
    dst[i] = (2 * opv + pl) * AR_PLANE + g * 1024;
  }
  auto issue = [&](int k0, int stage) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      // rows past the last valid key are clamped to it: their scores are masked to -inf below, P is exactly 0 there
      const int key = min(k0 + krel[i], len - 1);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (long)key * kstride),
                                       (__attribute__((address_space(3))) void*)(lds + stage * AR_STAGE + dst[i]), 16, 0, 0);
    }
  };
  const int nkt = (len + 31) >> 5;
  if (nkt > 0) issue(0, 0);
  if (NST > 2 && nkt > 1) issue(32, 1);

  // ---- Q planes: B operand of S^T = K Q^T.  Lane (query r16, kq) holds d = 32 ds + 8 kq + j, pre-scaled by log2(e) / 8 * q_scale
  u32x4 qf[QT][2][2];
  const float qsc = 0.125f * 1.44269504088896340736f * p.q_scale;
  const float sinv = 1.0f / (p.q_scale * p.k_scale);      // powers of two: exact
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int qi = q0 + 16 * t + r16;
    const float* qs = p.qkv + (rowbase + qi) * p.ld + h * 64 + 8 * kq;
#pragma unroll
    for (int ds = 0; ds < 2; ++ds) {
      f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0;
      if (qi < p.L) {
        t0 = *reinterpret_cast<const f32x4*>(qs + 32 * ds);
        t1 = *reinterpret_cast<const f32x4*>(qs + 32 * ds + 4);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x0 = (e < 2 ? t0[2 * e] : t1[2 * e - 4]) * qsc, x1 = (e < 2 ? t0[2 * e + 1] : t1[2 * e - 3]) * qsc;
        const Split2 sp = split2h_pair(x0, x1);
        qf[t][ds][0][e] = sp.h;
        qf[t][ds][1][e] = sp.l;
      }
    }
  }

  f32x4 o[QT][4];      // O^T[d = 16 db + 4 kq + e][query r16]
  float m_run[QT], l_run[QT];      // l_run: THIS LANE's keys only (reduced over the four lanes of a query at the end)
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    m_run[t] = -INFINITY;
    l_run[t] = 0.f;
#pragma unroll
    for (int db = 0; db < 4; ++db) o[t][db] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // per-lane read offsets inside a stage
  const int k_row = r16 * 128;                                   // K: + kb * 2048 + pl * AR_PLANE + slot
  const int vq = r16 >> 2, vp = lane & 3;                        // V^T: key within its 4-block, 8-byte piece (4 d)
  for (int kt = 0; kt < nkt; ++kt) {
    const int k0 = kt * 32;
    if (NST > 2 && kt + 1 < nkt) ar_wait_vmcnt<PPW>(); else ar_wait_vmcnt<0>();
    ar_barrier();
    if (NST > 2) { if (kt + 2 < nkt) issue(k0 + 64, (kt + 2) % NST); }
    else if (kt + 1 < nkt) issue(k0 + 32, (kt + 1) % NST);
    if (!active) continue;
    const unsigned char* const sK = lds + (kt % NST) * AR_STAGE;
    const unsigned char* const sV = sK + 2 * AR_PLANE;

    // ---- S^T[key][query] = sum_d K[key][d] Q[query][d]: each K fragment read once, used by all QT query tiles
    f32x4 s[QT][2];
#pragma unroll
    for (int t = 0; t < QT; ++t) { s[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; s[t][1] = s[t][0]; }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ds = 0; ds < 2; ++ds) {
        u32x4 a[2];
        const int key = 16 * kb + r16;
        const int ko = ((4 * ds + kq) ^ ark_swz(key)) << 4;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) a[pl] = *reinterpret_cast<const u32x4*>(sK + pl * AR_PLANE + kb * 2048 + k_row + ko);
#pragma unroll
        for (int t = 0; t < QT; ++t) s[t][kb] = ar_mfma3(a, qf[t][ds], s[t][kb]);
      }

    // ---- softmax (base 2), per query tile; P leaves as the B operand of the PV product
    const bool edge = k0 + 32 > len;      // (wave-uniform) only the last tile can straddle the key mask
    u32x4 pb[QT][2];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      if (edge) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int e = 0; e < 4; ++e) s[t][kb][e] = (k0 + 16 * kb + 4 * kq + e) < len ? s[t][kb][e] : -INFINITY;
      }
      float mt = fmaxf(fmaxf(fmaxf(s[t][0][0], s[t][0][1]), fmaxf(s[t][0][2], s[t][0][3])),
                       fmaxf(fmaxf(s[t][1][0], s[t][1][1]), fmaxf(s[t][1][2], s[t][1][3])));
      mt = ar_q_max(mt) * sinv;
      const float m_new = fmaxf(m_run[t], mt);
      const float alpha = __builtin_amdgcn_exp2f(m_run[t] - m_new);
      float lt = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s[t][kb][e] = __builtin_amdgcn_exp2f(fmaf(s[t][kb][e], sinv, 10.f - m_new));      // p * 2^10 (cancels in 1 / l)
          lt += s[t][kb][e];
        }
      l_run[t] = l_run[t] * alpha + lt;
      m_run[t] = m_new;
      if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
        for (int db = 0; db < 4; ++db) o[t][db] = o[t][db] * alpha;
      }
      // k-slot j of lane group kq: j < 4 -> key 4 kq + j of block 0, j >= 4 -> key 4 kq + j - 4 of block 1
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const Split2 sp = split2h_pair(s[t][kb][e], s[t][kb][e + 1]);
          pb[t][0][2 * kb + (e >> 1)] = sp.h;
          pb[t][1][2 * kb + (e >> 1)] = sp.l;
        }
    }

    // ---- O^T[d][query] += sum_key V[key][d] P[query][key]: each V^T fragment read once (transposing read), used by all tiles
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      u32x4 a[2];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
          const int key = 16 * rd + 4 * kq + vq;
          const int byte = (16 * db + 4 * vp) * 2;
          const int off = key * 128 + ((((byte >> 4) ^ arv_swz(key)) << 4) | (byte & 15));
          const fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
              (__attribute__((address_space(3))) fp16x4*)(const_cast<unsigned char*>(sV) + pl * AR_PLANE + off));
          const u32x2 u = __builtin_bit_cast(u32x2, v);
          a[pl][2 * rd] = u[0];
          a[pl][2 * rd + 1] = u[1];
        }
#pragma unroll
      for (int t = 0; t < QT; ++t) o[t][db] = ar_mfma3(a, pb[t], o[t][db]);
    }
  }

  // ---- result: O / l as the output projection's operand (fp16 planes), each wave through its own 4 KB patch of the idle
  // ring so that every store instruction writes whole 128-byte rows
  ar_lds_barrier();      // every wave is done with the ring
  if (!active) return;
  unsigned char* const so = lds + wave * (2 * 16 * 128);      // [2 planes][16 queries][128 B]
  const float vsc = p.v_scale;
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const float l = ar_q_sum(l_run[t]);
    const float inv = l > 0.f ? (1.0f / vsc) / l : 0.f;
    const int qbase = q0 + 16 * t;
    if (qbase >= p.L) break;      // (wave-uniform)
    if (p.out2) {
      const float sc = inv * p.out2_scale;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const Split2 a0 = split2h_pair(o[t][db][0] * sc, o[t][db][1] * sc);
        const Split2 a1 = split2h_pair(o[t][db][2] * sc, o[t][db][3] * sc);
        unsigned char* dp = so + r16 * 128 + (16 * db + 4 * kq) * 2;      // 4 consecutive d: 8 bytes
        *reinterpret_cast<u32x2*>(dp) = u32x2{a0.h, a1.h};
        *reinterpret_cast<u32x2*>(dp + 16 * 128) = u32x2{a0.l, a1.l};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (a wave's LDS accesses execute in order; the reads below see these writes)
      // 8 lanes per row of 128 B, 8 rows per instruction: 2 planes x 16 rows = 4 instructions
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int pl = it >> 1, qrow = (it & 1) * 8 + (lane >> 3), piece = lane & 7;
        const int qi = qbase + qrow;
        const u32x4 v = *reinterpret_cast<const u32x4*>(so + (pl * 16 + qrow) * 128 + piece * 16);
        if (qi < p.L) *reinterpret_cast<u32x4*>(p.out2 + (long)pl * p.out2_plane + (rowbase + qi) * p.ldo + h * 64 + piece * 8) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the patch is rewritten by the next tile
    } else {
      const int qi = qbase + r16;
      if (qi < p.L) {
        float* dp = p.out + (rowbase + qi) * p.ldo + h * 64 + 4 * kq;
#pragma unroll
        for (int db = 0; db < 4; ++db) *reinterpret_cast<f32x4*>(dp + 16 * db) = o[t][db] * inv;
      }
    }
  }
}

template <int QT, int NST>
void launch_r(const AttnArgs& a, hipStream_t st) {
  const int nqt = cdiv(a.L, AR_NW * 16 * QT);
  hipLaunchKernelGGL((attn64_r_kernel<QT, NST>), dim3(nqt * a.H * a.B), dim3(64 * AR_NW), 0, st, a);
}

}  // namespace

// queries per workgroup = 64 QT: the QT in {2 .. 5} that covers L with the fewest padded queries (workgroups x 64 QT), the
// taller tile on a tie (each K / V fragment is then used for more queries).  300 frames -> 5, 512 -> 4, 128 -> 2
int attention64_r_tiles(int L) {
  int best = 5;
  long best_q = 1L << 40;
  for (int qt = 5; qt >= 2; --qt) {
    const long padded = (long)cdiv(L, 64 * qt) * 64 * qt;
    if (padded < best_q) { best = qt; best_q = padded; }
  }
  return best;
}

// same contract as attention64_planes() (attention_pl.hip); no chunk-causal mask (streaming stays on attention64_planes)
int attention64_rows(const AttnArgs& a, hipStream_t st) {
  if (a.B <= 0 || a.L <= 0) return JV_OK;
  if (!a.kv2 || (a.kv_ld & 7) || !(a.q_scale > 0.f && a.k_scale > 0.f && a.v_scale > 0.f) || (a.ld & 3) || (a.ldo & 7) || a.chunk > 0)
    return fail(JV_ERR_ARG, "attention64_rows: needs K/V planes, the three scales, aligned strides, no chunk mask");
  const bool prof = prof_on();
  if (prof) prof_begin(st);
  int qt = attention64_r_tiles(a.L);
  if (const char* f = dyn_env("JV_ATTN_QT")) qt = atoi(f);
  const bool st3 = dyn_env("JV_ATTN_NST2") == nullptr;
  switch (qt) {
    case 2: if (st3) launch_r<2, 3>(a, st); else launch_r<2, 2>(a, st); break;
    case 3: if (st3) launch_r<3, 3>(a, st); else launch_r<3, 2>(a, st); break;
    case 4: if (st3) launch_r<4, 3>(a, st); else launch_r<4, 2>(a, st); break;
    case 5: if (st3) launch_r<5, 3>(a, st); else launch_r<5, 2>(a, st); break;
    default: return fail(JV_ERR_ARG, "attention64_rows: bad tile count");
  }
  if (prof) {
    static const char* const names[6] = {"", "", "attn64_r<128 q>", "attn64_r<192 q>", "attn64_r<256 q>", "attn64_r<320 q>"};
    const double bh = (double)a.B * a.H;
    prof_end(st, names[qt], 4.0 * bh * a.L * a.L * 64.0, 4.0 * bh * a.L * 64.0 * 4.0);
  }
  JV_HIP(hipGetLastError());
  return JV_OK;
}

}  // namespace jv
