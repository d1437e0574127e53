// fp16x3 self-attention of the estimator with K and V arriving PRE-SPLIT (two fp16 planes each, written once by the qkv
// GEMM's epilogue) -- jyutvoice/flow/transformer.py:380-389 -> diffusers AttnProcessor2_0 as restated in oracle/flow.py:46-58.
//
// attn64_x6_kernel (attention.hip) stages K / V through registers and splits them into planes in every one of a head's
// query-tile workgroups: round 1's PMC had its vector pipe 49 % busy against 25 % for the matrix pipe, a quarter of the
// vector instructions being that staging.  Here:
//   * K / V tiles of 32 keys go global -> LDS by LDS-DMA (global_load_lds_dwordx4, 16 one-KiB pieces per tile) into a
//     3-stage ring: counted s_waitcnt vmcnt, one bare s_barrier per key tile, two tiles in flight -- no staging VALU, no
//     staging registers, no ds_write;
//   * V stays row-major [key][d] in LDS (what the DMA can deliver) and the PV product's A operand, 8 consecutive keys of
//     one d, is read with gfx950's transposing ds_read_b64_tr_b16 (tools/probes/ds_read_tr16.hip): the 4-key blocks it
//     returns are exactly the key order the S^T accumulator already has (register e of lane-half h holds key
//     (e & 3) + 8 (e >> 2) + 4 h), so P still goes from the softmax to the MFMA with conversions only;
//   * the result leaves as the output projection's pre-split operand (fp16 planes) through LDS, as whole 128-byte rows.
// Arithmetic (scales, base-2 softmax on raw v_exp_f32, products hh' + hl' + lh', smallest first) is attention.hip's NP = 2
// path term for term.
#include <math.h>
#include <stdlib.h>

#include "jv_common.h"
#include "jv_device.h"

namespace jv {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int PL_PLANE = 32 * 128;                 // one plane of a 32-key tile: 128 bytes (64 d) per key
constexpr int PL_STAGE = 4 * PL_PLANE;             // K h, K l, V h, V l
constexpr int PL_NSTAGE = 3;

// 16-byte slot keys of a key row (128 B = 8 slots).  K is read by rows (ds_read_b128, lane = key): attention.hip's key.
// V is read transposed: a 32-lane half takes 4 consecutive keys x 64 contiguous bytes; swapping the row's 64-byte halves on
// keys with bit 1 set puts the four rows on the four 64-byte quarters of the 256-byte bank row (conflict-free).
__device__ __forceinline__ int plk_swz(int key) { return (key >> 1) & 7; }
__device__ __forceinline__ int plv_swz(int key) { return ((key >> 1) & 1) << 2; }

__device__ __forceinline__ float pl_half_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float pl_half_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ f32x16 pl_mfma3(const u32x4 (&a)[2], const u32x4 (&b)[2], f32x16 c) {
  auto mm = [&](const u32x4& x, const u32x4& y) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0);
  };
  mm(a[1], b[0]);
  mm(a[0], b[1]);
  mm(a[0], b[0]);
  return c;
}
template <int N>
__device__ __forceinline__ void pl_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void pl_barrier() { asm volatile("s_barrier" ::: "memory"); }
__device__ __forceinline__ void pl_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NW, int NST>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 2 : (NST == 2 ? 4 : 3)) void attn64_pl_kernel(const AttnArgs p) {
  // the ring, reused at the end for the [2 planes][32 NW queries][128 B] output image
  constexpr int LDS_BYTES = NST * PL_STAGE > 2 * 32 * NW * 128 ? NST * PL_STAGE : 2 * 32 * NW * 128;
  __shared__ __attribute__((aligned(256))) unsigned char lds[LDS_BYTES];
  constexpr int PPW = 16 / NW;      // DMA pieces per wave and key tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, half = lane >> 5;
  int b, h, qt;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
    const int lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    const int nqt = (p.L + 32 * NW - 1) / (32 * NW);
    // (Tried: a head's partial workgroup -- 44 of 128 queries at 300 frames -- scheduled behind ALL full ones instead of behind its
    // own head's: 36.9 against 34.3 ms per pass on the same box.  The interleaved order lets the short workgroups free slots
    // throughout the launch; at the end they leave the chip a quarter full.)
    qt = lid % nqt;
    h = (lid / nqt) % p.H;
    b = lid / (nqt * p.H);
  }
  const int q0 = qt * 32 * NW + wave * 32;
  const int len = p.lens ? min(p.lens[b], p.L) : p.L;
  // compact geometry (AttnArgs::uoff): the utterance owns len rows; what lies behind them is the next utterance
  const long rowbase = p.uoff ? (long)p.uoff[b] : (long)p.G + (long)b * p.S;
  const int Lq = p.uoff ? len : p.L;
  if (qt * 32 * NW >= Lq) return;      // (the whole workgroup: nothing of it lies inside the utterance; uniform geometry: never)
  const bool active = q0 < Lq;

  // Q planes: lane (query, half) holds d = 16 s + 8 half + j for k-step s, pre-scaled by log2(e) / 8 * q_scale
  u32x4 q[4][2];
  const float qsc = 0.125f * 1.44269504088896340736f * p.q_scale;
  const float sinv = 1.0f / (p.q_scale * p.k_scale);      // powers of two: exact
  {
    const int qi = q0 + r32;
    const float* src = p.qkv + (rowbase + qi) * p.ld + h * 64 + 8 * half;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0;
      if (qi < Lq) {
        t0 = *reinterpret_cast<const f32x4*>(src + 16 * s);
        t1 = *reinterpret_cast<const f32x4*>(src + 16 * s + 4);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x0 = (e < 2 ? t0[2 * e] : t1[2 * e - 4]) * qsc, x1 = (e < 2 ? t0[2 * e + 1] : t1[2 * e - 3]) * qsc;
        const Split2 t = split2h_pair(x0, x1);
        q[s][0][e] = t.h;
        q[s][1][e] = t.l;
      }
    }
  }

  // ---- this wave's DMA pieces of a key tile: piece pc = wave + NW i -> operand pc >> 3 (K, V), plane (pc >> 2) & 1, 8-key
  // group pc & 3; lane L lands on key 8 g + (L >> 3), slot L & 7 and therefore fetches the slot the read-side key maps there
  const unsigned short* src[PPW];
  int dst[PPW];
  int krel[PPW];
  const int kstride = p.kv_ld;
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int pc = wave + NW * i;
    const int opv = pc >> 3, pl = (pc >> 2) & 1, g = pc & 3;
    const int key = 8 * g + (lane >> 3);
    const int slot = (lane & 7) ^ (opv ? plv_swz(key) : plk_swz(key));
    krel[i] = key;
    src[i] = p.kv2 + (long)pl * p.kv2_plane + rowbase * kstride + opv * 512 + h * 64 + 8 * slot;
    dst[i] = (2 * opv + pl) * PL_PLANE + g * 1024;
  }
  auto issue = [&](int k0, int stage) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      // rows past the last valid key are clamped to it: their scores are masked to -inf below, P is exactly 0 there
      const int key = min(k0 + krel[i], len - 1);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (long)key * kstride),
                                       (__attribute__((address_space(3))) void*)(lds + stage * PL_STAGE + dst[i]), 16, 0, 0);
    }
  };

  int kend = len, kmax = len;
  if (p.chunk > 0) {      // chunk-causal (streaming) mask: jyutvoice/utils/mask.py:91-126
    kend = min(len, ((q0 + r32) / p.chunk + 1) * p.chunk);
    kmax = min(len, ((qt * 32 * NW + 32 * NW - 1) / p.chunk + 1) * p.chunk);
  }
  const int nkt = (kmax + 31) >> 5;

  f32x16 o0, o1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;

  if (nkt > 0) issue(0, 0);
  if (NST > 2 && nkt > 1) issue(32, 1);
  // per-lane read offsets
  const int k_off = r32 * 128;                                   // + pl * PL_PLANE + slot
  const int vq = (lane & 15) >> 2, vp = lane & 3, vg = (lane >> 4) & 1;      // transposed read: block row, 8-byte piece, 16-d group
  for (int kt = 0; kt < nkt; ++kt) {
    const int k0 = kt * 32;
    // NST = 3: tile kt has landed when only tile kt + 1's pieces are outstanding; tile kt + 2 goes into the stage tile kt - 1
    // left.  NST = 2 (32 KB: four workgroups per CU instead of three): nothing else is in flight at the wait, and tile kt + 1
    // is issued behind the barrier into the stage tile kt - 1 left
    if (NST > 2 && kt + 1 < nkt) pl_wait_vmcnt<PPW>(); else pl_wait_vmcnt<0>();
    pl_barrier();
    if (NST > 2) { if (kt + 2 < nkt) issue(k0 + 64, (kt + 2) % NST); }
    else if (kt + 1 < nkt) issue(k0 + 32, (kt + 1) % NST);
    if (!active) continue;
    const unsigned char* const sK = lds + (kt % NST) * PL_STAGE;
    const unsigned char* const sV = sK + 2 * PL_PLANE;
    // S^T[key][query] = sum_d K[key][d] * Q[query][d]
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      u32x4 a[2];
      const int ko = ((2 * st + half) ^ plk_swz(r32)) << 4;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) a[pl] = *reinterpret_cast<const u32x4*>(sK + pl * PL_PLANE + k_off + ko);
      s = pl_mfma3(a, q[st], s);
    }
    if (__builtin_amdgcn_ballot_w64(k0 + 32 > kend) != 0) {      // wave-uniform: only tiles that straddle a mask edge
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = k0 + (e & 3) + 8 * (e >> 2) + 4 * half;
        s[e] = key < kend ? s[e] : -INFINITY;
      }
    }
    float mt = s[0];
#pragma unroll
    for (int e = 1; e < 16; ++e) mt = fmaxf(mt, s[e]);
    mt = pl_half_max(mt) * sinv;
    const float m_new = attn_lazy_max(m_run, mt);      // (= max(m_run, mt) up to the margin: jv_device.h)
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float lt = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      s[e] = __builtin_amdgcn_exp2f(fmaf(s[e], sinv, 10.f - m_new));      // probabilities kept as p * 2^10 (cancels in 1 / l)
      lt += s[e];
    }
    lt = pl_half_sum(lt);
    l_run = l_run * alpha + lt;
    m_run = m_new;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
      for (int e = 0; e < 16; ++e) { o0[e] *= alpha; o1[e] *= alpha; }
    }
    // O^T[d][query] += sum_key V[key][d] * P[query][key]
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      u32x4 pb[2];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const Split2 t = split2h_pair(s[8 * st + 2 * e], s[8 * st + 2 * e + 1]);
        pb[0][e] = t.h;
        pb[1][e] = t.l;
      }
#pragma unroll
      for (int db = 0; db < 2; ++db) {      // d 0..31 -> o0, 32..63 -> o1
        u32x4 a[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
          for (int rd = 0; rd < 2; ++rd) {
            // element j of this lane's fragment is key 16 st + 8 (j >> 2) + 4 half + (j & 3): two 4-key blocks
            const int key = 16 * st + 8 * rd + 4 * half + vq;
            const int byte = (db * 32 + vg * 16 + 4 * vp) * 2;
            const int off = key * 128 + ((((byte >> 4) ^ plv_swz(key)) << 4) | (byte & 15));
            const fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                (__attribute__((address_space(3))) fp16x4*)(const_cast<unsigned char*>(sV) + pl * PL_PLANE + off));
            const u32x2 u = __builtin_bit_cast(u32x2, v);
            a[pl][2 * rd] = u[0];
            a[pl][2 * rd + 1] = u[1];
          }
        }
        if (db == 0) o0 = pl_mfma3(a, pb, o0);
        else o1 = pl_mfma3(a, pb, o1);
      }
    }
  }

  // ---- result: O / l, as the output projection's operand.  Through LDS so that every store instruction writes whole
  // 128-byte rows (from the MFMA layout a lane owns 16-byte pieces of 8 different 2 KB rows per instruction)
  pl_lds_barrier();      // every wave is done with the ring
  const float vsc = p.v_scale;
  const float inv = l_run > 0.f ? (1.0f / vsc) / l_run : 0.f;
  if (p.out2) {
    unsigned char* const so = lds;      // [2 planes][32 NW queries][128 B]
    if (active) {
      const float sc = inv * p.out2_scale;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const f32x16& o = db ? o1 : o0;
          const Split2 a0 = split2h_pair(o[4 * g] * sc, o[4 * g + 1] * sc);
          const Split2 a1 = split2h_pair(o[4 * g + 2] * sc, o[4 * g + 3] * sc);
          // d = 32 db + 8 g + 4 half .. + 3: 8 bytes, half of 16-byte chunk 4 db + g of the row; chunks XOR-keyed by the row
          // (unkeyed the 32 rows of a store meet 16 lanes per bank pair: attention_s.hip, profiles/r03_pmc_bench.md)
          unsigned char* dstp = so + (wave * 32 + r32) * 128 + (((4 * db + g) ^ ((r32 >> 1) & 7)) << 4) + 8 * half;
          *reinterpret_cast<u32x2*>(dstp) = u32x2{a0.h, a1.h};
          *reinterpret_cast<u32x2*>(dstp + 32 * NW * 128) = u32x2{a0.l, a1.l};
        }
      }
    }
    pl_lds_barrier();
    // 8 lanes per row of 128 B; 8 rows per wave-instruction
#pragma unroll
    for (int it = 0; it < (2 * 32 * NW * 8) / (64 * NW); ++it) {
      const int idx = it * 64 * NW + tid;
      const int pl = idx / (32 * NW * 8), rem = idx % (32 * NW * 8);
      const int qrow = rem >> 3, piece = rem & 7;
      const int qi = qt * 32 * NW + qrow;
      if (qi < Lq) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(so + (pl * 32 * NW + qrow) * 128 + ((piece ^ ((qrow >> 1) & 7)) << 4));
        *reinterpret_cast<u32x4*>(p.out2 + (long)pl * p.out2_plane + (rowbase + qi) * p.ldo + h * 64 + piece * 8) = v;
      }
    }
  } else {
    const int qi = q0 + r32;
    if (active && qi < Lq) {
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
}

template <int NW, int NST = PL_NSTAGE>
void launch_pl(const AttnArgs& a, hipStream_t st) {
  hipLaunchKernelGGL((attn64_pl_kernel<NW, NST>), dim3(cdiv(a.L, 32 * NW) * a.H * a.B), dim3(64 * NW), 0, st, a);
}

}  // namespace

// q: fp32 rows (a.qkv, a.ld); K / V: fp16 planes of k * k_scale / v * v_scale at a.kv2 ([2][rows][kv_ld], K at column 0, V at
// column 512, head h at + 64 h).  Same contract as attention64() otherwise.
int attention64_planes(const AttnArgs& a, hipStream_t st) {
  if (a.B <= 0 || a.L <= 0) return JV_OK;
  if (!a.kv2 || (a.kv_ld & 7) || !(a.q_scale > 0.f && a.k_scale > 0.f && a.v_scale > 0.f) || (a.ld & 3) || (a.ldo & 7))
    return fail(JV_ERR_ARG, "attention64_planes: needs K/V planes, the three scales, aligned strides");
  const bool prof = prof_on();
  if (prof) prof_begin(st);
  int nw = a.L <= 64 ? 2 : (round_up(a.L, 256) == round_up(a.L, 128) ? 8 : 4);
  if (const char* f = dyn_env("JV_ATTN_NW")) nw = atoi(f);
  switch (nw) {
    case 2: launch_pl<2>(a, st); break;
    case 4:
      // two stages (32 KB, <= 128 VGPRs): four workgroups per CU instead of three -- the fourth hides more of the others'
      // barriers and softmax chains than the third stage hid of the DMA latency (175.1 -> 171.7 ms per pass, same box)
      if (dyn_env("JV_ATTN_NST3")) launch_pl<4, 3>(a, st);
      else launch_pl<4, 2>(a, st);
      break;
    default: launch_pl<8>(a, st); break;
  }
  if (prof) {
    static const char* const names[9] = {"", "", "attn64_pl<2 waves>", "", "attn64_pl<4 waves>", "", "", "", "attn64_pl<8 waves>"};
    const double bh = (double)a.B * a.H;
    prof_end(st, names[nw], 4.0 * bh * a.L * a.L * 64.0, 4.0 * bh * a.L * 64.0 * 4.0);
  }
  JV_HIP(hipGetLastError());
  return JV_OK;
}

}  // namespace jv
