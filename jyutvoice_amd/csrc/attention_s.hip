// fp16x3 self-attention of the estimator, fourth form: ONE WAVE PER SIMD.  A two-wave workgroup per (utterance, head,
// 64 QT queries), each wave owning 32 QT queries (QT = 5: 160 queries per wave, 320 per workgroup -- a whole 300-frame head;
// 512 workgroups x 2 waves = one wave on every SIMD of the chip, one round) with up to 512 registers, and the three stages
// of a query tile -- S^T = K Q^T, softmax, O^T += V^T P^T -- SOFTWARE-PIPELINED over the wave's query tiles inside every
// key tile: while the vector pipe runs the softmax of tile t, the matrix pipe runs S^T of tile t + 1 and the PV product of
// tile t - 1.  jyutvoice/flow/transformer.py:380-389 -> diffusers AttnProcessor2_0 as restated in oracle/flow.py:46-58.
//
// Why.  attn64_pl (one 32-query tile per wave, four waves per SIMD) leaves the overlap of matrix and vector work to the
// hardware's wave interleaving and gets 27 % of the matrix pipe (profiles/r02_pmc_attention_pl.md); its launch is 1.5 rounds
// of workgroups.  attn64_r (five 16-query tiles per wave, two waves per SIMD) removed the quantisation and four fifths of the
// LDS fragment reads and was slower still: nothing hid its LDS and exp latencies.  A v_mfma_f32_32x32x16_f16 holds the vector
// issue for 8 of its 32 cycles (MI355X_MICROARCH.md): 24 cycles -- six simple vector instructions -- fit under every MFMA
// IF they sit behind it in the same wave's program order and do not depend on it.  This kernel writes the loop that way: the
// MFMAs of two neighbouring tiles and the softmax of the one between them are independent by construction, and
// sched_group_barrier interleaves them one MFMA to four vector instructions.
// Arithmetic: attention_pl.hip's, term for term (32x32x16 MFMAs, the same scales, base-2 softmax on raw v_exp_f32, P kept as
// p * 2^10, products hh' + hl' + lh' smallest first, the same lazy reference maximum, every sum in the same order): the two
// forms agree bit for bit, so an utterance's mel does not depend on which of them its batch size selects.
#include <math.h>
#include <stdlib.h>

#include <type_traits>

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

constexpr int AS_PLANE = 32 * 128;                 // one plane of a 32-key tile: 128 bytes (64 d) per key
constexpr int AS_STAGE = 4 * AS_PLANE;             // K h, K l, V h, V l
constexpr int AS_NW = 2;

// LDS slot keys: attention_pl.hip's (K read by rows with ds_read_b128, V transposed with ds_read_b64_tr_b16)
__device__ __forceinline__ int ask_swz(int key) { return (key >> 1) & 7; }
__device__ __forceinline__ int asv_swz(int key) { return ((key >> 1) & 1) << 2; }

// v_max3_f32 without the canonicalising v_max x, x the IEEE-mode fmaxf puts on operands of unknown origin (asm results)
__device__ __forceinline__ float as_max3(const float a, const float b, const float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float as_half_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float as_half_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ f32x16 as_mfma3(const u32x4 (&a)[2], const u32x4 (&b)[2], f32x16 c) {
  auto mm = [&](const u32x4& x, const u32x4& y) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0);
  };
  mm(a[1], b[0]);      // smallest terms first, as everywhere
  mm(a[0], b[1]);
  mm(a[0], b[0]);
  return c;
}
__device__ __forceinline__ int wave_of(const unsigned tid) { return __builtin_amdgcn_readfirstlane((int)(tid >> 6)); }
template <int N>
__device__ __forceinline__ void as_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void as_barrier() { asm volatile("s_barrier" ::: "memory"); }
__device__ __forceinline__ void as_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- explicit register residency.  A 512-register wave's upper 256 registers are AGPRs: matrix instructions, LDS and
// global loads can address them, vector instructions cannot.  Left to the compiler they are spill space -- first build: 640
// v_accvgpr_read / write per key tile; with "a" constraints: the O tiles rotated through v_accvgpr_mov at every phi; with
// physical-register constraints: the values lived elsewhere and were copied in before each use.  So here the compiler does not
// know about them at all: they are named in the asm text, the file is compiled with -amdgpu-spill-vgpr-to-agpr=0 (build.py),
// the compiler stays below a32 (tools/check_rowgemm_isa.py verifies it on the ISA), and one clobber of a255 makes the
// kernel descriptor allocate all 256.
//   a[0 .. 31]                          left to the compiler (at 256 VGPRs its allocator parks values in the lowest AGPRs)
//   a[32 + 16 N .. + 15]                O tile N = 2 t + d half (N < 10)
//   a[192 + 4 (2 st + pl) .. + 3]       K fragment of k-step st, plane pl           (the A operand of S^T = K Q^T)
//   a[224 + 4 (4 st + 2 db + pl) .. +3] V^T fragment of key step st, d half db      (the A operand of O^T += V^T P^T)
// S, P and BOTH planes' worth of Q that the matrix instructions read (high planes resident, low planes through LDS) are
// ordinary values in VGPRs.  Which operand sits where is not free: the phase stamps (JV_AS_EXPER=3) showed an MFMA with TWO
// VGPR source operands taking ~58 cycles instead of 32 as soon as vector instructions are issued between the MFMAs (they
// compete for the VGPR read ports), and 33 with one -- first assignment (Q high in AGPRs, K / V^T / P in VGPRs): the PV
// product and a third of the S^T product at 58.  Now every MFMA has its A operand in AGPRs and its B operand in VGPRs.
// The order is the source order: every matrix instruction is a volatile asm, sched_barrier(0) after every tick (one MFMA +
// one micro-step of the softmax) -- sched_group_barrier did not interleave anything (second build).
// Software-visible hazards (the hazard recogniser does not see inside inline asm): a vector read of an MFMA result needs the
// MFMA finished -- by construction at least 12 other MFMAs lie between, except after the head of a key tile and before the
// epilogue, where s_nop 15 x 2 stand; v_accvgpr_write -> MFMA source: s_nop 7; LDS reads into AGPRs: a hand-placed
// s_waitcnt lgkmcnt(0) ahead of the first MFMA of the key tile.
constexpr int AS_O = 32, AS_KF = 192, AS_VF = 224;      // first AGPR of the O tiles (five query tiles' worth) / the K / the V^T fragments
template <int N, int F>
__device__ __forceinline__ void as_mfma_o(const u32x4& b) {      // O tile N += V^T fragment F x b
  asm volatile("v_mfma_f32_32x32x16_f16 a[%1:%2], a[%3:%4], %0, a[%1:%2]" ::"v"(b), "n"(AS_O + 16 * N), "n"(AS_O + 16 * N + 15), "n"(AS_VF + 4 * F),
               "n"(AS_VF + 4 * F + 3));
}
template <int F>
__device__ __forceinline__ void as_mfma_s0(f32x16& c, const u32x4& b) {      // c = K fragment F x b
  asm volatile("v_mfma_f32_32x32x16_f16 %0, a[%2:%3], %1, 0" : "=&v"(c) : "v"(b), "n"(AS_KF + 4 * F), "n"(AS_KF + 4 * F + 3));
}
template <int F>
__device__ __forceinline__ void as_mfma_s(f32x16& c, const u32x4& b) {      // c += K fragment F x b
  asm volatile("v_mfma_f32_32x32x16_f16 %0, a[%2:%3], %1, %0" : "+v"(c) : "v"(b), "n"(AS_KF + 4 * F), "n"(AS_KF + 4 * F + 3));
}
template <int R>
__device__ __forceinline__ float as_agpr(void) {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(R));
  return x;
}
// a[R .. R + 15] *= f unless f == 1 in every lane.  The branch is INSIDE the asm: a branch the compiler can see ends the
// basic block, and its sinking passes then carried half of a slot's softmax past it into the next slot's first tick.
#define AS_SC1(e) "v_accvgpr_read_b32 %0, a[%" #e "]\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a[%" #e "], %0\n\t"
template <int R>
__device__ __forceinline__ void as_agpr_scale16(const float f) {
  float t;
  asm volatile("v_cmp_neq_f32 vcc, 1.0, %1\n\ts_cbranch_vccz 1f\n\t"      //
               AS_SC1(2) AS_SC1(3) AS_SC1(4) AS_SC1(5) AS_SC1(6) AS_SC1(7) AS_SC1(8) AS_SC1(9) AS_SC1(10) AS_SC1(11) AS_SC1(12)
                   AS_SC1(13) AS_SC1(14) AS_SC1(15) AS_SC1(16) AS_SC1(17) "s_nop 7\n1:"
               : "=&v"(t)
               : "v"(f), "n"(R), "n"(R + 1), "n"(R + 2), "n"(R + 3), "n"(R + 4), "n"(R + 5), "n"(R + 6), "n"(R + 7), "n"(R + 8),
                 "n"(R + 9), "n"(R + 10), "n"(R + 11), "n"(R + 12), "n"(R + 13), "n"(R + 14), "n"(R + 15)
               : "vcc");
}
#define AS_FENCE() __builtin_amdgcn_sched_barrier(0)

template <int N, class F, int... Is>
__device__ __forceinline__ void as_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void as_for(F&& f) {
  as_for_impl<N>(f, std::make_integer_sequence<int, N>{});
}

constexpr int AS_NOPS = 21;      // micro-steps of one tile's softmax (below), each within the 24 issue cycles an MFMA leaves

#ifdef JV_TUNING
__device__ unsigned long long as_stamps[12];
#endif
// EXPER (tuning builds, JV_AS_EXPER): 1 = no DMA after the first tiles (stale K / V: timing only), 2 = DMA and barriers only
template <int QT, int NST, int EXPER = 0>
__global__ __launch_bounds__(64 * AS_NW, 1) void attn64_s_kernel(const AttnArgs p) {
  asm volatile("" ::: "a255");      // the wave owns all 256 AGPRs (see above)
  unsigned long long t_last = 0, t_acc[12] = {};
  auto stamp = [&](const int slot) {      // (EXPER == 3 only: cycles since the previous stamp, added to as_stamps[slot] at the end;
#ifdef JV_TUNING                          //  no memory operation here -- a global access would put a vmcnt(0) on the K / V DMA)
    if (EXPER == 3) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      if (slot >= 0) t_acc[slot] += t - t_last;
      t_last = t;
    }
#endif
  };
  stamp(-1);
  // the K / V ring; after the key loop each wave's [2 planes][32 queries][128 B] output patch (8 KB per wave)
  // ... and, behind the ring, the LOW planes of the waves' Q operands ([QT][4 k-steps][64 lanes][16 B] per wave, lane-linear;
  // the high planes pass through the same bytes on their way into AGPRs): read back per (tile, k-step) where the MFMA needs it
  __shared__ __attribute__((aligned(256))) unsigned char lds[NST * AS_STAGE + AS_NW * QT * 4 * 1024];
  static_assert(NST * AS_STAGE >= AS_NW * 2 * 32 * 128, "the output patches fit the ring");
  static_assert(NST == 2, "the rotated loop keeps one tile in flight beside the one in registers");
  unsigned char* const qlo = lds + NST * AS_STAGE + wave_of(threadIdx.x) * (QT * 4 * 1024) + (threadIdx.x & 63) * 16;
  constexpr int PPW = 16 / AS_NW;      // DMA pieces per wave and key tile
  constexpr int QW = 32 * QT;          // queries per wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, half = lane >> 5;
  const int nqt = (p.L + AS_NW * QW - 1) / (AS_NW * QW);      // workgroups per head
  const int qt = blockIdx.x % nqt;
  const int h = (blockIdx.x / nqt) % p.H;
  const int b = blockIdx.x / (nqt * p.H);
  const int q0 = qt * AS_NW * QW + wave * QW;
  const int len = p.lens ? min(p.lens[b], p.L) : p.L;
  // compact geometry (AttnArgs::uoff): the utterance owns len rows; what lies behind them is the next utterance
  const long rowbase = p.uoff ? (long)p.uoff[b] : (long)p.G + (long)b * p.S;
  const int Lq = p.uoff ? len : p.L;
  const bool active = q0 < Lq;

  // ---- this wave's DMA pieces of a key tile (attention_pl.hip): piece pc = wave + 2 i -> operand pc >> 3, plane (pc >> 2) & 1,
  // 8-key group pc & 3; lane L lands on key 8 g + (L >> 3), slot L & 7
  const unsigned short* src[PPW];
  int dst[PPW], krel[PPW];
  const int kstride = p.kv_ld;
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int pc = wave + AS_NW * i;
    const int opv = pc >> 3, pl = (pc >> 2) & 1, g = pc & 3;
    const int key = 8 * g + (lane >> 3);
    const int slot = (lane & 7) ^ (opv ? asv_swz(key) : ask_swz(key));
    krel[i] = key;
    src[i] = p.kv2 + (long)pl * p.kv2_plane + rowbase * kstride + opv * 512 + h * 64 + 8 * slot;
    dst[i] = (2 * opv + pl) * AS_PLANE + g * 1024;
  }
  auto issue_piece = [&](const int i, int k0, int stage) {
    const int key = min(k0 + krel[i], len - 1);      // rows past the last valid key: clamped (masked to -inf below)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (long)key * kstride),
                                     (__attribute__((address_space(3))) void*)(lds + stage * AS_STAGE + dst[i]), 16, 0, 0);
  };
  auto issue = [&](int k0, int stage) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) issue_piece(i, k0, stage);
  };
  const int nkt = (len + 31) >> 5;
  if (nkt > 0) issue(0, 0);

  // ---- Q planes: lane (query r32, half) holds d = 16 s + 8 half + j for k-step s, pre-scaled by log2(e) / 8 * q_scale.
  // High planes: staged through the wave's LDS patch and read back INTO AGPRs (ds_read with an "a" destination: the only
  // way to have the tuple defined there); low planes: left in the patch.
  const float qsc = 0.125f * 1.44269504088896340736f * p.q_scale;
  const float sinv = 1.0f / (p.q_scale * p.k_scale);      // powers of two: exact
  // High planes stay in registers (80 for five tiles), low planes go to the wave's LDS patch and come back per (tile, k-step).
  // (the next tile's global loads are issued ahead of this tile's arithmetic)
  u32x4 qh[QT][4];
#ifndef JV_AS_QDEPTH
#define JV_AS_QDEPTH 2      // Q tiles in flight in the prologue (tuning: -DJV_AS_QDEPTH=n).  3 measured the same and 5 SLOWER (54.1 against
                            // 52.8 us per launch, same box): the prologue's 39 MB of Q is a bandwidth burst, not a latency chain
#endif
  constexpr int QD = JV_AS_QDEPTH < QT ? JV_AS_QDEPTH : QT;
  f32x4 qraw[QD][8];
  auto fetch_q = [&](const int t, f32x4 (&r)[8]) {
    const int qi = q0 + 32 * t + r32;
    const float* qs = p.qkv + (rowbase + max(min(qi, Lq - 1), 0)) * p.ld + h * 64 + 8 * half;      // (clamped: zeroed below)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      r[2 * s] = *(const __attribute__((address_space(1))) f32x4*)(qs + 16 * s);
      r[2 * s + 1] = *(const __attribute__((address_space(1))) f32x4*)(qs + 16 * s + 4);
    }
  };
#pragma unroll
  for (int t = 0; t < QD - 1; ++t) fetch_q(t, qraw[t]);
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    if (t + QD - 1 < QT) fetch_q(t + QD - 1, qraw[(t + QD - 1) % QD]);
    const float live = (q0 + 32 * t + r32) < Lq ? qsc : 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const f32x4 t0 = qraw[t % QD][2 * s], t1 = qraw[t % QD][2 * s + 1];
      u32x4 lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x0 = (e < 2 ? t0[2 * e] : t1[2 * e - 4]) * live, x1 = (e < 2 ? t0[2 * e + 1] : t1[2 * e - 3]) * live;
        const Split2 sp = split2h_pair(x0, x1);
        qh[t][s][e] = sp.h;
        lo[e] = sp.l;
      }
      *reinterpret_cast<u32x4*>(qlo + (t * 4 + s) * 1024) = lo;      // (read back by this lane only: no barrier needed)
    }
  }

  float m_run[QT], l_run[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    m_run[t] = -INFINITY;
    l_run[t] = 0.f;
  }
  // O^T tile 2 t + db: d 32 db .. 32 db + 31 x query r32 of query tile t
  as_for<32 * QT>([&](auto rc) { asm volatile("v_accvgpr_write_b32 a[%0], 0" ::"n"(AS_O + decltype(rc)::value)); });

  const int k_off = r32 * 128;
  const int vq = (lane & 15) >> 2, vp = lane & 3, vg = (lane >> 4) & 1;
  // ---- the key tile's K and V^T fragments live in registers, ONCE for all the wave's query tiles (they do not depend on the
  // query tile: a fifth of attn64_pl's LDS fragment traffic, and nothing but the low Q planes is read inside the slots).
  // The loop is rotated so that the NEXT tile's fragments are read under the tail of this one: K after the last S^T product
  // (before the tail's PV MFMAs), V^T after them (before the next head's S^T MFMAs).
  // (lane-dependent LDS addresses once; stage, plane and key step are immediates / one add)
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  unsigned ka[4], va[2];
#pragma unroll
  for (int st = 0; st < 4; ++st) ka[st] = lds0 + k_off + (((2 * st + half) ^ ask_swz(r32)) << 4);
#pragma unroll
  for (int db = 0; db < 2; ++db) {
    const int key = 4 * half + vq, byte = (db * 32 + vg * 16 + 4 * vp) * 2;      // (asv_swz depends on vq only)
    va[db] = lds0 + 2 * AS_PLANE + key * 128 + ((((byte >> 4) ^ asv_swz(key)) << 4) | (byte & 15));
  }
  auto read_kf = [&](const int stage) {
    as_for<8>([&](auto fc) {
      constexpr int F = decltype(fc)::value, st = F >> 1, pl = F & 1;
      (void)ka, (void)stage;
      const unsigned addr = ka[st] + stage * AS_STAGE;
      asm volatile("ds_read_b128 a[%1:%2], %0 offset:%3" ::"v"(addr), "n"(AS_KF + 4 * F), "n"(AS_KF + 4 * F + 3),
                   "n"(pl * AS_PLANE)
                   : "memory");
    });
    AS_FENCE();
  };
  auto read_vf = [&](const int stage) {
    as_for<16>([&](auto fc) {
      constexpr int X = decltype(fc)::value, F = X >> 1, rd = X & 1, st = F >> 2, db = (F >> 1) & 1, pl = F & 1;
      (void)va, (void)stage;
      const unsigned addr = va[db] + stage * AS_STAGE;
      asm volatile("ds_read_b64_tr_b16 a[%1:%2], %0 offset:%3" ::"v"(addr), "n"(AS_VF + 4 * F + 2 * rd),
                   "n"(AS_VF + 4 * F + 2 * rd + 1), "n"(pl * AS_PLANE + (16 * st + 8 * rd) * 128)
                   : "memory");
    });
    AS_FENCE();
  };
  if (nkt > 0) {      // tile 0 has landed: the second tile's DMA goes out, the first one's fragments come in
    as_wait_vmcnt<0>();
    as_barrier();
    if (EXPER != 1 && nkt > 1) issue(32, 1);
    if (active) {
      read_kf(0);
      read_vf(0);
    }
  }
  // the low Q planes of the tile whose S^T is formed in a slot are read from the wave's LDS patch a slot AHEAD (two buffers
  // by tile parity): read at the top of their own slot, the second MFMA of the slot waited ~100 cycles for them
  u32x4 ql[2][4];
  auto load_ql = [&](const int t) {
#pragma unroll
    for (int st = 0; st < 4; ++st) ql[t & 1][st] = *reinterpret_cast<const u32x4*>(qlo + (t * 4 + st) * 1024);
  };
  // ---- result of query tile T: O / l as the output projection's operand (fp16 planes; fp32 rows without out2), transposed
  // through an 8 KB LDS patch [2 planes][32 queries][128 B] so that the stores are whole 128-byte rows.  Called from inside the
  // LAST key tile as soon as a tile's PV product is complete (after slot T + 1), through the ring stage no DMA will write
  // again: the 40 MB of output leave under the remaining tiles' arithmetic instead of in one burst after it.
  const float vsc = p.v_scale;
  auto write_tile = [&](auto tc, unsigned char* const so) __attribute__((always_inline)) {
    constexpr int T = decltype(tc)::value;
    (void)vsc, (void)l_run, (void)rowbase, (void)h, (void)Lq;
    // (everything below is invariant over the key loop this is called from: made opaque, or the compiler hoists ~70 address
    //  and predicate registers out of the loop and parks them in AGPRs -- the ones this kernel owns)
    int q0l = q0, lane = tid & 63;
    asm volatile("" : "+v"(q0l), "+v"(lane));
    const int r32 = lane & 31, half = lane >> 5;
    const int qbase = q0l + 32 * T;
    if (qbase >= Lq) return;      // (wave-uniform)
    asm volatile("s_nop 15\n\ts_nop 15");      // the tile's last PV MFMA may have been issued one instruction ago
    const float l = l_run[T];
    const float inv = l > 0.f ? (1.0f / vsc) / l : 0.f;
    if (p.out2) {
      const float sc = inv * p.out2_scale;
      as_for<8>([&](auto gc) {      // four values at a time: the call sites sit inside the key loop, where registers are scarce
        constexpr int db = decltype(gc)::value >> 2, g = decltype(gc)::value & 3, R = AS_O + 32 * T + 16 * db + 4 * g;
        const Split2 a0 = split2h_pair(as_agpr<R>() * sc, as_agpr<R + 1>() * sc);
        const Split2 a1 = split2h_pair(as_agpr<R + 2>() * sc, as_agpr<R + 3>() * sc);
        // d = 32 db + 8 g + 4 half .. + 3: 8 bytes, the half of 16-byte chunk 4 db + g of the query's row; chunks XOR-keyed by
        // the row (ask_swz) -- unkeyed, the 32 rows of a store put 16 lanes on each bank pair (PMC: 4.9 M conflict cycles of
        // 7.0 M LDS-active per launch), keyed two
        unsigned char* dp = so + r32 * 128 + (((4 * db + g) ^ ask_swz(r32)) << 4) + 8 * half;
        *reinterpret_cast<u32x2*>(dp) = u32x2{a0.h, a1.h};
        *reinterpret_cast<u32x2*>(dp + 32 * 128) = u32x2{a0.l, a1.l};
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (a wave's LDS accesses execute in order)
      // 8 lanes per row of 128 B, 8 rows per instruction: 2 planes x 32 rows = 8 instructions
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int pl = it >> 2, qrow = (it & 3) * 8 + (lane >> 3), piece = lane & 7;
        const int qi = qbase + qrow;
        const u32x4 v = *reinterpret_cast<const u32x4*>(so + (pl * 32 + qrow) * 128 + ((piece ^ ask_swz(qrow)) << 4));
        if (qi < Lq) *(__attribute__((address_space(1))) u32x4*)(p.out2 + (long)pl * p.out2_plane + (rowbase + qi) * p.ldo + h * 64 + piece * 8) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the patch is rewritten by the next tile
    } else {
      const int qi = qbase + r32;
      as_for<8>([&](auto gc) {
        constexpr int db = decltype(gc)::value >> 2, g = decltype(gc)::value & 3, R = AS_O + 32 * T + 16 * db + 4 * g;
        const f32x4 a = {as_agpr<R>() * inv, as_agpr<R + 1>() * inv, as_agpr<R + 2>() * inv, as_agpr<R + 3>() * inv};
        if (qi < Lq) *(__attribute__((address_space(1))) f32x4*)(p.out + (rowbase + qi) * p.ldo + h * 64 + 4 * half + 32 * db + 8 * g) = a;
      });
    }
  };
  const bool work = active && EXPER != 2;
  if (work) load_ql(0);
  for (int kt = 0; kt < nkt; ++kt) {
    const int k0 = kt * 32;

    // One key tile.  EDGE (the last tile of the head: the only one that can straddle the key mask, the only one without a
    // successor) is a compile-time copy of the body.
    auto key_tile = [&](auto edge_c) {
      constexpr bool EDGE = decltype(edge_c)::value;
      f32x16 s[2];            // S^T of the tile in softmax / of the next one, by tile parity
      u32x4 pb[2][2][2];      // P planes [parity][16-key step][plane]
      struct { float m0, m1, m2, m3, mt, dm, alpha, bias, lt; } w;

      // MFMA I of S^T[key][query] = sum_d K[key][d] Q[query][d] for tile t: k-step I / 3, term I % 3 (smallest first)
      auto qk_mfma = [&](auto ic, auto tc) {
        constexpr int I = decltype(ic)::value, t = decltype(tc)::value, st = I / 3, term = I % 3;
        (void)ql, (void)qh;      // (named outside the discarded branches: implicit capture in a generic lambda)
        f32x16& c = s[t & 1];
        if constexpr (I == 0) as_mfma_s0<1>(c, qh[t][0]);                 // K low x Q high
        else if constexpr (term == 0) as_mfma_s<2 * st + 1>(c, qh[t][st]);
        else if constexpr (term == 1) as_mfma_s<2 * st>(c, ql[t & 1][st]);       // K high x Q low
        else as_mfma_s<2 * st>(c, qh[t][st]);                             // K high x Q high
      };
      // MFMA I of O^T[d][query] += sum_key V[key][d] P[query][key] for tile t: 16-key step I / 6, d half (I / 3) & 1, term I % 3
      auto pv_mfma = [&](auto ic, auto tc) {
        constexpr int I = decltype(ic)::value, t = decltype(tc)::value, st = I / 6, db = (I / 3) & 1, term = I % 3;
        constexpr int F = 4 * st + 2 * db;
        const u32x4(&P)[2] = pb[t & 1][st];
        if constexpr (term == 0) as_mfma_o<2 * t + db, F + 1>(P[0]);      // V low x P high
        else if constexpr (term == 1) as_mfma_o<2 * t + db, F>(P[1]);     // V high x P low
        else as_mfma_o<2 * t + db, F>(P[0]);                              // V high x P high
      };
      // micro-step I of tile t's softmax: s -> P planes (B operand of PV) and the running statistics.
      // The reference maximum is lazy (attn_lazy_max, jv_device.h) and every sum is formed in attention_pl.hip's order (keys
      // of the lane in sequence, the two lane halves per tile, l = l alpha + that): the two forms agree bit for bit.
      auto smx = [&](auto ic, const int t) {
        constexpr int I = decltype(ic)::value;
        (void)w, (void)pb, (void)m_run, (void)l_run, (void)sinv, (void)len, (void)k0, (void)half;
        f32x16& x = s[t & 1];
        auto ex = [&](const int e) { x[e] = __builtin_amdgcn_exp2f(fmaf(x[e], sinv, w.bias)); };      // 12 issue cycles
        auto add2 = [&](const int j) { w.lt = j ? (w.lt + x[2 * j]) + x[2 * j + 1] : (0.f + x[0]) + x[1]; };  // 8
        auto split = [&](const int j) {                                                                // 16-18
          const int st = j >> 2, e = j & 3;
          const Split2 sp = split2h_pair(x[8 * st + 2 * e], x[8 * st + 2 * e + 1]);
          pb[t & 1][st][0][e] = sp.h;
          pb[t & 1][st][1][e] = sp.l;
        };
        if constexpr (I == 0) {
          if (EDGE) {
#pragma unroll
            for (int e = 0; e < 16; ++e) x[e] = (k0 + (e & 3) + 8 * (e >> 2) + 4 * half) < len ? x[e] : -INFINITY;
          }
          w.m0 = as_max3(x[0], x[1], x[2]);
          w.m1 = as_max3(x[3], x[4], x[5]);
          w.m2 = as_max3(x[6], x[7], x[8]);
          w.m3 = as_max3(x[9], x[10], x[11]);
        } else if constexpr (I == 1) {
          const float m4 = as_max3(x[12], x[13], x[14]);
          w.m0 = as_max3(w.m0, w.m1, w.m2);
          w.m3 = as_max3(w.m3, m4, x[15]);
          w.mt = as_max3(w.m0, w.m3, w.m3);
        } else if constexpr (I == 2) {
          const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(w.mt), __float_as_uint(w.mt), false, false);
          w.mt = as_max3(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[1])) * sinv;
        } else if constexpr (I == 3) {
          const float m_new = attn_lazy_max(m_run[t], w.mt);      // (m_run = -inf at the first tile: always moves)
          w.dm = m_run[t] - m_new;
          w.bias = 10.f - m_new;      // p * 2^10 (cancels in 1 / l)
          m_run[t] = m_new;
        } else if constexpr (I == 4) {
          w.alpha = __builtin_amdgcn_exp2f(w.dm);
          ex(0);
        } else if constexpr (I <= 11) {
          ex(2 * (I - 5) + 1);
          ex(2 * (I - 5) + 2);
        } else if constexpr (I == 12) {
          ex(15);
          add2(0);
        } else if constexpr (I <= 19) {
          split(I - 13);
          add2(I - 12);
        } else {
          split(7);
          l_run[t] = l_run[t] * w.alpha + as_half_sum(w.lt);
        }
      };

      if (work) {
        // ---- head: S^T of tile 0 (beside it: the V^T fragment reads issued under the previous tile's tail)
        if (QT > 1) load_ql(1);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(QT > 1 ? 4 : 0) : "memory");      // this tile's K / V^T fragments (read into AGPRs
                                                                                   // under the last tail) and tile 0's low planes
        AS_FENCE();
        as_for<12>([&](auto ic) { qk_mfma(ic, std::integral_constant<int, 0>{}); });
        asm volatile("s_nop 15\n\ts_nop 15");      // the last MFMA's result is read by the next vector instruction
        AS_FENCE();
        stamp(1);
        // ---- slots: slot T runs S^T of tile T + 1 and the PV product of tile T - 1 on the matrix pipe beside the softmax
        // of tile T on the vector pipe -- tick I = MFMA I, then micro-step I of the softmax (a 32-cycle MFMA holds the
        // vector issue for 8 cycles: 24 cycles of vector instructions fit behind it, a micro-step is sized to that)
        as_for<QT>([&](auto tc) {
          constexpr int T = decltype(tc)::value;
          constexpr bool HQ = T + 1 < QT, HP = T > 0;
          constexpr int NQ = HQ ? 12 : 0, NM = NQ + (HP ? 12 : 0);
          if constexpr (T + 2 < QT) load_ql(T + 2);
          else if constexpr (T + 1 == QT && !EDGE) load_ql(0);      // (the next key tile's first: the same bytes again)
          as_for<(NM > AS_NOPS ? NM : AS_NOPS)>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            if constexpr (I < NQ) qk_mfma(ic, std::integral_constant<int, T + 1>{});
            else if constexpr (I < NM) pv_mfma(std::integral_constant<int, I - NQ>{}, std::integral_constant<int, T - 1>{});
            if constexpr (I < AS_NOPS) smx(ic, T);
            AS_FENCE();
          });
          as_agpr_scale16<AS_O + 32 * T>(w.alpha);      // (between slots: O[T] is next touched by PV(T), 12 MFMAs on)
          as_agpr_scale16<AS_O + 32 * T + 16>(w.alpha);
          AS_FENCE();
          if constexpr (EDGE && T > 0) {      // tile T - 1 is complete: out it goes
            write_tile(std::integral_constant<int, T - 1>{}, lds + ((kt + 1) % NST) * AS_STAGE + wave * (2 * 32 * 128));
            AS_FENCE();
          }
          stamp(2 + T);
        });
      }
      if constexpr (!EDGE) {      // the next tile has landed (issued a tile ago); this tile's stage is free (its fragments are
        as_wait_vmcnt<0>();       // in registers, on both waves once they meet here): the tile after next goes there
        as_barrier();
        if (EXPER != 1 && kt + 2 < nkt) issue(k0 + 64, kt % NST);      // (issued piecewise in the gaps of the tail's MFMAs it
        if (active) read_kf((kt + 1) % NST);                            //  measured slower: 55.2 vs 53.0 us per launch)
      }
      stamp(7);
      if (work) {      // ---- tail: the PV product of the last tile
        as_for<12>([&](auto ic) { pv_mfma(ic, std::integral_constant<int, QT - 1>{}); });
        AS_FENCE();
        if constexpr (EDGE) write_tile(std::integral_constant<int, QT - 1>{}, lds + ((kt + 1) % NST) * AS_STAGE + wave * (2 * 32 * 128));
      }
      stamp(8);
      if constexpr (!EDGE) {
        if (active) read_vf((kt + 1) % NST);
      }
      stamp(9);
#ifdef JV_TUNING
      if (EXPER == 3) t_acc[11] += 1;
#endif
    };
    if (kt + 1 < nkt) key_tile(std::false_type{});
    else key_tile(std::true_type{});
  }
  if (nkt == 0 && active) {      // an utterance without keys: zeros
    for (int r = lane >> 3; r < QW; r += 8) {
      const int qi = q0 + r, piece = lane & 7;
      if (qi >= Lq) break;
      if (p.out2) {
        for (int pl = 0; pl < 2; ++pl)
          *(__attribute__((address_space(1))) u32x4*)(p.out2 + (long)pl * p.out2_plane + (rowbase + qi) * p.ldo + h * 64 + piece * 8) = u32x4{0u, 0u, 0u, 0u};
      } else {
        for (int c = 0; c < 2; ++c)
          *(__attribute__((address_space(1))) f32x4*)(p.out + (rowbase + qi) * p.ldo + h * 64 + piece * 8 + 4 * c) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  stamp(10);
#ifdef JV_TUNING
  if (EXPER == 3 && blockIdx.x == 0 && threadIdx.x == 0)
    for (int i = 0; i < 12; ++i) as_stamps[i] = t_acc[i];
#endif
}

template <int QT, int NST>
void launch_s(const AttnArgs& a, hipStream_t st) {
  const int nqt = cdiv(a.L, AS_NW * 32 * QT);
#ifdef JV_TUNING
  if (const char* e = dyn_env("JV_AS_EXPER")) {
    if (atoi(e) == 1) hipLaunchKernelGGL((attn64_s_kernel<QT, NST, 1>), dim3(nqt * a.H * a.B), dim3(64 * AS_NW), 0, st, a);
    else if (atoi(e) == 3) {      // phase stamps of workgroup 0, wave 0 (cycles, summed over the key tiles)
      static const char* const nm[12] = {"prologue", "head", "slot0", "slot1", "slot2", "slot3", "slot4", "switch", "tail", "vf", "epilogue", "tiles"};
      unsigned long long z[12] = {}, h[12];
      hipMemcpyToSymbol(HIP_SYMBOL(as_stamps), z, sizeof z);
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipEventRecord(e0, st);
      hipLaunchKernelGGL((attn64_s_kernel<QT, NST, 3>), dim3(nqt * a.H * a.B), dim3(64 * AS_NW), 0, st, a);
      hipEventRecord(e1, st);
      hipStreamSynchronize(st);
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      hipMemcpyFromSymbol(h, HIP_SYMBOL(as_stamps), sizeof h);
      static int once = 0;
      if (once++ == 3) {
        unsigned long long tot = 0;
        for (int i = 0; i < 11; ++i) tot += h[i];
        for (int i = 0; i < 12; ++i) fprintf(stderr, "as_stamp %-9s %8llu\n", nm[i], h[i]);
        fprintf(stderr, "as_stamp total %llu ticks of s_memtime in %.1f us of the launch (events)\n", tot, ms * 1e3);
      }
    }
    else hipLaunchKernelGGL((attn64_s_kernel<QT, NST, 2>), dim3(nqt * a.H * a.B), dim3(64 * AS_NW), 0, st, a);
    return;
  }
#endif
  hipLaunchKernelGGL((attn64_s_kernel<QT, NST>), dim3(nqt * a.H * a.B), dim3(64 * AS_NW), 0, st, a);
}

}  // namespace

static int single_qt(const AttnArgs& a) {
  // queries per workgroup = 64 QT: the QT in {2 .. 5} with the fewest padded queries, the taller tile on a tie
  int qt = 5;
  long best = 1L << 40;
  for (int c = 5; c >= 2; --c) {
    const long padded = (long)cdiv(a.L, 64 * c) * 64 * c;
    if (padded < best) { qt = c; best = padded; }
  }
  return qt;
}

// Is this the form to use?  One wave per SIMD means 512 workgroups are one round of the chip and 513 are two: the form pays
// when its workgroups fill their rounds (32 or 64 utterances of <= 320 frames x 8 heads: 512 / 1024); at 40 utterances
// (640 workgroups: 1.25 rounds run as 2) attn64_pl's small workgroups lose less.  tests/test_gpu_regimes.py holds the seam.
bool attention64_single_fits(const AttnArgs& a) {
  if (a.B <= 0 || a.L <= 0 || a.chunk != 0) return false;
  // ragged batch (compact geometry): one workgroup per head is as long as its utterance's keys and queries, and with one round
  // of workgroups the launch is as long as the longest -- attn64_pl's 128-query workgroups even the lengths out
  if (a.uoff) return false;
  const long wgs = (long)cdiv(a.L, AS_NW * 32 * single_qt(a)) * a.H * a.B;
  const long slots = (long)cdiv(wgs, 512) * 512;
  return wgs * 100 >= slots * 85;
}

// same contract as attention64_planes() (attention_pl.hip); no chunk-causal mask
int attention64_single(const AttnArgs& a, hipStream_t st) {
  if (a.B <= 0 || a.L <= 0) return JV_OK;
  if (!a.kv2 || (a.kv_ld & 7) || !(a.q_scale > 0.f && a.k_scale > 0.f && a.v_scale > 0.f) || (a.ld & 3) || (a.ldo & 7) || a.chunk > 0)
    return fail(JV_ERR_ARG, "attention64_single: needs K/V planes, the three scales, aligned strides, no chunk mask");
  int qt = single_qt(a);
  if (const char* f = dyn_env("JV_ATTN_QT")) qt = atoi(f);
  if (qt < 2 || qt > 5) return fail(JV_ERR_ARG, "attention64_single: bad tile count");      // before the profiler's start event
  const bool prof = prof_on();
  if (prof) prof_begin(st);
  switch (qt) {
    case 2: launch_s<2, 2>(a, st); break;
    case 3: launch_s<3, 2>(a, st); break;
    case 4: launch_s<4, 2>(a, st); break;
    case 5: launch_s<5, 2>(a, st); break;
    default: return fail(JV_ERR_ARG, "attention64_single: bad tile count");
  }
  if (prof) {
    static const char* const names[6] = {"", "", "attn64_s<128 q>", "attn64_s<192 q>", "attn64_s<256 q>", "attn64_s<320 q>"};
    const double bh = (double)a.B * a.H;
    prof_end(st, names[qt], 4.0 * bh * a.L * a.L * 64.0, 4.0 * bh * a.L * 64.0 * 4.0);
  }
  JV_HIP(hipGetLastError());
  return JV_OK;
}

}  // namespace jv
