// One launch for a ResBlock's convolution PAIR in the vocoder (jyutvoice/hifigan/generator.py:90-97):
//   xt = Conv1d(C, C, k, dilation d)(Snake1(x));  xt = Conv1d(C, C, k)(Snake2(xt));  x = xt + x
// at C = 64 / 128 channels (the 120 T + 1 and 40 T rows per utterance: 1.15 M / 384 K rows at the benchmarked size).
//
// Why.  As two hiftconv_kernel launches the pair moves five tensors -- x in, xt out, xt in, x in again as the residual, the
// result out -- for 2 k C^2 multiply-adds per element: at C = 64 that is 0.86 GB per launch for 28 GFLOP x k, and the launch
// takes 360 us where its MFMAs need 42 / 97 / 153 us (k = 3 / 7 / 11, fp16x3 at the clock the chip holds) and its bytes ~150:
// load, Snake, MFMAs and stores of a workgroup run one after the other and two co-resident workgroups overlap them only in
// part.  Here the intermediate never leaves LDS: three tensor passes instead of five, one prologue and one epilogue per pair.
//
// A workgroup (hiftconv_kernel's shape: NG C / 32 waves, each a 80 x 32 tile of both products) owns R = 80 NG rows of the
// INTERMEDIATE, i.e. RO = R - (k - 1) output rows (the second convolution's halo is recomputed by the neighbours: 2 - 7 % of
// the first product):
//   1. the window of x -- R + (k - 1) d rows -- through Snake1 and the plane split into LDS, once (hiftconv_kernel's pass);
//   2. first product: intermediate rows m0 - h2 .. m0 - h2 + R - 1 (h2 = (k - 1) / 2), accumulators in registers;
//   3. the accumulators -> + bias -> mask (rows outside the utterance's frames read as ZERO in the second convolution: its
//      "same" padding) -> Snake2 -> x scale2 -> planes, written over the window's image in the operand layout (rowblock_kernel's
//      GELU pass: neighbouring lanes trade one row of each pair and store whole dwords);
//   4. second product (dilation 1) over that image: output rows m0 .. m0 + RO - 1 are tile rows 0 .. RO - 1;
//   5. per-wave epilogue: + bias + x (re-read: 41 KB per workgroup, L2-warm) (+ a second residual), scaling, accumulation,
//      measured-bound tracking of what is stored -- hiftconv_kernel's.
// The weights of both convolutions arrive as ONE fragment stream (k-step major fragments: the second's steps follow the first's,
// registry.hip) through the register double buffer with hand-counted waits (rowgemm_wd_kernel's rules; tools/check_rowgemm_isa.py).
//
// The intermediate's fp16x3 scale.  Unfused, the first launch MEASURES max |xt| per utterance and the second derives its power
// of two from it.  Here no such maximum exists before the intermediate is consumed, so the scale comes from a BOUND instead:
//   |xt| <= L1 (amax_x + e1) + max |b1|,   L1 = the largest row L1 norm of the first convolution's weights (load time),
// amax_x the measured bound of x of the row's utterance, e1 / e2 what Snake1 / Snake2 can add (max 1 / alpha).  Overflow is
// impossible by construction as before; the bound is looser than a measurement (typically 5 - 20 x), which costs nothing in
// relative precision (22 bits per element down to 2^-17 of the bound) and moves the absolute floor (2^-40 of the bound) by
// as much.  A function of the utterance alone: results do not depend on the batch or on where a row sits.  Fused and unfused
// agree to rounding (tests/test_gpu_ops.py, tests/test_gpu_pipeline.py).
#pragma once
#include "hiftconv_kernel.h"

namespace jv {

struct HiftPairArgs {
  const float* A;                  // the block's input x, fp32 rows [rows, C]; also the residual
  long a_rows;                     // rows of A that may be read (others read as zero)
  int M;                           // output rows
  int ntaps, dil;                  // k (both convolutions), the first one's dilation; "same" padding on both
  const unsigned char* rowmask;    // per row or null: 0 -> the row reads as zero, as x and as the intermediate; not tracked
  const float *alpha1, *alpha2;    // Snake parameters per channel
  const unsigned short* Wf;        // fragment order over K = 2 k C: the first convolution's k C / 32 steps, then the second's
  long wf_plane;
  const float *cs1, *b1, *cs2, *b2;      // column scales (2^-e_n) and biases of the two convolutions
  const float* amax_in;            // per-utterance measured bound of x
  float e1, e2;                    // max 1 / (alpha + 1e-9) of Snake1 / Snake2
  float l1max, b1max;              // largest row L1 norm and largest |bias| of the first convolution
  int slot_G, slot_S, slot_nb;     // slot(row) = clamp((row - slot_G) / slot_S, 0, slot_nb - 1)
  const int* slot_map;             // or, when set: slot(row) = slot_map[row] (HiftConvArgs::slot_map)
  float* out;                      // [rows, C]: out = ((acc2 + b2) + x + res2) * out_scale (+ previous out)
  const float* res2;               // [rows, C] or null
  float out_scale;
  int accumulate;
  float* amax_out;                 // tracking of what is stored, per utterance slot (rows with rowmask == 0 excluded)
  long alg_rows;
};

template <int NG> constexpr int hp_rows() { return HC_RG * NG; }
template <int C, int NG> inline int hp_lds_bytes(int ntaps, int dil) {
  const int R = hp_rows<NG>();
  const int w1 = hc_wrpad(R, ntaps, dil), w2 = hc_wrpad(R, ntaps, 1);
  const int wr = w1 > w2 ? w1 : w2;
  return ((w1 * 4 + 255) & ~255) + R * 16 + (C / 32) * 2 * wr * 64 + NG * (C / 32) * 16 * 36 * 4;
}

template <int N, class F, int... Is>
__device__ __forceinline__ void hp_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void hp_for(F&& f) {
  hp_for_impl<N>(f, std::make_integer_sequence<int, N>{});
}

// KT: the kernel size, a template parameter -- the step sequence of both products is then straight-line code, every step's
// weight buffer (step mod NB) and A-fragment parity (step mod 2) a compile-time constant.  (Chosen at run time through a
// switch per step, the six (buffer, parity) combinations made the register allocator copy the buffers at every join: 567
// spilled registers, and copies of registers with loads in flight are exactly what tools/check_rowgemm_isa.py forbids.)
template <int C, int NG, int KT>
__global__ __launch_bounds__(64 * NG * (C / 32), 2) void hiftpair_kernel(const HiftPairArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char hc_lds[];
  constexpr int RT = HC_RT, RG = HC_RG;
  constexpr int NCH = C / 32, CW = C / 32, R = RG * NG, NT = 64 * NG * CW;
  constexpr int C4 = C / 4;                     // float4 per row
  constexpr int RPI = NT / C4;                  // window rows staged per pass of the workgroup
  static_assert(R + 56 <= NT, "one window row per thread in the facts pass");
  constexpr int NWL = 4;
  typedef const __attribute__((address_space(1))) unsigned char* gbytes;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
  const int grp = wave / CW, cp = wave % CW;    // row group, 32-column pair
  constexpr int k = KT;
  const int h2 = (k - 1) >> 1, h1 = (p.dil * (k - 1)) >> 1;
  const int RO = R - (k - 1);                   // output rows of this workgroup
  const int m0 = blockIdx.x * RO;
  const int WR = R + (k - 1) * p.dil;           // window rows of x really used
  const int WRP = ((WR + 7) & ~7) + 1;
  const int WRP2 = ((R + (k - 1) + 7) & ~7) + 1;      // rows of the intermediate's image (its last k - 1 are never written: they
                                                      // reach discarded output rows only)
  const int PS = WRP * 64, CS = 2 * PS;         // plane and chunk strides of the window's image
  const int PS2 = WRP2 * 64, CS2 = 2 * PS2;     // ... and of the intermediate's, laid over it
  constexpr int KS1 = k * NCH;                  // 32-deep steps per convolution (even: NCH is)
  constexpr int KS = 2 * KS1;
  float* const wscale = reinterpret_cast<float*>(hc_lds);                                   // [WRP]: < 0 = the row reads as zero
  float* const yinv1 = reinterpret_cast<float*>(hc_lds + ((WRP * 4 + 255) & ~255));         // [R]: 1 / scale1 of the intermediate row's utterance
  float* const ysc2 = yinv1 + R;                                                            // [R]: scale2 of it; < 0 = the row reads as zero
  int2* const rowinfo = reinterpret_cast<int2*>(ysc2 + R);                                  // [R]: output rows (x = bits of 1 / scale2)
  unsigned char* const img = reinterpret_cast<unsigned char*>(rowinfo) + R * 8;            // [NCH][2][max(WRP, WRP2)][64 B]
  const int wrm = WRP > WRP2 ? WRP : WRP2;
  float* const patch = reinterpret_cast<float*>(img + NCH * 2 * wrm * 64) + wave * (16 * 36);

  // ---- W: fragment order, [plane][KS][C / 16 blocks][64 lanes][8 halves] (k-step major); this wave's column blocks are 2 cp + nt ----
  const unsigned short* wp[2][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) wp[nt][pl] = p.Wf + (long)pl * p.wf_plane + (long)(2 * cp + nt) * 512 + lane * 8;
  int wk = 0;
  // NB weight buffers: a step's fragments are requested NB steps ahead.  Two (the trunk kernels' double buffer) is 1.5 steps
  // of lookahead at the first wait -- at ONE wave per SIMD and workgroup, as here, less than an L2 round trip (rowblock's
  // staggered schedule measured it: 930 cycles per step for a lone wave against 480 of MFMA issue); this kernel has the 16
  // registers a third buffer costs.
#ifndef JV_HP_WBUF
#define JV_HP_WBUF 3
#endif
  constexpr int NB = JV_HP_WBUF;
  static_assert(NB == 2 || NB == 3, "two or three weight buffers");
  rg_u32x4 bq[NB][2][2];
#pragma unroll
  for (int i = 0; i < 4 * NB; ++i) bq[i >> 2][(i >> 1) & 1][i & 1] = rg_u32x4{0u, 0u, 0u, 0u};
  auto load_frag = [](rg_u32x4& dst, const unsigned short* ptr) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(dst) : "v"(ptr) : "memory");
  };
  auto load_w = [&](auto par_tag, auto nttag) {
    constexpr int par = decltype(par_tag)::value, nt = decltype(nttag)::value;
    load_frag(bq[par][nt][0], wp[nt][0]);
    load_frag(bq[par][nt][1], wp[nt][1]);
  };
  auto landed_w = [](rg_u32x4& b00, rg_u32x4& b01, rg_u32x4& b10, rg_u32x4& b11) {
    asm volatile("" : "+v"(b00), "+v"(b01), "+v"(b10), "+v"(b11)::"memory");
  };
  auto advance_w = [&]() {      // + C / 16 blocks x 1 KB (512 halves) per step; past the end: back to the first step (weights that exist)
    constexpr long WSTEP = 512L * (C / 16);
    const long d = ++wk == KS ? WSTEP - WSTEP * KS : WSTEP;
    if (wk == KS) wk = 0;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) wp[nt][pl] += d;
  };
  // the weights start first: everything below is global-load latency they spend in flight
  load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
  advance_w();
  load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
  load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
  advance_w();
  if constexpr (NB == 3) {
    load_w(std::integral_constant<int, NB - 1>{}, std::integral_constant<int, 0>{});
    load_w(std::integral_constant<int, NB - 1>{}, std::integral_constant<int, 1>{});
    advance_w();
  }

  // ---- per-row facts (one level of unconditional loads on clamped indices: rowconv_wd_kernel) ----
  auto slot_of = [&](const long row) -> int {
    if (p.slot_map) {      // compact geometry of ragged batches (hift.hip): by table; every caller clamps `row` to the buffer
      const int q = p.slot_map[row];
      return q < 0 ? 0 : (q >= p.slot_nb ? p.slot_nb - 1 : q);
    }
    if (p.slot_S <= 0) return 0;
    const int q = (int)((row - p.slot_G) / p.slot_S);
    return q < 0 ? 0 : (q >= p.slot_nb ? p.slot_nb - 1 : q);
  };
  constexpr int RI_TRACK = 1 << 30;
  {
    // window row tid <-> global row m0 - h2 - h1 + tid; intermediate row tid <-> m0 - h2 + tid; output row tid <-> m0 + tid
    const long ar = (long)m0 - h2 - h1 + tid;
    const bool in_w = tid < WR && ar >= 0 && ar < p.a_rows;
    const long arc = ar < 0 ? 0 : (ar < p.a_rows ? ar : p.a_rows - 1);
    const int mk = ((gbytes)(p.rowmask ? p.rowmask : hc_ones_page))[p.rowmask ? arc : 0];
    const float am_w = p.amax_in[slot_of(arc)];
    const long yr = (long)m0 - h2 + (tid < R ? tid : 0);
    const bool in_y = yr >= 0 && yr < p.M;
    const long yrc = yr < 0 ? 0 : (yr < p.M ? yr : (long)p.M - 1);
    const int mky = ((gbytes)(p.rowmask ? p.rowmask : hc_ones_page))[p.rowmask ? yrc : 0];
    const float am_y = p.amax_in[slot_of(yrc)];
    const long mt = (long)m0 + (tid < R ? tid : 0);
    const long mc = mt < p.M ? mt : (long)p.M - 1;
    const int sl_o = slot_of(mc);
    const int trk = ((gbytes)(p.rowmask ? p.rowmask : hc_ones_page))[p.rowmask ? mc : 0];
    const float am_o = p.amax_in[sl_o];
    if (tid < WRP) wscale[tid] = (in_w && mk != 0) ? h3_scale_dev(am_w + p.e1) : -1.f;
    if (tid < R) {
      // the intermediate's scale from its BOUND (header): a function of the utterance's measured bound of x alone
      yinv1[tid] = 1.0f / h3_scale_dev(am_y + p.e1);
      ysc2[tid] = (in_y && mky != 0) ? h3_scale_dev(p.l1max * (am_y + p.e1) + p.b1max + p.e2) : -1.f;
      const bool ok = tid < RO && mt < p.M;
      const float inv = ok ? 1.0f / h3_scale_dev(p.l1max * (am_o + p.e1) + p.b1max + p.e2) : 0.f;
      rowinfo[tid] = int2{(int)__float_as_uint(inv), sl_o | ((p.amax_out && ok && trk != 0) ? RI_TRACK : 0)};
    }
  }
  __syncthreads();

  // ---- the window, once: fp32 rows -> Snake1 -> x scale -> two fp16 planes, chunk-major operand image (hiftconv_kernel) ----
  {
    const int c4 = tid % C4, r0 = tid / C4;      // this thread's four channels, its first window row
    const rg_f32x4 al = *reinterpret_cast<const rg_f32x4*>(p.alpha1 + 4 * c4);
    rg_f32x4 ai;
#pragma unroll
    for (int e = 0; e < 4; ++e) ai[e] = 1.0f / (al[e] + 1e-9f);
    const int chunk = c4 >> 3, cslot = (c4 & 7) >> 1, chalf = (c4 & 1) << 3;
    const float* const abase = p.A + ((long)m0 - h2 - h1) * C + 4 * c4;
    constexpr int U = 4;
    for (int rb = r0; rb < WR; rb += U * RPI) {
      rg_f32x4 x[U];
      float sc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = rb + u * RPI;
        sc[u] = r < WR ? wscale[r] : -1.f;
        const float* src = sc[u] >= 0.f ? abase + (long)r * C : hc_zero_page;
        x[u] = *(const __attribute__((address_space(1))) rg_f32x4*)src;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = rb + u * RPI;
        if (r >= WR) continue;
        rg_f32x4 v = x[u];
        if (sc[u] >= 0.f) {
          float arg[4];
          bool big = false;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            arg[e] = v[e] * al[e];
            big = big || fabsf(arg[e]) > 32768.f;      // false for NaN, which sin2_small propagates
          }
          if (snake_args_small(big)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] + ai[e] * sin2_small(arg[e]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float sn = sinf(arg[e]);
              v[e] = v[e] + ai[e] * (sn * sn);
            }
          }
          v = v * sc[u];
        } else {
          v = rg_f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const Split2 s0 = split2h_pair(v[0], v[1]);
        const Split2 s1 = split2h_pair(v[2], v[3]);
        unsigned char* d = img + chunk * CS + r * 64 + (((cslot ^ rg_key(r))) << 4) + chalf;
        *reinterpret_cast<rg_u32x2*>(d) = rg_u32x2{s0.h, s1.h};
        *reinterpret_cast<rg_u32x2*>(d + PS) = rg_u32x2{s0.l, s1.l};
      }
    }
  }
  // per-column constants: this lane's columns in the MFMA layout (the intermediate) and in the row-wise passes (the epilogue)
  float cs1c[2], b1c[2], al2c[2], ai2c[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int col = 32 * cp + 16 * nt + r16;
    cs1c[nt] = p.cs1[col];
    b1c[nt] = p.b1 ? p.b1[col] : 0.f;
    al2c[nt] = p.alpha2[col];
    ai2c[nt] = 1.0f / (al2c[nt] + 1e-9f);
  }
  const int ncol = 32 * cp + 4 * (lane & 7);
  rg_f32x4 cs4 = *reinterpret_cast<const rg_f32x4*>(p.cs2 + ncol);
  rg_f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
  if (p.b2) b4 = *reinterpret_cast<const rg_f32x4*>(p.b2 + ncol);
  rg_wait_vmcnt<0>();
  landed_w(bq[0][0][0], bq[0][0][1], bq[0][1][0], bq[0][1][1]);
  landed_w(bq[1][0][0], bq[1][0][1], bq[1][1][0], bq[1][1][1]);
  if constexpr (NB == 3) landed_w(bq[NB - 1][0][0], bq[NB - 1][0][1], bq[NB - 1][1][0], bq[NB - 1][1][1]);
  rg_lds_barrier();      // the window's image is complete

  // ---- the two products: step ks = (tap j = ks / NCH, chunk c = ks % NCH); A fragments of row tile mt at image row
  // 80 grp + 16 mt + r16 + j dil (the 16-byte slot key depends on r16 + j dil alone) ----
  rg_f32x4 acc[RT][2];
  rg_u32x4 af[2][RT][2];
  int ips = PS, ics = CS, idil = p.dil;      // the image being read: plane / chunk strides, row step per tap
  auto read_a = [&](auto par_tag, const int c, const int j) {
    constexpr int par = decltype(par_tag)::value;
    const int lrow = r16 + j * idil;
    const unsigned char* const a = img + grp * RG * 64 + c * ics + lrow * 64 + ((kq ^ rg_key(lrow)) << 4);
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) af[par][mt][pl] = *reinterpret_cast<const rg_u32x4*>(a + pl * ips + mt * 1024);
  };
  int nc = 1, nj = 0;      // chunk and tap of the NEXT step
  // a step multiplies the A fragments of parity `par` (requested by the step before) with the weight buffer `buf` and
  // refills that buffer for the step NB ahead
  auto step = [&](auto buf_tag, auto par_tag, const bool last) {
    constexpr int buf = decltype(buf_tag)::value, par = decltype(par_tag)::value;
    auto block = [&](auto nttag) {
      constexpr int nt = decltype(nttag)::value;
#pragma unroll
      for (int mt = 0; mt < RT; ++mt) {
        rg_f32x4 t = acc[mt][nt];
        auto mm = [&](const rg_u32x4& x, const rg_u32x4& y) {
          t = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(rg_f16x8, x), __builtin_bit_cast(rg_f16x8, y), t, 0, 0, 0);
        };
        mm(af[par][mt][1], bq[buf][nt][0]);      // smallest terms first, as everywhere
        mm(af[par][mt][0], bq[buf][nt][1]);
        mm(af[par][mt][0], bq[buf][nt][0]);
        acc[mt][nt] = t;
      }
      __builtin_amdgcn_sched_barrier(0);
      load_w(buf_tag, nttag);
      __builtin_amdgcn_sched_barrier(0);
    };
    // this wave's memory operations in program order: ... W0(s+NB-1), W1(s+NB-1) | W0(s+NB), <wait>, W1(s+NB) | ...; needed at the
    // wait: W0(s+1) and W1(s); behind W0(s+1): W1(s+1) .. W0(s+NB) = 2 (NB - 1) load pairs (rowgemm_wa_kernel's count at NB = 2)
    block(std::integral_constant<int, 0>{});
    rg_wait_vmcnt<NWL * (NB - 1)>();
    landed_w(bq[buf][1][0], bq[buf][1][1], bq[(buf + 1) % NB][0][0], bq[(buf + 1) % NB][0][1]);
    if (!last) read_a(std::integral_constant<int, par ^ 1>{}, nc, nj);
    __builtin_amdgcn_sched_barrier(0);
    block(std::integral_constant<int, 1>{});
    advance_w();
    if (++nc == NCH) { nc = 0; ++nj; }
  };
  // product P0 / KS1 (P0 = its first step's number over both products): weight buffer (P0 + i) % NB, A parity i & 1 (KS1 is even)
  auto product = [&](auto p0_tag) {
    constexpr int P0 = decltype(p0_tag)::value;
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = rg_f32x4{0.f, 0.f, 0.f, 0.f};
    nc = 1; nj = 0;
    read_a(std::integral_constant<int, 0>{}, 0, 0);
    hp_for<KS1>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      step(std::integral_constant<int, (P0 + i) % NB>{}, std::integral_constant<int, i & 1>{}, i + 1 >= KS1);
    });
  };
  product(std::integral_constant<int, 0>{});      // the intermediate, before its bias, x scale1 x 2^e

  // ---- the intermediate: + bias -> mask -> Snake2 -> x scale2 -> planes, over the window's image ----
  rg_lds_barrier();      // every wave is done reading the window
  {
    unsigned char* const hs = img + cp * CS2 + grp * RG * 64;      // this wave's 32 channels are chunk cp, its rows group grp
    const int key = (kq & 1) << 1;      // rg_key of image row 80 grp + 16 mt + 4 kq + e (bit 2 of it is bit 0 of kq)
    const int odd = r16 & 1;
    // v_perm_b32 selectors (rowblock_kernel.h's GELU pass): even lane: (mine.lo16 | partner.lo16 << 16), odd: (partner.hi16 | mine.hi16 << 16)
    const unsigned sel = odd ? 0x03020706u : 0x05040100u;
#pragma unroll
    for (int mt = 0; mt < RT; ++mt) {
      const int i0 = grp * RG + mt * 16 + kq * 4;      // this lane's four intermediate rows
      const rg_f32x4 inv4 = *reinterpret_cast<const rg_f32x4*>(yinv1 + i0);
      const rg_f32x4 sc4 = *reinterpret_cast<const rg_f32x4*>(ysc2 + i0);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int slot = (2 * nt + (r16 >> 3)) ^ key;
        float g4[4], arg[4];
        bool big = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          g4[e] = acc[mt][nt][e] * (cs1c[nt] * inv4[e]) + b1c[nt];
          arg[e] = g4[e] * al2c[nt];
          big = big || (sc4[e] >= 0.f && fabsf(arg[e]) > 32768.f);
        }
        if (snake_args_small(big)) {
#pragma unroll
          for (int e = 0; e < 4; ++e) g4[e] = g4[e] + ai2c[nt] * sin2_small(arg[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float sn = sinf(arg[e]);
            g4[e] = g4[e] + ai2c[nt] * (sn * sn);
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) g4[e] = sc4[e] >= 0.f ? g4[e] * sc4[e] : 0.f;      // (selected, never multiplied)
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const Split2 sp = split2h_pair(g4[e], g4[e + 1]);      // (row e | row e + 1 << 16) of this lane's column
          const unsigned ph = (unsigned)__builtin_amdgcn_update_dpp(0, (int)sp.h, 0xB1, 0xf, 0xf, true);      // the neighbour's
          const unsigned pl = (unsigned)__builtin_amdgcn_update_dpp(0, (int)sp.l, 0xB1, 0xf, 0xf, true);
          const unsigned wh = __builtin_amdgcn_perm(ph, sp.h, sel), wl = __builtin_amdgcn_perm(pl, sp.l, sel);
          const int row = mt * 16 + kq * 4 + e + odd;
          unsigned char* d = hs + row * 64 + (slot << 4) + (r16 & 6) * 2;
          *reinterpret_cast<unsigned*>(d) = wh;
          *reinterpret_cast<unsigned*>(d + PS2) = wl;
        }
      }
    }
  }
  rg_lds_barrier();      // the intermediate's image is complete
  ips = PS2; ics = CS2; idil = 1;
  product(std::integral_constant<int, KS1>{});

  // ---- epilogue, per wave (hiftconv_kernel): 16 rows at a time through the wave's private patch; out = ((acc + b2) + x + res2)
  // * out_scale (+ previous out) for tile rows < RO ----
  {
    const int prow = lane >> 3;
    const float* const r1b = p.A + ncol;
    const float* const r2b = p.res2 ? p.res2 + ncol : nullptr;
    float* const ob = p.out + ncol;
    const int rowb = grp * RG;
    struct RowIn { rg_f32x4 r1[2], r2[2], pv[2]; float inv[2]; int info[2]; bool ok[2]; };
    auto request = [&](const int mt, RowIn& in) {      // the two 8-row halves of row tile mt
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const int trow = rowb + mt * 16 + ps * 8 + prow;
        const long m = (long)m0 + trow;
        in.ok[ps] = trow < RO && m < p.M;
        const long mc = in.ok[ps] ? m : (m < p.M ? m : (long)p.M - 1);
        const int2 ri = rowinfo[trow];
        in.inv[ps] = __uint_as_float((unsigned)ri.x);
        in.info[ps] = ri.y;
        const rg_f32x4 z = {0.f, 0.f, 0.f, 0.f};
        in.r1[ps] = *(const __attribute__((address_space(1))) rg_f32x4*)(r1b + mc * C);
        in.r2[ps] = r2b ? *(const __attribute__((address_space(1))) rg_f32x4*)(r2b + mc * C) : z;
        in.pv[ps] = p.accumulate ? *(const __attribute__((address_space(1))) rg_f32x4*)(ob + mc * C) : z;
      }
    };
    auto finish = [&](const int mt, const RowIn& in) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) patch[(kq * 4 + e) * 36 + nt * 16 + r16] = acc[mt][nt][e];
      unsigned u = 0u;
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const int trow = rowb + mt * 16 + ps * 8 + prow;
        const long m = (long)m0 + trow;
        rg_f32x4 x = *reinterpret_cast<const rg_f32x4*>(patch + (ps * 8 + prow) * 36 + 4 * (lane & 7));
        x = x * in.inv[ps];      // 1 / the power of two the row's utterance was staged with
        rg_f32x4 t = x * cs4 + b4;
        t = (t + in.r1[ps]) + in.r2[ps];
        const rg_f32x4 res = t * p.out_scale + in.pv[ps];
        if (in.ok[ps]) {
          *(__attribute__((address_space(1))) rg_f32x4*)(ob + m * C) = res;
          if (in.info[ps] & RI_TRACK) {
#pragma unroll
            for (int e = 0; e < 4; ++e) u = max(u, __float_as_uint(res[e]) & 0x7fffffffu);
          }
        }
      }
      if (p.amax_out) {
        // the 16 rows of a tile almost always belong to one utterance: one wave-level maximum, one atomic -- and none once the
        // slot already holds a larger value (hiftconv_kernel).  Rows of two utterances: per lane.
        const int s_lo = rowinfo[rowb + mt * 16].y & (RI_TRACK - 1), s_hi = rowinfo[rowb + mt * 16 + 15].y & (RI_TRACK - 1);
        if (s_lo == s_hi) {
          const unsigned seen = *(const __attribute__((address_space(1))) unsigned*)(p.amax_out + s_lo);
          if (__builtin_amdgcn_ballot_w64(u > seen) != 0) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) u = max(u, (unsigned)__shfl_xor((int)u, o));
            if (lane == 0) hc_atomic_max(p.amax_out + s_lo, u);
          }
        } else {
#pragma unroll
          for (int ps = 0; ps < 2; ++ps) {
            unsigned v = 0u;
            const int trow = rowb + mt * 16 + ps * 8 + prow;
            const long m = (long)m0 + trow;
            if (in.ok[ps] && (in.info[ps] & RI_TRACK)) {
              const rg_f32x4 res = *(const __attribute__((address_space(1))) rg_f32x4*)(ob + m * C);      // (this lane's own store)
#pragma unroll
              for (int e = 0; e < 4; ++e) v = max(v, __float_as_uint(res[e]) & 0x7fffffffu);
              hc_atomic_max(p.amax_out + (in.info[ps] & (RI_TRACK - 1)), v);
            }
          }
        }
      }
    };
    RowIn cur, nxt;
    request(0, cur);
#pragma unroll
    for (int mt = 0; mt < RT; ++mt) {
      if (mt + 1 < RT) request(mt + 1, nxt);      // a tile ahead: its loads travel while this one is finished
      finish(mt, cur);
      cur = nxt;
    }
  }
  // the wrapped-around W loads of the last two steps: bq stays reserved until they have landed (rowgemm_wd_kernel)
  rg_wait_vmcnt<0>();
  landed_w(bq[0][0][0], bq[0][0][1], bq[0][1][0], bq[0][1][1]);
  landed_w(bq[1][0][0], bq[1][0][1], bq[1][1][0], bq[1][1][1]);
  if constexpr (NB == 3) landed_w(bq[NB - 1][0][0], bq[NB - 1][0][1], bq[NB - 1][1][0], bq[NB - 1][1][1]);
}

}  // namespace jv
