// One launch for a whole CausalResnetBlock1D of the estimator (jyutvoice/flow/decoder.py:98-115, 767-795):
//   h2  = Mish(LayerNorm(conv3_causal(x * mask))) * mask + mlp(t)            (block1 + the time embedding)
//   out = Mish(LayerNorm(conv3_causal(h2 * mask))) * mask + res_conv(x * mask)   (block2 + the 1 x 1 residual convolution)
// on the row-owning skeleton of rowconv_wd_kernel / rowblock_kernel (one 8-wave workgroup per CU, weights in fragment order
// through the register double buffer, v_mfma_f32_16x16x32_f16, fp16x3).
//
// Why.  As two rowconv launches (block1 with res_conv folded in, then block2) a resnet takes 52.7 + 45.7 us where a
// workgroup's stamped work is ~37 + ~32: every launch pays ~14 us of ramp and tail that the phase stamps do not see, and h2
// (20 MB) travels to HBM and back in between.  Here the workgroup that owns R = 16 RT rows of h2 keeps them in LDS:
//   1. product 1 -- rowconv_wd_kernel<RT, true>'s loop: the three taps of block1 into `acc`, res_conv's chunk (a fourth
//      fragment step, at tap 2's row offset) into `accr`; the window of x is split per 32-channel chunk with the
//      per-utterance measured scale, double-buffered in LDS;
//   2. epilogue 1 -- acc through the slab, row-wise: + bias, LayerNorm, Mish, mask, + time embedding -> x scale2 -> fp16
//      planes INTO LDS as block2's operand image (rowblock_kernel's X layout: [k-step][plane][row][64 B]); accr through the
//      slab -> res_conv's rows of this wave's OUTPUT rows, kept in registers;
//   3. product 2 -- block2's 24 steps over the resident image (row offsets 0, 1, 2), A fragments prefetched a step ahead;
//   4. epilogue 2 -- rowconv_wd_kernel's: + bias, LayerNorm, Mish, mask, + res -> out rows (+ the following block's
//      LayerNorm1 planes, + measured-bound tracking);
//   5. (QKV) product 3 -- the following block's to_q | to_k | to_v over those LayerNorm1 planes, which then stay in LDS (the
//      image of product 2 is free): rowblock_kernel's phase C, six 256-column chunks of 8 steps, per-wave chunk epilogues
//      through patches over the slab, q as fp32 rows, k and v as scaled fp16 planes -- the stage's first attention follows
//      the resnet directly (as a launch of its own, rowgemm_wa_kernel, the same product took 48.7 us where phase C's
//      stamps show 31).  Same K order and epilogue expressions as rowgemm_wa_kernel: the same bits.
// block2's causal window needs h2 rows m - 2 .. m: the workgroup computes R rows of h2 (global rows m0 - 2 .. m0 + R - 3) and
// RO = R - 2 output rows (m0 .. m0 + R - 3); 2.5 % of product 1 is recomputed by the neighbour (250 workgroups instead of 244
// at the benchmarked size: still one round of the chip).
//
// LDS (32 RT KB, all of it at RT = 5): lower half = product 1's window buffers, then the h2 image; upper half = the slab
// (unpadded, XOR-swizzled: rowblock_kernel's).  Per-row facts live in registers (lane j holds row j of its wave).
//
// h2's fp16x3 scale.  Unfused, block1's launch measures max |h2| per utterance and block2's derives its power of two from
// it.  Here the scale must exist before h2 does, so it comes from a BOUND: |Mish(LayerNorm(.))| <= sqrt(255) max|g| + max|b|
// (load time, registry.hip -- Mish(v) <= max(v, 0.31)) plus max |mlp(t)| of this step's embedding (256 values, reduced at
// kernel start).  Overflow stays impossible by construction, the bound is a few times looser than a measurement (22
// significant bits either way, the absolute floor 2^-40 of the bound moves with it), and it depends on the weights and on t
// alone: batch invariance holds a fortiori.  Fused and unfused agree to the cross-regime bound (tests).
#pragma once
#include "rowblock_kernel.h"
#include "rowconv_kernel.h"

namespace jv {

struct RowResArgs {
  const float* A;      // the resnet's input x, fp32 rows [a_rows, lda]
  long lda, a_rows;
  int M, Cin;          // output rows; input channels (256, 512: a multiple of 64)
  const unsigned char* rowmask;      // per row (or null): 0 = padding -- reads as zero as x and as h2, written as zero before the residual
  const float* amax_in;              // per-utterance measured bound of x
  int slot_G, slot_S, slot_nb;       // slot(row) = clamp((row - slot_G) / slot_S, 0, slot_nb - 1); slot_S < 0: row_slot[row]
  const int* row_slot;
  // block1 | res_conv: fragment order over K = 4 Cin (ResnetW::wf4)
  const unsigned short* Wf1;
  long wf1_plane;
  const float *cs1, *b1, *ln1_g, *ln1_b;
  const float *csr, *br;             // res_conv's column scales and bias
  const float* temb;                 // [256]: this step's time embedding of this resnet (the same for every row)
  float h2_bound;                    // sqrt(255) max |ln1_g| + max |ln1_b| (load time)
  float ln_eps;
  // block2: fragment order over K = 3 x 256
  const unsigned short* Wf2;
  long wf2_plane;
  const float *cs2, *b2, *ln2_g, *ln2_b;
  float* out;                        // [rows, ldo]
  long ldo;
  float* amax_out;                   // tracking of what is stored (rows with rowmask == 0 excluded), slot as above
  // the following transformer block's LayerNorm1 of the stored row, x lnf_scale, as fp16 planes [2][rows][256] (RowConvArgs::ln2_out)
  unsigned short* lnf_out;
  long lnf_plane;
  const float *lnf_g, *lnf_b;
  float lnf_scale;
  // (rowres_kernel<RT, true>) the following block's q | k | v from those planes (RowBlockArgs' phase C: N = 1536, K = 256, no bias)
  const unsigned short* Wqf;         // fragment order, plane stride wqf_plane halves
  long wqf_plane;
  const float* csq;
  float* q;                          // fp32 rows [., 512]
  unsigned short* kv2;               // planes [2][rows][1024]: k * k_scale in columns 0..511, v * v_scale in 512..1023
  long kv2_plane;
  float k_scale, v_scale;
  long alg_rows;
};

template <int RT> constexpr int rr_lds_bytes() { return 16 * rgw_stage_bytes<RT>(); }

template <int RT, bool QKV>
__global__ __launch_bounds__(512, 2) void rowres_kernel(const RowResArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rr_lds[];
  constexpr int R = 16 * RT, RO = R - 2, WR = R + 2;
  constexpr int STAGE = rgw_stage_bytes<RT>();      // one 32-deep k-step of R rows as two fp16 planes
  constexpr int XP = R * 64;                        // plane stride inside a stage
  constexpr int A_PLANE = rc_a_plane<RT>(), A_BUF = 2 * A_PLANE;      // product 1's window buffers (in the lower half)
  static_assert(2 * A_BUF <= 8 * STAGE, "the window buffers fit the lower half");
  constexpr int NI = (WR * 8 + 511) / 512;
  constexpr int NWL = 4;
  constexpr int NRW = 2 * RT;                       // rows per wave in the row passes
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.x * RO;                   // first OUTPUT row; h2 row i <-> global row m0 - 2 + i; window row w <-> m0 - 4 + w
  const int NCH = p.Cin >> 5;
  float* const slab = reinterpret_cast<float*>(rr_lds + 8 * STAGE);

  // ---- L2 warm-up of block1 | res_conv's fragments (rowgemm_kernel.h); block2's are touched during product 1 ----
  const int wgrp = blockIdx.x >> 3, wngrp = (gridDim.x + 7) >> 3;
  auto warm_lines = [&](const unsigned short* base, long plane_halves, long lines_per_plane) -> float {
    const long per = (2 * lines_per_plane + wngrp - 1) / wngrp;
    const long l = (long)wgrp * per + tid;
    float v = 0.f;
    if (tid < per && l < 2 * lines_per_plane) {
      const int pl = l >= lines_per_plane;
      v = *(const __attribute__((address_space(1))) float*)(reinterpret_cast<const char*>(base + (long)pl * plane_halves) + ((l - pl * lines_per_plane) << 7));
    }
    return v;
  };
  float warm = warm_lines(p.Wf1, p.wf1_plane, ((long)256 * 4 * p.Cin * 2) >> 7), warm2 = 0.f;

  // ---- W walker: one register double buffer for both products (rowgemm_wd_kernel's rules) ----
  const unsigned short* wbase[2][2];
  auto set_wbase = [&](const unsigned short* wf, long plane) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) wbase[nt][pl] = wf + (long)pl * plane + (long)(wave * 2 + nt) * 512 + lane * 8;
  };
  set_wbase(p.Wf1, p.wf1_plane);
  long woff = 0;       // halves: fragment step of the NEXT step to load, x 16 column blocks x 512
  int wj = 0, wc = 0;  // its tap and chunk
  int wnj = 4, wnch = NCH;      // taps per chunk and chunks of the matrix being walked (product 1: 4 x NCH; product 2: 3 x 8)
  rg_u32x4 bq[2][2][2];
#pragma unroll
  for (int i = 0; i < 8; ++i) bq[i >> 2][(i >> 1) & 1][i & 1] = rg_u32x4{0u, 0u, 0u, 0u};
  auto load_frag = [](rg_u32x4& dst, const unsigned short* ptr) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(dst) : "v"(ptr) : "memory");
  };
  auto load_w = [&](auto par_tag, auto nttag) {
    constexpr int par = decltype(par_tag)::value, nt = decltype(nttag)::value;
    load_frag(bq[par][nt][0], wbase[nt][0] + woff);
    load_frag(bq[par][nt][1], wbase[nt][1] + woff);
  };
  auto landed_w = [](rg_u32x4& b00, rg_u32x4& b01, rg_u32x4& b10, rg_u32x4& b11) {
    asm volatile("" : "+v"(b00), "+v"(b01), "+v"(b10), "+v"(b11)::"memory");
  };
  auto advance_w = [&]() {      // step (c, j) is fragment step j nch + c; (c, nj - 1) -> (c + 1, 0); past the end: back to 0
    if (++wj == wnj) {
      wj = 0;
      if (++wc == wnch) { wc = 0; woff = 0; }
      else woff += (1L - (long)(wnj - 1) * wnch) * (16 * 512);
    } else {
      woff += (long)wnch * (16 * 512);
    }
  };
  load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
  advance_w();
  load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
  load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
  advance_w();

  // ---- per-row facts, in registers: lane j < NRW holds those of the wave's row j (broadcast with v_readlane in the row passes)
  //   pass 1, h2 row i = wave NRW + j <-> global gi = m0 - 2 + i:  inv1 = 1 / scale of x there, live = a real frame
  //   pass 2, output row t = wave NRW + j <-> global m0 + t:         ok, keep (real frame), slot, 1 / scale of x there (res_conv)
  typedef const __attribute__((address_space(1))) unsigned char* gbytes;
  auto slot_of = [&](const long row) -> int {
    if (p.slot_S < 0) {
      const int q = p.row_slot[row];
      return q < 0 ? 0 : (q >= p.slot_nb ? p.slot_nb - 1 : q);
    }
    if (p.slot_S == 0) return 0;
    const int q = (int)((row - p.slot_G) / p.slot_S);
    return q < 0 ? 0 : (q >= p.slot_nb ? p.slot_nb - 1 : q);
  };
  float f_inv1 = 0.f, f_invo = 0.f;
  int f_live = 0, f_ok = 0, f_keep = 0, f_slot = 0;
  unsigned f_seen = 0xffffffffu;
  {
    const int j = lane < NRW ? lane : 0;
    const long gi = (long)m0 - 2 + wave * NRW + j;
    const long gic = gi < 0 ? 0 : (gi < p.M ? gi : (long)p.M - 1);
    const int mk_i = ((gbytes)(p.rowmask ? p.rowmask : rc_ones_page))[p.rowmask ? gic : 0];
    const float am_i = p.amax_in[slot_of(gic)];
    const int t = wave * NRW + j;
    const long go = (long)m0 + t;
    const long goc = go < p.M ? go : (long)p.M - 1;
    const int mk_o = ((gbytes)(p.rowmask ? p.rowmask : rc_ones_page))[p.rowmask ? goc : 0];
    const int sl_o = slot_of(goc);
    const float am_o = p.amax_in[sl_o];
    f_inv1 = 1.0f / h3_scale_dev(am_i);
    f_live = (gi >= 0 && gi < p.M && mk_i != 0) ? 1 : 0;
    f_ok = (t < RO && go < p.M) ? 1 : 0;
    f_keep = (f_ok && mk_o != 0) ? 1 : 0;
    f_slot = sl_o;
    f_invo = 1.0f / h3_scale_dev(am_o);
    if (p.amax_out) f_seen = *reinterpret_cast<const unsigned*>(p.amax_out + sl_o);      // (a maximum read early is a valid lower bound)
  }

  // ---- product 1's window staging (rowconv_wd_kernel): thread -> (window row, float4) pairs, fixed over the chunks ----
  const float* asrc[NI];
  int adst[NI];
  float ascale[NI];      // 0: the row reads as zero (outside the buffer, or masked)
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int idx = tid + 512 * i;
    const int r = idx >> 3, c4 = idx & 7;
    const long ar = (long)m0 - 4 + r;
    const bool ok = r < WR && ar >= 0 && ar < p.a_rows;
    const long arc = ar < 0 ? 0 : (ar < p.a_rows ? ar : p.a_rows - 1);
    asrc[i] = p.A + arc * p.lda + 4 * c4;
    adst[i] = r < WR ? r * 64 + ((((c4 >> 1) ^ rg_key(r)) << 4) | ((c4 & 1) << 3)) : -1;
    const int mk = ((gbytes)(p.rowmask ? p.rowmask : rc_ones_page))[p.rowmask ? arc : 0];
    const float am = p.amax_in[slot_of(arc)];
    ascale[i] = (ok && mk != 0) ? h3_scale_dev(am) : 0.f;
  }
  rg_f32x4 pa[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) pa[i] = rg_f32x4{0.f, 0.f, 0.f, 0.f};
  auto load_A = [&](int c) {
#pragma unroll
    for (int i = 0; i < NI; ++i) asm volatile("global_load_dwordx4 %0, %1, off ; window rows" : "+v"(pa[i]) : "v"(asrc[i] + c * 32) : "memory");
  };
  auto landed_a = [&]() {
#pragma unroll
    for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(pa[i])::"memory");
  };
  auto store_A = [&](int buf) {
    unsigned char* const base = rr_lds + buf * A_BUF;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (adst[i] >= 0) {
        const bool use = ascale[i] != 0.f;      // (selected, never multiplied: a row that reads as zero may hold anything)
        const Split2 s0 = split2h_pair(use ? pa[i][0] * ascale[i] : 0.f, use ? pa[i][1] * ascale[i] : 0.f);
        const Split2 s1 = split2h_pair(use ? pa[i][2] * ascale[i] : 0.f, use ? pa[i][3] * ascale[i] : 0.f);
        *reinterpret_cast<rg_u32x2*>(base + adst[i]) = rg_u32x2{s0.h, s1.h};
        *reinterpret_cast<rg_u32x2*>(base + A_PLANE + adst[i]) = rg_u32x2{s0.l, s1.l};
      }
    }
  };
  load_A(0);

  rg_f32x4 acc[RT][2], accr[RT][2];
#pragma unroll
  for (int mt = 0; mt < RT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      acc[mt][nt] = rg_f32x4{0.f, 0.f, 0.f, 0.f};
      accr[mt][nt] = rg_f32x4{0.f, 0.f, 0.f, 0.f};
    }
  rg_wait_vmcnt<0>();      // the first window's rows, the facts, the first fragments
  landed_a();
  store_A(0);
  if (NCH > 1) load_A(1);
  if (NCH > 1) rg_wait_vmcnt<NI>(); else rg_wait_vmcnt<0>();
  landed_w(bq[0][0][0], bq[0][0][1], bq[0][1][0], bq[0][1][1]);
  landed_w(bq[1][0][0], bq[1][0][1], bq[1][1][0], bq[1][1][1]);

  // ================================ product 1 (rowconv_wd_kernel<RT, true>'s step) ================================
  {
    auto step = [&](auto par_tag, const int c, auto jtag) {
      constexpr int par = decltype(par_tag)::value, j = decltype(jtag)::value;
      constexpr int jrow = j < 3 ? j : 2;      // res_conv's step reads the h2 row's own row of x: tap 2's row offset
      if (j == 0) {
        rg_lds_barrier();      // every thread's plane stores of this chunk's window are complete; the other buffer is free
        if (c + 1 < NCH) {
          if (c == 0) rg_wait_vmcnt<0>();
          landed_a();
          store_A((c + 1) & 1);
        }
      }
      const unsigned char* const sa = rr_lds + (c & 1) * A_BUF;
      rg_u32x4 af[RT][2];
#pragma unroll
      for (int mt = 0; mt < RT; ++mt) {
        const int row = mt * 16 + r16 + jrow;
        const int a_off = row * 64 + ((kq ^ rg_key(row)) << 4);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) af[mt][pl] = *reinterpret_cast<const rg_u32x4*>(sa + pl * A_PLANE + a_off);
      }
      __builtin_amdgcn_sched_barrier(0);
      auto block = [&](auto nttag, rg_f32x4 (&ac)[RT][2]) {
        constexpr int nt = decltype(nttag)::value;
#pragma unroll
        for (int mt = 0; mt < RT; ++mt) {
          rg_f32x4 t = ac[mt][nt];
          auto mm = [&](const rg_u32x4& x, const rg_u32x4& y) {
            t = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(rg_f16x8, x), __builtin_bit_cast(rg_f16x8, y), t, 0, 0, 0);
          };
          mm(af[mt][1], bq[par][nt][0]);
          mm(af[mt][0], bq[par][nt][1]);
          mm(af[mt][0], bq[par][nt][0]);
          ac[mt][nt] = t;
        }
        __builtin_amdgcn_sched_barrier(0);
        load_w(par_tag, nttag);
        __builtin_amdgcn_sched_barrier(0);
      };
      auto blocks = [&](rg_f32x4 (&ac)[RT][2]) {
        block(std::integral_constant<int, 0>{}, ac);
        if (j == 1 && c + 2 < NCH) rg_wait_vmcnt<NWL + NI>(); else rg_wait_vmcnt<NWL>();
        landed_w(bq[par][1][0], bq[par][1][1], bq[par ^ 1][0][0], bq[par ^ 1][0][1]);
        block(std::integral_constant<int, 1>{}, ac);
      };
      if constexpr (j == 3) blocks(accr); else blocks(acc);
      advance_w();
      if (j == 0 && c + 2 < NCH) load_A(c + 2);      // behind this step's weight loads (the wait counts above rely on it)
    };
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
      if (c == (NCH >> 1)) warm2 = warm_lines(p.Wf2, p.wf2_plane, ((long)256 * 3 * 256 * 2) >> 7);      // block2's weights, half a product ahead
      step(std::integral_constant<int, 0>{}, c, std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{}, c, std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 0>{}, c, std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 1>{}, c, std::integral_constant<int, 3>{});
    }
  }
  // per-column constants of epilogue 1 and the time embedding, requested HERE (not ahead of product 1: its loop has no registers
  // to carry 28 values through -- they were spilled, each reload behind a vmcnt(0)); their latency passes under the wait below
  const rg_f32x4 cs1 = *reinterpret_cast<const rg_f32x4*>(p.cs1 + 4 * lane);
  rg_f32x4 b1 = {0.f, 0.f, 0.f, 0.f};
  if (p.b1) b1 = *reinterpret_cast<const rg_f32x4*>(p.b1 + 4 * lane);
  const rg_f32x4 g1 = *reinterpret_cast<const rg_f32x4*>(p.ln1_g + 4 * lane);
  const rg_f32x4 o1 = *reinterpret_cast<const rg_f32x4*>(p.ln1_b + 4 * lane);
  const rg_f32x4 csr = *reinterpret_cast<const rg_f32x4*>(p.csr + 4 * lane);
  rg_f32x4 br = {0.f, 0.f, 0.f, 0.f};
  if (p.br) br = *reinterpret_cast<const rg_f32x4*>(p.br + 4 * lane);

  // the time embedding's largest magnitude (256 values: four per lane) -> h2's scale, the same for every row
  const rg_f32x4 te4 = *reinterpret_cast<const rg_f32x4*>(p.temb + 4 * lane);
  float tmax = fmaxf(fmaxf(fabsf(te4[0]), fabsf(te4[1])), fmaxf(fabsf(te4[2]), fabsf(te4[3])));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o));
  // (NaN in the embedding: fmaxf drops it here and the planes below carry it -- the result is NaN where the unfused form's is)
  const float scale2 = h3_scale_dev(fmaxf(p.h2_bound, 0.31f) + tmax);
  const float inv2 = 1.0f / scale2;

  // the wrapped-around loads of the last two steps land in registers nothing reads; then block2's first two steps are requested
  rg_wait_vmcnt<0>();
  landed_w(bq[0][0][0], bq[0][0][1], bq[0][1][0], bq[0][1][1]);
  landed_w(bq[1][0][0], bq[1][0][1], bq[1][1][0], bq[1][1][1]);
  asm volatile("" ::"v"(warm), "v"(warm2));
  // (QKV) to_q | to_k | to_v's fragments into this XCD's L2 a whole product ahead: cold, every step of product 3 began with
  // an HBM round trip taken by all workgroups at once -- 47 us for the product, what it takes as a launch of its own
  float warm3 = 0.f;
  if constexpr (QKV) warm3 = warm_lines(p.Wqf, p.wqf_plane, ((long)1536 * 256 * 2) >> 7);
  set_wbase(p.Wf2, p.wf2_plane);
  woff = 0; wj = 0; wc = 0; wnj = 3; wnch = 8;
  load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
  advance_w();
  load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
  load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
  advance_w();

  auto acc_to_slab = [&](const rg_f32x4 (&a)[RT][2]) {
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) slab[rb_slab(mt * 16 + kq * 4 + e, wave * 32 + nt * 16 + r16)] = a[mt][nt][e];
  };
  auto bcast_f = [&](const float v, const int j) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), j)); };

  // ================================ epilogue 1: h2 -> the operand image; res_conv's rows -> registers ================================
  acc_to_slab(acc);      // (the slab is the upper half: nothing of product 1 lives there)
  rg_lds_barrier();      // slab complete; every wave is done with the window buffers: the image goes over them
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    rg_f32x4 v[RT];
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      const int trow = wave * NRW + ps * RT + j;
      v[j] = *reinterpret_cast<const rg_f32x4*>(slab + rb_slab(trow, 4 * lane)) * (cs1 * bcast_f(f_inv1, ps * RT + j)) + b1;
    }
    float sum[RT], sq[RT];
#pragma unroll
    for (int j = 0; j < RT; ++j) sum[j] = wave_sum((v[j][0] + v[j][1]) + (v[j][2] + v[j][3]));
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      const rg_f32x4 d = v[j] - sum[j] * (1.f / 256.f);
      sq[j] = wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]));
    }
    float var_l = sq[0];
#pragma unroll
    for (int j = 1; j < RT; ++j) var_l = lane == j ? sq[j] : var_l;
    const float rstd_l = 1.0f / sqrtf(var_l * (1.f / 256.f) + p.ln_eps);
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      const int trow = wave * NRW + ps * RT + j;
      const float mean = sum[j] * (1.f / 256.f);
      const float rstd = bcast_f(rstd_l, j);
      rg_f32x4 y = (v[j] - mean) * rstd * g1 + o1;
      const bool live = __builtin_amdgcn_readlane(f_live, ps * RT + j) != 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = live ? (mish_fast(y[e]) + te4[e]) * scale2 : 0.f;      // (a padding row reads as zero in block2)
      const Split2 s0 = split2h_pair(y[0], y[1]);
      const Split2 s1 = split2h_pair(y[2], y[3]);
      // columns 4 lane .. + 3 = k-step lane >> 3, 16-byte slot (lane & 7) >> 1 (swizzled), its half lane & 1 (rowblock_kernel's X)
      unsigned char* d = rr_lds + (lane >> 3) * STAGE + trow * 64 + (((((lane & 7) >> 1) ^ rg_key(trow))) << 4) + (lane & 1) * 8;
      *reinterpret_cast<rg_u32x2*>(d) = rg_u32x2{s0.h, s1.h};
      *reinterpret_cast<rg_u32x2*>(d + XP) = rg_u32x2{s0.l, s1.l};
    }
  }
  rg_lds_barrier();      // every wave has read its slab rows: res_conv's product goes through the slab next
  acc_to_slab(accr);
  rg_lds_barrier();
  rg_f32x4 res[NRW];      // res_conv(x * mask) of this wave's output rows t = wave NRW + j: h2 row t + 2
#pragma unroll
  for (int j = 0; j < NRW; ++j) {
    const int t = wave * NRW + j;
    const int srow = t + 2 < R ? t + 2 : R - 1;      // (t >= RO: no output row)
    res[j] = *reinterpret_cast<const rg_f32x4*>(slab + rb_slab(srow, 4 * lane)) * (csr * bcast_f(f_invo, j)) + br;
  }
  rg_wait_vmcnt<0>();      // block2's first fragments
  asm volatile("" ::"v"(warm3));      // (issued an epilogue ago)
  landed_w(bq[0][0][0], bq[0][0][1], bq[0][1][0], bq[0][1][1]);
  landed_w(bq[1][0][0], bq[1][0][1], bq[1][1][0], bq[1][1][1]);
  rg_lds_barrier();      // the image is complete (and the slab has been read: epilogue 2 writes it again)

  // ================================ product 2: block2 over the resident image ================================
#pragma unroll
  for (int mt = 0; mt < RT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = rg_f32x4{0.f, 0.f, 0.f, 0.f};
  {
    rg_u32x4 af[2][RT][2];
    auto read_a = [&](auto par_tag, const int c, const int j) {      // chunk c = stage c of the image, tap j = row offset j
      constexpr int par = decltype(par_tag)::value;
#pragma unroll
      for (int mt = 0; mt < RT; ++mt) {
        const int row = mt * 16 + r16 + j;      // (rows R, R + 1 of the last tile lie in the next plane: garbage into discarded output rows)
        const unsigned char* const a = rr_lds + c * STAGE + row * 64 + ((kq ^ rg_key(row)) << 4);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) af[par][mt][pl] = *reinterpret_cast<const rg_u32x4*>(a + pl * XP);
      }
    };
    int nc = 0, nj = 1;      // chunk and tap of the NEXT step
    auto step = [&](auto par_tag, const bool last) {
      constexpr int par = decltype(par_tag)::value;
      auto block = [&](auto nttag) {
        constexpr int nt = decltype(nttag)::value;
#pragma unroll
        for (int mt = 0; mt < RT; ++mt) {
          rg_f32x4 t = acc[mt][nt];
          auto mm = [&](const rg_u32x4& x, const rg_u32x4& y) {
            t = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(rg_f16x8, x), __builtin_bit_cast(rg_f16x8, y), t, 0, 0, 0);
          };
          mm(af[par][mt][1], bq[par][nt][0]);
          mm(af[par][mt][0], bq[par][nt][1]);
          mm(af[par][mt][0], bq[par][nt][0]);
          acc[mt][nt] = t;
        }
        __builtin_amdgcn_sched_barrier(0);
        load_w(par_tag, nttag);
        __builtin_amdgcn_sched_barrier(0);
      };
      block(std::integral_constant<int, 0>{});
      rg_wait_vmcnt<NWL>();
      landed_w(bq[par][1][0], bq[par][1][1], bq[par ^ 1][0][0], bq[par ^ 1][0][1]);
      if (!last) read_a(std::integral_constant<int, par ^ 1>{}, nc, nj);
      __builtin_amdgcn_sched_barrier(0);
      block(std::integral_constant<int, 1>{});
      advance_w();
      if (++nj == 3) { nj = 0; ++nc; }
    };
    read_a(std::integral_constant<int, 0>{}, 0, 0);
#pragma unroll 1
    for (int ks = 0; ks < 24; ks += 2) {
      step(std::integral_constant<int, 0>{}, false);
      step(std::integral_constant<int, 1>{}, ks + 2 >= 24);
    }
  }

  // ================================ epilogue 2 (rowconv_wd_kernel's row pass) ================================
  // per-column constants of epilogue 2 (requested here, behind product 2 -- its loop has no registers to carry them; the slab write and its barrier cover most of their latency)
  const rg_f32x4 cs2 = *reinterpret_cast<const rg_f32x4*>(p.cs2 + 4 * lane);
  rg_f32x4 b2 = {0.f, 0.f, 0.f, 0.f};
  if (p.b2) b2 = *reinterpret_cast<const rg_f32x4*>(p.b2 + 4 * lane);
  const rg_f32x4 g2 = *reinterpret_cast<const rg_f32x4*>(p.ln2_g + 4 * lane);
  const rg_f32x4 o2 = *reinterpret_cast<const rg_f32x4*>(p.ln2_b + 4 * lane);
  rg_f32x4 gf = {1.f, 1.f, 1.f, 1.f}, of = {0.f, 0.f, 0.f, 0.f};
  if (QKV || p.lnf_out) {
    gf = *reinterpret_cast<const rg_f32x4*>(p.lnf_g + 4 * lane);
    of = *reinterpret_cast<const rg_f32x4*>(p.lnf_b + 4 * lane);
  }
  // (QKV) product 3's walker: chunk qc (256 of the 1536 columns), step qks -> fragment step qks, column blocks 16 qc ..
  int qks = 0, qc = 0;
  auto advance_q = [&]() {      // past the end: wrap around to weights that exist
    if (++qks == 8) { qks = 0; if (++qc == 6) qc = 0; }
    woff = ((long)qks * 96 + qc * 16) * 512;
  };
  acc_to_slab(acc);      // (the slab was last read before product 2's barrier)
  rg_lds_barrier();      // (and every wave is done reading the image: the LayerNorm1 planes go over it below)
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    rg_f32x4 v[RT];
    int drow0 = m0 + wave * NRW + ps * RT;
    asm volatile("" : "+s"(drow0));
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      const int trow = wave * NRW + ps * RT + j;
      v[j] = *reinterpret_cast<const rg_f32x4*>(slab + rb_slab(trow, 4 * lane)) * (cs2 * inv2) + b2;
    }
    {
      float sum[RT], sq[RT];
#pragma unroll
      for (int j = 0; j < RT; ++j) sum[j] = wave_sum((v[j][0] + v[j][1]) + (v[j][2] + v[j][3]));
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        const rg_f32x4 d = v[j] - sum[j] * (1.f / 256.f);
        sq[j] = wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]));
      }
      float var_l = sq[0];
#pragma unroll
      for (int j = 1; j < RT; ++j) var_l = lane == j ? sq[j] : var_l;
      const float rstd_l = 1.0f / sqrtf(var_l * (1.f / 256.f) + p.ln_eps);
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        const float mean = sum[j] * (1.f / 256.f);
        v[j] = (v[j] - mean) * bcast_f(rstd_l, j) * g2 + o2;
      }
    }
    bool ok[RT];
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      const int jj = ps * RT + j;
      ok[j] = __builtin_amdgcn_readlane(f_ok, jj) != 0;
      const bool keep = __builtin_amdgcn_readlane(f_keep, jj) != 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[j][e] = keep ? mish_fast(v[j][e]) : 0.f;
      v[j] = v[j] + res[jj];
      if (ok[j]) *(__attribute__((address_space(1))) rg_f32x4*)(p.out + (long)(drow0 + j) * p.ldo + 4 * lane) = v[j];
    }
    if (QKV || p.lnf_out) {      // (uniform) the following block's norm1 of the stored rows -> operand planes (rowconv_wd_kernel)
      float sum[RT], sq[RT];
#pragma unroll
      for (int j = 0; j < RT; ++j) sum[j] = wave_sum((v[j][0] + v[j][1]) + (v[j][2] + v[j][3]));
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        const rg_f32x4 d = v[j] - sum[j] * (1.f / 256.f);
        sq[j] = wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]));
      }
      float var_l = sq[0];
#pragma unroll
      for (int j = 1; j < RT; ++j) var_l = lane == j ? sq[j] : var_l;
      const float rstd_l = 1.0f / sqrtf(var_l * (1.f / 256.f) + p.ln_eps);
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        const float mean = sum[j] * (1.f / 256.f);
        const rg_f32x4 y = (v[j] - mean) * bcast_f(rstd_l, j) * gf + of;
        const Split2 s0 = split2h_pair(y[0] * p.lnf_scale, y[1] * p.lnf_scale);
        const Split2 s1 = split2h_pair(y[2] * p.lnf_scale, y[3] * p.lnf_scale);
        if constexpr (QKV) {      // the operand image of product 3 (rows >= RO / past the end: never stored, and a row's product reads its own row only)
          const int trow = wave * NRW + ps * RT + j;
          unsigned char* d = rr_lds + (lane >> 3) * STAGE + trow * 64 + (((((lane & 7) >> 1) ^ rg_key(trow))) << 4) + (lane & 1) * 8;
          *reinterpret_cast<rg_u32x2*>(d) = rg_u32x2{s0.h, s1.h};
          *reinterpret_cast<rg_u32x2*>(d + XP) = rg_u32x2{s0.l, s1.l};
        } else if (ok[j]) {
          unsigned short* const o2p = p.lnf_out + (long)(drow0 + j) * 256 + 4 * lane;
          *(__attribute__((address_space(1))) rg_u32x2*)(o2p) = rg_u32x2{s0.h, s1.h};
          *(__attribute__((address_space(1))) rg_u32x2*)(o2p + p.lnf_plane) = rg_u32x2{s0.l, s1.l};
        }
      }
    }
    if (p.amax_out) {
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        const int jj = ps * RT + j;
        unsigned u = 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) u = max(u, __float_as_uint(v[j][e]) & 0x7fffffffu);
        const bool trk = __builtin_amdgcn_readlane(f_keep, jj) != 0;
        const unsigned seen = (unsigned)__builtin_amdgcn_readlane((int)f_seen, jj);
        if (trk && __builtin_amdgcn_ballot_w64(u > seen) != 0) {      // wave-uniform: nothing to do once the slot holds a larger value
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) u = max(u, (unsigned)__shfl_xor((int)u, o));
          if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(p.amax_out + __builtin_amdgcn_readlane(f_slot, jj)), u);
        }
      }
    }
  }

  // ================================ product 3 (QKV): rowblock_kernel's phase C over the LayerNorm1 image ================================
  if constexpr (QKV) {
    // product 2's wrapped-around loads land in registers nothing reads; then q | k | v's first two steps are requested (HERE, behind
    // the row pass: requested ahead of it the 32 fragment registers pushed the pass's values into scratch)
    rg_wait_vmcnt<0>();
    landed_w(bq[0][0][0], bq[0][0][1], bq[0][1][0], bq[0][1][1]);
    landed_w(bq[1][0][0], bq[1][0][1], bq[1][1][0], bq[1][1][1]);
    set_wbase(p.Wqf, p.wqf_plane);
    woff = 0;
    load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    advance_q();
    load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
    load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
    advance_q();
    rg_lds_barrier();      // the image is complete; the slab has been read: the patches go over it
    rg_wait_vmcnt<0>();      // (the fragments' round trip passes under the barrier; the row stores are waited for with them)
    landed_w(bq[0][0][0], bq[0][0][1], bq[0][1][0], bq[0][1][1]);
    landed_w(bq[1][0][0], bq[1][0][1], bq[1][1][0], bq[1][1][1]);
    constexpr long LDQ = 512, LDKV = 1024;
    constexpr int NPATCH = RT >= 3 ? 2 : 1;      // (RT = 2: the slab is 32 KB, eight pairs of patches are 36)
    const float inv_q = 1.0f / p.lnf_scale;
    float* const ws0 = slab + wave * (NPATCH * 16 * 36);
    const int prow = lane >> 3, pc4 = (lane & 7) * 4;
    rg_u32x4 af[2][RT][2];
    const int a_off = r16 * 64 + ((kq ^ rg_key(r16)) << 4);
    auto read_a = [&](auto par_tag, const int ks) {
      constexpr int par = decltype(par_tag)::value;
#pragma unroll
      for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) af[par][mt][pl] = *reinterpret_cast<const rg_u32x4*>(rr_lds + ks * STAGE + a_off + pl * XP + mt * 1024);
    };
    bool waited = true;      // (the wait was done ahead of an epilogue's memory operations: rowblock_kernel's step)
    auto step = [&](auto par_tag, const int ks_next) {
      constexpr int par = decltype(par_tag)::value;
      auto block = [&](auto nttag) {
        constexpr int nt = decltype(nttag)::value;
#pragma unroll
        for (int mt = 0; mt < RT; ++mt) {
          rg_f32x4 t = acc[mt][nt];
          auto mm = [&](const rg_u32x4& x, const rg_u32x4& y) {
            t = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(rg_f16x8, x), __builtin_bit_cast(rg_f16x8, y), t, 0, 0, 0);
          };
          mm(af[par][mt][1], bq[par][nt][0]);
          mm(af[par][mt][0], bq[par][nt][1]);
          mm(af[par][mt][0], bq[par][nt][0]);
          acc[mt][nt] = t;
        }
        __builtin_amdgcn_sched_barrier(0);
        load_w(par_tag, nttag);
        __builtin_amdgcn_sched_barrier(0);
      };
      block(std::integral_constant<int, 0>{});
      if (!waited) rg_wait_vmcnt<NWL>();
      waited = false;
      landed_w(bq[par][1][0], bq[par][1][1], bq[par ^ 1][0][0], bq[par ^ 1][0][1]);
      if (ks_next >= 0) read_a(std::integral_constant<int, par ^ 1>{}, ks_next);
      __builtin_amdgcn_sched_barrier(0);
      block(std::integral_constant<int, 1>{});
      advance_q();
    };
    read_a(std::integral_constant<int, 0>{}, 0);
#pragma unroll 1
    for (int c = 0; c < 6; ++c) {
#pragma unroll
      for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = rg_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
      for (int ks = 0; ks < 8; ks += 2) {
        step(std::integral_constant<int, 0>{}, ks + 1);
        // (the last step of a chunk requests stage 0 again: the next chunk's first step)
        step(std::integral_constant<int, 1>{}, ks + 2 < 8 ? ks + 2 : (c + 1 < 6 ? 0 : -1));
      }
      if (c + 1 < 6) {
        rg_wait_vmcnt<2>();
        waited = true;
      }
      // the chunk's epilogue, per wave, through its private 16 x 36-float patches: q -> fp32 rows, k / v -> planes
      const int nw = c * 256 + wave * 32 + pc4;
      const float sc = c < 4 ? p.k_scale : p.v_scale;
      if (c < 2) {      // q: fp32 rows, 8 lanes x 16 B per row segment
        rg_f32x4 cw = *reinterpret_cast<const rg_f32x4*>(p.csq + nw);
        cw = cw * inv_q;
#pragma unroll
        for (int mt = 0; mt < RT; ++mt) {
          float* const ws = ws0 + (mt & (NPATCH - 1)) * (16 * 36);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) ws[(kq * 4 + e) * 36 + nt * 16 + r16] = acc[mt][nt][e];
#pragma unroll
          for (int ps = 0; ps < 2; ++ps) {
            const int trow = mt * 16 + ps * 8 + prow;
            const long mrow = (long)m0 + trow;
            const rg_f32x4 v = *reinterpret_cast<const rg_f32x4*>(ws + (ps * 8 + prow) * 36 + pc4) * cw;
            if (trow >= RO || mrow >= p.M) continue;
            *(__attribute__((address_space(1))) rg_f32x4*)(p.q + mrow * LDQ + nw) = v;
          }
        }
      } else {
        // k / v: fp16 planes, a lane takes EIGHT columns of a row (16-byte stores per plane: rowblock_kernel)
        const int prow16 = lane >> 2, pc8 = (lane & 3) * 8;
        const int nw8 = c * 256 + wave * 32 + pc8;
        rg_f32x4 cw0 = *reinterpret_cast<const rg_f32x4*>(p.csq + nw8), cw1 = *reinterpret_cast<const rg_f32x4*>(p.csq + nw8 + 4);
        cw0 = cw0 * (inv_q * sc);      // (powers of two: the same bits as scaling the product)
        cw1 = cw1 * (inv_q * sc);
#pragma unroll
        for (int mt = 0; mt < RT; ++mt) {
          float* const ws = ws0 + (mt & (NPATCH - 1)) * (16 * 36);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) ws[(kq * 4 + e) * 36 + nt * 16 + r16] = acc[mt][nt][e];
          const int trow = mt * 16 + prow16;
          const long mrow = (long)m0 + trow;
          const rg_f32x4 v0 = *reinterpret_cast<const rg_f32x4*>(ws + prow16 * 36 + pc8) * cw0;
          const rg_f32x4 v1 = *reinterpret_cast<const rg_f32x4*>(ws + prow16 * 36 + pc8 + 4) * cw1;
          if (trow >= RO || mrow >= p.M) continue;
          const Split2 s0 = split2h_pair(v0[0], v0[1]), s1 = split2h_pair(v0[2], v0[3]);
          const Split2 s2 = split2h_pair(v1[0], v1[1]), s3 = split2h_pair(v1[2], v1[3]);
          unsigned short* const o2 = p.kv2 + mrow * LDKV + (nw8 - 512);
          *(__attribute__((address_space(1))) rg_u32x4*)(o2) = rg_u32x4{s0.h, s1.h, s2.h, s3.h};
          *(__attribute__((address_space(1))) rg_u32x4*)(o2 + p.kv2_plane) = rg_u32x4{s0.l, s1.l, s2.l, s3.l};
        }
      }
    }
  }
  // the wrapped-around W loads of the last two steps: bq stays reserved until they have landed (rowgemm_wd_kernel)
  rg_wait_vmcnt<0>();
  landed_w(bq[0][0][0], bq[0][0][1], bq[0][1][0], bq[0][1][1]);
  landed_w(bq[1][0][0], bq[1][0][1], bq[1][1][0], bq[1][1][1]);
}

}  // namespace jv
