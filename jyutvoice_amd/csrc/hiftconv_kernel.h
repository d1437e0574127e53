// Row-owning fp16x3 convolution for the vocoder's ResBlocks (jyutvoice/hifigan/generator.py:90-97: xt = Snake(x) ->
// Conv1d(C, C, k, dilation d) -> Snake -> Conv1d(C, C, k) -> x + xt, kernel sizes 3 / 7 / 11, dilations 1 / 3 / 5, at
// C = 256, 128, 64 channels and 8 T, 40 T, 120 T + 1 rows per utterance): the trunk's treatment (rowconv_kernel.h,
// rowgemm_wa_kernel) for the 72 convolutions that are 85 % of the vocoder's arithmetic.
//
// The tile kernels (conv_gemm_x6_kernel.h) restage a 32-channel slice of the window for every K chunk -- Snake (a sin^2 per
// element), the fp16 split, an LDS store, two barriers -- and stream the weights through LDS as well: 0.19 - 0.30 of the
// fp16x3 ceiling, the 64-channel stage lowest (1.15 M rows of it).  Here one 8-wave workgroup owns 80 NG rows x all C output
// channels (NG = 256 / C row groups of 80 rows: every wave a 80 x 32 tile, 30 MFMAs per 32-deep step, as in the trunk):
//   * the WHOLE window -- 80 NG + (k - 1) d rows x C channels -- goes through Snake and the plane split ONCE, into LDS
//     (130 x 256, 210 x 128 or 370 x 64 values: 93 - 139 KB), and every tap of every chunk reads it in place at a row offset:
//     no staging, no DMA and no barrier inside the main loop;
//   * the weights arrive in fragment order through the register double buffer with hand-counted waits (rowgemm_wd_kernel);
//   * v_mfma_f32_16x16x32_f16; the epilogue (bias, residuals, scaling, accumulation, measured-bound tracking) runs per wave
//     through a private transposition patch, no workgroup barrier: one wave's stores run under the others' MFMAs.
// Arithmetic: the tile kernels' (same Snake evaluation, same per-utterance power-of-two scale from the measured bound, products
// hh' + hl' + lh' smallest first, fp32 accumulate over K ascending); 16x16x32 instead of 32x32x16 MFMAs sum a 32-deep step
// in another internal order, so results agree with the tile kernels to rounding.
#pragma once
#include "rowgemm_kernel.h"

namespace jv {

struct HiftConvArgs {
  const float* A;                  // fp32 row buffer [rows, C]; output row m, tap j reads row m + tap_row0 + j dil
  long a_rows;                     // rows of A that may be read (others read as zero)
  int M;                           // output rows
  int ntaps, dil, tap_row0;
  const unsigned char* rowmask_in; // per A row or null: 0 -> the row reads as zero
  const float* alpha;              // Snake: x + sin^2(alpha x) / (alpha + 1e-9), per input channel
  const unsigned short* Wf;        // fp16 planes of W[n][j C + ci] * 2^e_n in fragment order (pack_wfrag over K = ntaps C)
  long wf_plane;
  const float* colscale;           // 2^-e_n
  const float* bias;
  const float* amax_in;            // per-utterance measured bound of A; scale = h3_scale_dev(amax_in[slot] + a_extra)
  float a_extra;                   // what Snake can add to |x|: max 1 / (alpha + 1e-9)
  int slot_G, slot_S, slot_nb;     // slot(row) = clamp((row - slot_G) / slot_S, 0, slot_nb - 1)
  const int* slot_map;             // or, when set: slot(row) = slot_map[row] (the compact geometry of ragged batches: a table per level)
  float* out;                      // [rows, C]:  out = ((acc + bias) + res1 + res2) * out_scale (+ previous out)
  const float *res1, *res2;        // [rows, C] or null (either may alias out)
  float out_scale;
  int accumulate;
  float* amax_out;                 // tracking of what is stored, per utterance slot; rows with amax_mask == 0 are padding
  const unsigned char* amax_mask;
  long alg_rows;
};

__device__ __attribute__((aligned(64))) const float hc_zero_page[16] = {};
__device__ __attribute__((aligned(16))) const unsigned char hc_ones_page[16] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};

__device__ __forceinline__ void hc_atomic_max(float* slot, unsigned v) {      // integer max of the bit pattern of a non-negative float
  __hip_atomic_fetch_max((__attribute__((address_space(1))) unsigned*)slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int HC_RT = 5, HC_RG = 16 * HC_RT;      // rows per wave (row group)
// NG row groups of 80 rows per workgroup; a workgroup has NG * C / 32 waves (each a 80 x 32 tile)
template <int NG> constexpr int hc_rows() { return HC_RG * NG; }
template <int C, int NG> constexpr int hc_threads() { return 64 * NG * (C / 32); }
// window rows, padded: a multiple of 8 plus 1 (an ODD row count puts consecutive chunks 64 bytes apart modulo the 256-byte
// bank row instead of on top of each other -- the staging stores of a wave span two to eight chunks)
inline int hc_wrpad(int R, int ntaps, int dil) { return ((R + (ntaps - 1) * dil + 7) & ~7) + 1; }
template <int C, int NG> inline int hc_lds_bytes(int ntaps, int dil) {
  const int R = hc_rows<NG>(), wr = hc_wrpad(R, ntaps, dil);
  return ((wr * 4 + 255) & ~255) + R * 8 + (C / 32) * 2 * wr * 64 + NG * (C / 32) * 16 * 36 * 4;
}

template <int C, int NG>
__global__ __launch_bounds__(64 * NG * (C / 32), 2) void hiftconv_kernel(const HiftConvArgs p) {
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
  const int m0 = blockIdx.x * R;
  const int WR = R + (p.ntaps - 1) * p.dil;     // window rows really used
  const int WRP = ((WR + 7) & ~7) + 1;
  const int PS = WRP * 64, CS = 2 * PS;         // plane and chunk strides of the operand image
  const int KS = p.ntaps * NCH;                 // 32-deep steps (even: NCH is)
  float* const wscale = reinterpret_cast<float*>(hc_lds);                                   // [WRP]: < 0 = the row reads as zero
  int2* const rowinfo = reinterpret_cast<int2*>(hc_lds + ((WRP * 4 + 255) & ~255));         // [R]
  unsigned char* const img = reinterpret_cast<unsigned char*>(rowinfo) + R * 8;            // [NCH][2][WRP][64 B]
  float* const patch = reinterpret_cast<float*>(img + NCH * CS) + wave * (16 * 36);

  // ---- W: fragment order, [plane][KS][C / 16 blocks][64 lanes][8 halves] (k-step major); this wave's column blocks are 2 cp + nt ----
  const unsigned short* wp[2][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) wp[nt][pl] = p.Wf + (long)pl * p.wf_plane + (long)(2 * cp + nt) * 512 + lane * 8;
  int wk = 0;
  rg_u32x4 bq[2][2][2];
#pragma unroll
  for (int i = 0; i < 8; ++i) bq[i >> 2][(i >> 1) & 1][i & 1] = rg_u32x4{0u, 0u, 0u, 0u};
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

  // ---- per-row facts: the window rows' scales (one level of unconditional loads on clamped indices: rowconv_wd_kernel) ----
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
    // (WR <= R + 50 < NT: one window row and one output row per thread)
    const long ar = (long)m0 + p.tap_row0 + tid;
    const bool in_w = tid < WR && ar >= 0 && ar < p.a_rows;
    const long arc = ar < 0 ? 0 : (ar < p.a_rows ? ar : p.a_rows - 1);
    const int mk = ((gbytes)(p.rowmask_in ? p.rowmask_in : hc_ones_page))[p.rowmask_in ? arc : 0];
    const float am_w = p.amax_in[slot_of(arc)];
    const long mt = (long)m0 + (tid < R ? tid : 0);
    const long mc = mt < p.M ? mt : (long)p.M - 1;
    const int sl_o = slot_of(mc);
    const int trk = ((gbytes)(p.amax_mask ? p.amax_mask : hc_ones_page))[p.amax_mask ? mc : 0];
    const float am_o = p.amax_in[sl_o];
    if (tid < WRP) wscale[tid] = (in_w && mk != 0) ? h3_scale_dev(am_w + p.a_extra) : -1.f;
    if (tid < R) {
      const float inv = mt < p.M ? 1.0f / h3_scale_dev(am_o + p.a_extra) : 0.f;
      rowinfo[tid] = int2{(int)__float_as_uint(inv), sl_o | ((p.amax_out && mt < p.M && trk != 0) ? RI_TRACK : 0)};
    }
  }
  __syncthreads();

  // ---- the window, once: fp32 rows -> Snake -> x scale -> two fp16 planes, chunk-major operand image ----
  {
    const int c4 = tid % C4, r0 = tid / C4;      // this thread's four channels, its first window row
    const rg_f32x4 al = *reinterpret_cast<const rg_f32x4*>(p.alpha + 4 * c4);
    rg_f32x4 ai;
#pragma unroll
    for (int e = 0; e < 4; ++e) ai[e] = 1.0f / (al[e] + 1e-9f);
    const int chunk = c4 >> 3, cslot = (c4 & 7) >> 1, chalf = (c4 & 1) << 3;
    const float* const abase = p.A + ((long)m0 + p.tap_row0) * C + 4 * c4;
    // four rows in flight per thread and pass.  (All 12 - 17 of a thread's rows at once measured SLOWER on the same box: the
    // vocoder stage 32.2 ms against 24.6 -- the loop below is then one 8 000-line unrolled body, and two co-resident workgroups
    // hide each other's latency anyway.)
#ifndef JV_HC_U
#define JV_HC_U 4
#endif
    constexpr int U = JV_HC_U;
    for (int rb = r0; rb < WR; rb += U * RPI) {
      rg_f32x4 x[U];
      float sc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = rb + u * RPI;
        sc[u] = r < WR ? wscale[r] : -1.f;
        // (explicitly GLOBAL loads; a row that reads as zero is fetched from a page of zeros and never multiplied: guard rows
        // may hold anything)
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
  // the per-column constants of the epilogue, requested here so that their latency is not paid behind the main loop
  const int ncol = 32 * cp + 4 * (lane & 7);      // this lane's four columns in the row-wise passes
  rg_f32x4 cs4 = *reinterpret_cast<const rg_f32x4*>(p.colscale + ncol);
  rg_f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) b4 = *reinterpret_cast<const rg_f32x4*>(p.bias + ncol);
  rg_wait_vmcnt<0>();
  landed_w(bq[0][0][0], bq[0][0][1], bq[0][1][0], bq[0][1][1]);
  landed_w(bq[1][0][0], bq[1][0][1], bq[1][1][0], bq[1][1][1]);
  rg_lds_barrier();      // the operand image is complete: the only barrier ahead of the epilogue

  // ---- main loop: step ks = (tap j = ks / NCH, chunk c = ks % NCH); A fragments of row tile mt at window row
  // 80 grp + 16 mt + r16 + j dil (the 16-byte slot key depends on r16 + j dil alone) ----
  rg_f32x4 acc[RT][2];
#pragma unroll
  for (int mt = 0; mt < RT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = rg_f32x4{0.f, 0.f, 0.f, 0.f};
  rg_u32x4 af[2][RT][2];
  const unsigned char* const gimg = img + grp * RG * 64;
  auto read_a = [&](auto par_tag, const int c, const int j) {
    constexpr int par = decltype(par_tag)::value;
    const int lrow = r16 + j * p.dil;
    const unsigned char* const a = gimg + c * CS + lrow * 64 + ((kq ^ rg_key(lrow)) << 4);
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) af[par][mt][pl] = *reinterpret_cast<const rg_u32x4*>(a + pl * PS + mt * 1024);
  };
  int nc = 1, nj = 0;      // chunk and tap of the NEXT step
  read_a(std::integral_constant<int, 0>{}, 0, 0);
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
        mm(af[par][mt][1], bq[par][nt][0]);      // smallest terms first, as everywhere
        mm(af[par][mt][0], bq[par][nt][1]);
        mm(af[par][mt][0], bq[par][nt][0]);
        acc[mt][nt] = t;
      }
      __builtin_amdgcn_sched_barrier(0);
      load_w(par_tag, nttag);
      __builtin_amdgcn_sched_barrier(0);
    };
    // this wave's memory operations in program order: ... W0(s+1), W1(s+1) | W0(s+2), <wait>, W1(s+2) | ...; needed at the wait:
    // W0(s+1) and W1(s); behind W0(s+1): W1(s+1) and W0(s+2) = NWL loads (rowgemm_wa_kernel)
    block(std::integral_constant<int, 0>{});
    rg_wait_vmcnt<NWL>();
    landed_w(bq[par][1][0], bq[par][1][1], bq[par ^ 1][0][0], bq[par ^ 1][0][1]);
    if (!last) read_a(std::integral_constant<int, par ^ 1>{}, nc, nj);
    __builtin_amdgcn_sched_barrier(0);
    block(std::integral_constant<int, 1>{});
    advance_w();
    if (++nc == NCH) { nc = 0; ++nj; }
  };
#pragma unroll 1
  for (int ks = 0; ks < KS; ks += 2) {
    step(std::integral_constant<int, 0>{}, false);
    step(std::integral_constant<int, 1>{}, ks + 2 >= KS);
  }

  // ---- epilogue, per wave (no workgroup barrier): 16 rows at a time through the wave's private 16 x 36-float patch (MFMA
  // layout in: lane = column, 4 rows; rows out: 8 lanes x 16 B per row), so that every store writes whole 128-byte row segments
  {
    const int prow = lane >> 3;
    const float* const r1b = p.res1 ? p.res1 + ncol : nullptr;
    const float* const r2b = p.res2 ? p.res2 + ncol : nullptr;
    float* const ob = p.out + ncol;
    const int rowb = grp * RG;
    struct RowIn { rg_f32x4 r1[2], r2[2], pv[2]; float inv[2]; int info[2]; bool ok[2]; };
    auto request = [&](const int mt, RowIn& in) {      // the two 8-row halves of row tile mt
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const int trow = rowb + mt * 16 + ps * 8 + prow;
        const long m = (long)m0 + trow;
        in.ok[ps] = m < p.M;
        const long mc = in.ok[ps] ? m : (long)p.M - 1;
        const int2 ri = rowinfo[trow];
        in.inv[ps] = __uint_as_float((unsigned)ri.x);
        in.info[ps] = ri.y;
        const rg_f32x4 z = {0.f, 0.f, 0.f, 0.f};
        in.r1[ps] = r1b ? *(const __attribute__((address_space(1))) rg_f32x4*)(r1b + mc * C) : z;
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
        // slot already holds a larger value (a plain, cacheable read of it: a slot only grows).  Rows of two utterances: per lane.
        const int s_lo = rowinfo[rowb + mt * 16].y & (RI_TRACK - 1), s_hi = rowinfo[rowb + mt * 16 + 15].y & (RI_TRACK - 1);
        if (s_lo == s_hi) {
          // (explicitly GLOBAL, like every load and store here: a flat access counts on lgkmcnt too and would sit in every LDS wait)
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
}

}  // namespace jv
