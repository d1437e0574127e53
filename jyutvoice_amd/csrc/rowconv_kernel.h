// Row-owning fp16x3 causal convolution (k = 3, 256 output channels) for the estimator's trunk: the two CausalBlock1D
// convolutions of every CausalResnetBlock1D and the down / up / final causal convolutions (jyutvoice/flow/decoder.py:110-115,
// 767-788, 976-1012), with what follows each of them row-wise in its epilogue: LayerNorm over the 256 channels -> Mish ->
// mask -> + time embedding -> + residual (the former ln_epilogue_rows pass), and the measured-bound tracking.
//
// Same skeleton as rowgemm_kernel.h (one 8-wave workgroup per CU owns 16 RT whole rows x all 256 columns; weights by LDS-DMA
// into a 3-stage ring with counted waits and one barrier per step; v_mfma_f32_16x16x32_f16), with two differences:
//   * a step is (32-channel chunk c, tap j): the weight tile of a step is W[:, j Cin + 32 c .. + 32]; the A operand of the
//     three taps of a chunk is ONE window of 16 RT + 2 rows, staged once per chunk and read at row offsets 0, 1, 2 (the
//     slot key of rowgemm_kernel.h is conflict-free at every row offset);
//   * A is the residual stream itself -- fp32 rows whose bound is MEASURED, per utterance (ConvGemmArgs::amax_in) -- so it is
//     split here: 16 RT + 2 rows x 32 channels per chunk over 512 threads is two float4 per thread, against a 64x64 tile
//     kernel that split the same window once per column tile (4x) -- the vector work per MFMA drops 4x, the barriers 3x.
#pragma once
#include "rowgemm_kernel.h"

namespace jv {

struct RowConvArgs {
  const float* A;      // fp32 row buffer; output row m reads rows m - 2, m - 1, m (rows outside [0, a_rows) or masked: zero)
  long lda, a_rows;
  int M, Cin;          // Cin % 32 == 0
  const unsigned char* rowmask_in;
  const unsigned short* W2;      // fp16 planes [2][256][ldw] of W[n][j Cin + ci] * 2^e_n, colscale[n] = 2^-e_n
  long w2_plane;
  int ldw;
  const unsigned short* Wf;      // the same planes in MFMA-fragment order (pack_wfrag over K = 3 Cin; rowconv_wd_kernel); null: W2 by LDS-DMA
  long wf_plane;
  const float* colscale;
  const float* amax_in;          // per-utterance bound of A (slot = row_slot[row])
  const int* row_slot;
  int slot_G, slot_S, slot_nb;   // (slot_S < 0: no arithmetic, row_slot is read -- the compact geometry of ragged batches) the same slots by arithmetic, clamp((row - slot_G) / slot_S, 0, slot_nb - 1) (slot_S = 0: slot 0):
                                 // rowconv_wd_kernel computes them instead of loading row_slot (one level of dependent loads less)
  const float* bias;
  float* out;
  long ldo;
  int ln;                        // LayerNorm over the 256 columns, then act, in the epilogue
  const float *ln_g, *ln_b;
  float ln_eps;
  int act;
  const unsigned char* rowmask_out;      // 0 -> value := 0 (before the additions below)
  const float* rowvec;                   // + rowvec[row_slot[m] * rowvec_ld + n]
  int rowvec_ld;
  const float* res;                      // + res[m * ldr + n]
  long ldr;
  float* amax_out;                       // tracking of what is stored (rows with row_mask == 0 excluded)
  const unsigned char* row_mask;
  // rowconv_wd_kernel: LayerNorm_256 of the STORED row (gain ln2_g, offset ln2_b, ln_eps), times ln2_scale, as fp16 planes
  // [2][rows][256] at plane stride ln2_plane -- the norm1 of the transformer block that follows a resnet's second
  // convolution (transformer.py:355-364), which was a launch of its own (layernorm256_planes) per stage
  unsigned short* ln2_out;
  long ln2_plane;
  const float *ln2_g, *ln2_b;
  float ln2_scale;
  // rowconv_wd_kernel<RT, true>: the resnet's 1 x 1 res_conv (decoder.py:110-115: output = block2(..) + res_conv(x * mask)) computed
  // by block1's launch, which stages the very rows res_conv reads: Wf then holds FOUR fragment steps per 32-channel chunk --
  // the three taps and, as a fourth, res_conv's chunk -- the fourth step multiplies the window at the row offset of tap 2 (the
  // output row itself) into a second accumulator, and a second slab pass stores res_out = acc_r * res_cs / a_scale + res_bias
  // (fp32 rows [., 256]) for block2's launch to add.  Replaces a tile-kernel launch per resnet and its re-read of the input.
  float* res_out;
  const float *res_cs, *res_bias;
  long alg_rows;
  int ablate;      // tuning aid (JV_RG_ABLATE, tuning builds): 1 no weight DMA in the loop, 2 no LDS reads + MFMAs, 4 no waits / barriers,
                   // 8 no A staging in the loop, 16 no epilogue
  unsigned long long* stamps;      // tuning aid (JV_RB_STAMPS, tuning builds; rowconv_wd_kernel): [workgroup][8] last-wave s_memtime per phase
};

__device__ __attribute__((aligned(64))) const float rc_zero_page[16] = {};
// what an absent (null) row mask reads as: the loads stay unconditional -- behind `mask ? mask[row] : 1` each became a branch
// with its own s_waitcnt vmcnt(0) at the join
__device__ __attribute__((aligned(16))) const unsigned char rc_ones_page[16] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};

template <int RT> constexpr int rc_a_plane() { return (16 * RT + 16) * 64; }
template <int RT> constexpr int rc_lds_bytes() { return 3 * 32768 + 2 * 2 * rc_a_plane<RT>() + 16 * RT * 8; }
static_assert(16 * 5 * RG_SLD * 4 <= 3 * 32768, "the epilogue slab of the tallest tile fits the weight ring");

template <int RT>
__global__ __launch_bounds__(512, 2) void rowconv_kernel(const RowConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rc_lds[];
  constexpr int R = 16 * RT, WR = R + 2;
  constexpr int WSTAGE = 32768, W_PLANE = 256 * 64;
  constexpr int A_PLANE = rc_a_plane<RT>(), A_BUF = 2 * A_PLANE, A_OFF = 3 * WSTAGE;
  constexpr int NI = (WR * 8 + 511) / 512;      // float4 per thread per chunk (window rows x 8 float4)
  constexpr int PPW = 4;                        // weight pieces per wave and step: 2 planes x 16 groups / 8 waves
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.x * R;
  const int NCH = p.Cin >> 5;
  const int total = 3 * NCH;

  // warm this XCD's L2 with the weight planes (see rowgemm_kernel.h)
  float warm = 0.f;
  {
    const long lpp = ((long)256 * p.ldw * 2) >> 7;
    const int grp = blockIdx.x >> 3, ngrp = (gridDim.x + 7) >> 3;
    const long per = (2 * lpp + ngrp - 1) / ngrp;
    const long l = (long)grp * per + tid;
    if (tid < per && l < 2 * lpp) {
      const int pl = l >= lpp;
      warm = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.W2 + (long)pl * p.w2_plane) + ((l - pl * lpp) << 7));
    }
  }

  // ---- weight pieces of this wave: piece pc = wave + 8 i -> plane pc >> 4, 16-column group pc & 15
  const unsigned short* cur[PPW];
  int dst[PPW];
  {
    const int prow = lane >> 2, pslot = (lane & 3) ^ rg_key(prow);
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int pc = wave + 8 * i, pl = pc >> 4, g = pc & 15;
      cur[i] = p.W2 + (long)pl * p.w2_plane + (long)(g * 16 + prow) * p.ldw + 8 * pslot;
      dst[i] = pl * W_PLANE + g * 1024;
    }
  }
  int ij = 0;      // tap of the next step to issue
  auto issue_piece = [&](auto itag, int stage) {
    constexpr int i = decltype(itag)::value;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)cur[i],
                                     (__attribute__((address_space(3))) void*)(rc_lds + stage * WSTAGE + dst[i]), 16, 0, 0);
  };
  auto advance = [&]() {      // k offset of step (c, j) is j Cin + 32 c
    const bool wrap = ++ij == 3;
    if (wrap) ij = 0;
    const int d = wrap ? 32 - 2 * p.Cin : p.Cin;
#pragma unroll
    for (int i = 0; i < PPW; ++i) cur[i] += d;
  };
  auto issue_all = [&](int stage) {
    issue_piece(std::integral_constant<int, 0>{}, stage);
    issue_piece(std::integral_constant<int, 1>{}, stage);
    issue_piece(std::integral_constant<int, 2>{}, stage);
    issue_piece(std::integral_constant<int, 3>{}, stage);
    advance();
  };

  // The weight ring starts first: everything below (row slots -> measured bounds -> first A window) is a chain of dependent
  // global loads, several microseconds that the first two weight tiles spend in flight
  issue_all(0);
  issue_all(1);      // total >= 3

  // ---- per-row facts of this tile, read once here (two dependent loads per row) and kept in LDS for the epilogue:
  // x = bits of 1 / a_scale of the row's utterance, y = its slot | flags
  int2* const rowinfo = reinterpret_cast<int2*>(rc_lds + A_OFF + 2 * A_BUF);
  constexpr int RI_KEEP = 1 << 29, RI_TRACK = 1 << 30;
  if (tid < R) {
    const long m = (long)m0 + tid;
    int y = 0;
    float inv = 0.f;
    if (m < p.M) {
      const int sl = p.row_slot[m];
      inv = 1.0f / h3_scale_dev(p.amax_in[sl]);
      y = sl;
      if (!p.rowmask_out || p.rowmask_out[m] != 0) y |= RI_KEEP;
      if (p.amax_out && (!p.row_mask || p.row_mask[m] != 0)) y |= RI_TRACK;
    }
    rowinfo[tid] = int2{(int)__float_as_uint(inv), y};
  }

  // ---- A window staging: thread -> (window row, float4) pairs, fixed over the chunks
  const float* asrc[NI];
  int astep[NI], adst[NI];
  float ascale[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int idx = tid + 512 * i;
    const int r = idx >> 3, c4 = idx & 7;
    const long ar = (long)m0 - 2 + r;
    bool ok = r < WR && ar >= 0 && ar < p.a_rows;
    if (ok && p.rowmask_in) ok = p.rowmask_in[ar] != 0;
    asrc[i] = ok ? p.A + ar * p.lda + 4 * c4 : rc_zero_page;
    astep[i] = ok ? 1 : 0;
    ascale[i] = ok ? h3_scale_dev(p.amax_in[p.row_slot[ar]]) : 0.f;
    adst[i] = r < WR ? r * 64 + ((((c4 >> 1) ^ rg_key(r)) << 4) | ((c4 & 1) << 3)) : -1;
  }
  rg_f32x4 pa[NI];
  auto load_A = [&](int c) {
#pragma unroll
    for (int i = 0; i < NI; ++i)      // explicitly GLOBAL loads: a generic pointer (A or the zero page) would make them flat_load, which
                                      // counts on lgkmcnt as well and returns out of order -- no counted wait is valid beside it
      pa[i] = *(const __attribute__((address_space(1))) rg_f32x4*)(asrc[i] + c * 32 * astep[i]);
  };
  auto store_A = [&](int buf) {
    unsigned char* const base = rc_lds + A_OFF + buf * A_BUF;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (adst[i] >= 0) {
        const Split2 s0 = split2h_pair(pa[i][0] * ascale[i], pa[i][1] * ascale[i]);
        const Split2 s1 = split2h_pair(pa[i][2] * ascale[i], pa[i][3] * ascale[i]);
        *reinterpret_cast<rg_u32x2*>(base + adst[i]) = rg_u32x2{s0.h, s1.h};
        *reinterpret_cast<rg_u32x2*>(base + A_PLANE + adst[i]) = rg_u32x2{s0.l, s1.l};
      }
    }
  };

  const int w_off = (wave * 32 + r16) * 64 + ((kq ^ rg_key(r16)) << 4);      // + pl * W_PLANE + nt * 1024

  rg_f32x4 acc[RT][2];
#pragma unroll
  for (int mt = 0; mt < RT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = rg_f32x4{0.f, 0.f, 0.f, 0.f};

  // the epilogue's per-column constants, fetched here so that their latency is not paid behind the main loop
  const rg_f32x4 cs4 = *reinterpret_cast<const rg_f32x4*>(p.colscale + 4 * lane);
  rg_f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) b4 = *reinterpret_cast<const rg_f32x4*>(p.bias + 4 * lane);
  rg_f32x4 gg = {1.f, 1.f, 1.f, 1.f}, bb = {0.f, 0.f, 0.f, 0.f};
  if (p.ln) {
    gg = *reinterpret_cast<const rg_f32x4*>(p.ln_g + 4 * lane);
    bb = *reinterpret_cast<const rg_f32x4*>(p.ln_b + 4 * lane);
  }
  load_A(0);
  store_A(0);
  if (NCH > 1) load_A(1);
  // vector-memory queue of this wave from here on, oldest first: W(0) W(1) (both waited for with A(0)) [A loads of chunk 1]
  int s = 0;
  for (int c = 0; c < NCH; ++c) {
#pragma unroll
    for (int j = 0; j < 3; ++j, ++s) {
      // W(s) has landed once at most the operations issued after it are outstanding: W(s + 1), and the A loads of chunk
      // c + 2 when they were issued behind W(s + 1) or W(s + 2) (the end of step (c, 0))
      if (!JV_ABLATE(p, 4)) {
        if (s + 1 >= total) rg_wait_vmcnt<0>();
        else if (j != 0 && c + 2 < NCH && !JV_ABLATE(p, 8)) rg_wait_vmcnt<PPW + NI>();
        else rg_wait_vmcnt<PPW>();
        rg_lds_barrier();      // (+ this wave's A-plane stores of the previous chunk step are complete)
      }
      const bool more = s + 2 < total && !JV_ABLATE(p, 1);
      const int nstage = (s + 2) % 3;
      if (j == 0 && c + 1 < NCH && !JV_ABLATE(p, 8)) store_A((c + 1) & 1);      // its registers were loaded a chunk ago; buffer last read in chunk c - 1
      if (JV_ABLATE(p, 2)) {
        if (more) issue_all(nstage);
        if (j == 0 && c + 2 < NCH && !JV_ABLATE(p, 8)) load_A(c + 2);
        continue;
      }
      const unsigned char* const sw = rc_lds + (s % 3) * WSTAGE;
      const unsigned char* const sa = rc_lds + A_OFF + (c & 1) * A_BUF;
      rg_u32x4 b[2][2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) b[nt][pl] = *reinterpret_cast<const rg_u32x4*>(sw + w_off + pl * W_PLANE + nt * 1024);
      auto group = [&](auto mtag) {
        constexpr int mt = decltype(mtag)::value;
        const int row = mt * 16 + r16 + j;
        const int a_off = row * 64 + ((kq ^ rg_key(row)) << 4);
        rg_u32x4 a[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) a[pl] = *reinterpret_cast<const rg_u32x4*>(sa + pl * A_PLANE + a_off);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          rg_f32x4 t = acc[mt][nt];
          auto mm = [&](const rg_u32x4& x, const rg_u32x4& y) {
            t = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(rg_f16x8, x), __builtin_bit_cast(rg_f16x8, y), t, 0, 0, 0);
          };
          mm(a[1], b[nt][0]);
          mm(a[0], b[nt][1]);
          mm(a[0], b[nt][0]);
          acc[mt][nt] = t;
        }
        if (more) {
          if constexpr (mt < PPW) issue_piece(std::integral_constant<int, (mt < PPW ? mt : 0)>{}, nstage);
          if constexpr (mt == RT - 1 && RT < PPW) {      // fewer groups than pieces: the rest behind the last group
            if constexpr (RT <= 1) issue_piece(std::integral_constant<int, 1>{}, nstage);
            if constexpr (RT <= 2) issue_piece(std::integral_constant<int, 2>{}, nstage);
            if constexpr (RT <= 3) issue_piece(std::integral_constant<int, 3>{}, nstage);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      group(std::integral_constant<int, 0>{});
      if constexpr (RT > 1) group(std::integral_constant<int, 1>{});
      if constexpr (RT > 2) group(std::integral_constant<int, 2>{});
      if constexpr (RT > 3) group(std::integral_constant<int, 3>{});
      if constexpr (RT > 4) group(std::integral_constant<int, 4>{});
      if (more) advance();
      if (j == 0 && c + 2 < NCH && !JV_ABLATE(p, 8)) load_A(c + 2);      // behind this step's weight pieces (the wait counts above rely on it)
    }
  }

  // ---- epilogue: the whole tile through ONE slab laid over the (now idle) weight ring -- a single chunk has nothing left to
  // prefetch -- then 2 RT rows per wave, RT at a time so that their loads, reductions and transcendentals overlap.  (In 32-row
  // passes through a slab of their own, two barriers each, the tail took 13 - 19 us of a 50 us launch.)
  if (JV_ABLATE(p, 16)) return;
  const bool mish = p.act == ACT_MISH;      // (uniform) the one activation the estimator uses here
  rg_lds_barrier();      // every wave is done reading the ring
  float* const slab = reinterpret_cast<float*>(rc_lds);
#pragma unroll
  for (int mt = 0; mt < RT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e) slab[(mt * 16 + kq * 4 + e) * RG_SLD + wave * 32 + nt * 16 + r16] = acc[mt][nt][e];
  rg_lds_barrier();
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    constexpr int RPW = RT;      // rows per wave and group: the wave owns tile rows 2 RT wave ... 2 RT wave + 2 RT - 1
    rg_f32x4 v[RPW], r[RPW], rv[RPW];
    long mrow[RPW];
    bool ok[RPW], keep[RPW], tracked[RPW];
    unsigned seen[RPW];
    int slot[RPW];
#pragma unroll
    for (int jj = 0; jj < RPW; ++jj) {
      const int trow = wave * 2 * RT + ps * RT + jj;
      mrow[jj] = (long)m0 + trow;
      ok[jj] = mrow[jj] < p.M;
      const int2 ri = rowinfo[trow];
      slot[jj] = ri.y & (RI_KEEP - 1);
      const float inv = ok[jj] ? __uint_as_float((unsigned)ri.x) : 0.f;      // 1 / the power of two its window rows were staged with
      v[jj] = *reinterpret_cast<const rg_f32x4*>(slab + trow * RG_SLD + 4 * lane) * (cs4 * inv) + b4;
      keep[jj] = ok[jj] && (ri.y & RI_KEEP) != 0;
      r[jj] = (p.res && ok[jj]) ? *reinterpret_cast<const rg_f32x4*>(p.res + mrow[jj] * p.ldr + 4 * lane) : rg_f32x4{0.f, 0.f, 0.f, 0.f};
      rv[jj] = (p.rowvec && ok[jj]) ? *reinterpret_cast<const rg_f32x4*>(p.rowvec + (long)slot[jj] * p.rowvec_ld + 4 * lane)
                                    : rg_f32x4{0.f, 0.f, 0.f, 0.f};
      tracked[jj] = ok[jj] && (ri.y & RI_TRACK) != 0;
      seen[jj] = 0xffffffffu;
      if (tracked[jj]) seen[jj] = *reinterpret_cast<const unsigned*>(p.amax_out + slot[jj]);
    }
    if (p.ln) {
      float sum[RPW], sq[RPW];
#pragma unroll
      for (int jj = 0; jj < RPW; ++jj) sum[jj] = wave_sum((v[jj][0] + v[jj][1]) + (v[jj][2] + v[jj][3]));
#pragma unroll
      for (int jj = 0; jj < RPW; ++jj) {
        const rg_f32x4 d = v[jj] - sum[jj] * (1.f / 256.f);
        sq[jj] = wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]));
      }
#pragma unroll
      for (int jj = 0; jj < RPW; ++jj) {
        const float mean = sum[jj] * (1.f / 256.f);
        const float rstd = 1.0f / sqrtf(sq[jj] * (1.f / 256.f) + p.ln_eps);
        v[jj] = (v[jj] - mean) * rstd * gg + bb;
      }
    }
#pragma unroll
    for (int jj = 0; jj < RPW; ++jj) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[jj][e] = keep[jj] ? (mish ? mish_fast(v[jj][e]) : act_apply(v[jj][e], p.act)) : 0.f;
      v[jj] = (v[jj] + rv[jj]) + r[jj];
      if (ok[jj]) *reinterpret_cast<rg_f32x4*>(p.out + mrow[jj] * p.ldo + 4 * lane) = v[jj];
    }
    if (p.amax_out) {
#pragma unroll
      for (int jj = 0; jj < RPW; ++jj) {
        unsigned u = 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) u = max(u, __float_as_uint(v[jj][e]) & 0x7fffffffu);
        if (tracked[jj] && __builtin_amdgcn_ballot_w64(u > seen[jj]) != 0) {
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) u = max(u, (unsigned)__shfl_xor((int)u, o));
          if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(p.amax_out + slot[jj]), u);
        }
      }
    }
  }
  asm volatile("" ::"v"(warm));
}

// ---- W-direct form (see rowgemm_wd_kernel): weights in fragment order through a register double buffer, LDS holds the two
// A window buffers, the row facts and a slab of its own; one workgroup barrier per 32-channel chunk instead of one per step
template <int RT> constexpr int rcw_slab_off() { return (2 * 2 * rc_a_plane<RT>() + 16 * RT * 8 + 255) & ~255; }
template <int RT> constexpr int rcw_lds_bytes() { return rcw_slab_off<RT>() + 16 * RT * RG_SLD * 4; }

template <int RT, bool RES>
__global__ __launch_bounds__(512, 2) void rowconv_wd_kernel(const RowConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rc_lds[];
  constexpr int R = 16 * RT, WR = R + 2;
  constexpr int A_PLANE = rc_a_plane<RT>(), A_BUF = 2 * A_PLANE, A_OFF = 0;
  constexpr int NI = (WR * 8 + 511) / 512;      // float4 per thread per chunk (window rows x 8 float4)
  constexpr int NWL = 4;                        // weight loads per wave and step: 2 column blocks x 2 planes
  constexpr int SLAB_OFF = rcw_slab_off<RT>();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.x * R;
  const int NCH = p.Cin >> 5;
  constexpr int NJ = RES ? 4 : 3;      // fragment steps per chunk: the taps (+ res_conv's chunk)
  int stamp_i = 0;
  auto stamp = [&]() {      // (compiled out unless JV_TUNING)
    if (JV_STAMP(p)) {
      if (lane == 0 && stamp_i < 8) atomicMax(&p.stamps[(long)blockIdx.x * 8 + stamp_i], (unsigned long long)__builtin_amdgcn_s_memtime());
      ++stamp_i;
    }
  };
  stamp();      // 0: start

  // warm this XCD's L2 with the weight planes (see rowgemm_kernel.h)
  float warm = 0.f;
  {
    const long lpp = ((long)256 * NJ * p.Cin * 2) >> 7;
    const int grp = blockIdx.x >> 3, ngrp = (gridDim.x + 7) >> 3;
    const long per = (2 * lpp + ngrp - 1) / ngrp;
    const long l = (long)grp * per + tid;
    if (tid < per && l < 2 * lpp) {
      const int pl = l >= lpp;
      warm = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.Wf + (long)pl * p.wf_plane) + ((l - pl * lpp) << 7));
    }
  }

  // ---- W: fragment order over K = NJ Cin (k = j Cin + 32 c is fragment step j NCH + c), a register double buffer loaded by
  // inline asm with counted waits -- rowgemm_wd_kernel's, including its rules (tools/check_rowgemm_isa.py covers this kernel)
  const unsigned short* wbase[2][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) wbase[nt][pl] = p.Wf + (long)pl * p.wf_plane + (long)(wave * 2 + nt) * 512 + lane * 8;
  long woff = 0;       // halves: fragment step of the NEXT step to load, x 16 column blocks x 512 (k-step major: pack_wfrag)
  int wj = 0, wc = 0;  // its tap and chunk
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
  auto advance_w = [&]() {      // (c, j) -> (c, j + 1): + NCH steps; (c, NJ - 1) -> (c + 1, 0): + 1 - (NJ - 1) NCH; past the end: back to 0
    if (++wj == NJ) {
      wj = 0;
      if (++wc == NCH) { wc = 0; woff = 0; }
      else woff += (1L - (long)(NJ - 1) * NCH) * (16 * 512);
    } else {
      woff += (long)NCH * (16 * 512);
    }
  };
  // The weights start first: everything below (row slots -> measured bounds -> first A window) is a chain of dependent
  // global loads, several microseconds that the first two steps' fragments spend in flight
  load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
  advance_w();
  load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
  load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
  advance_w();

  // ---- per-row facts of this tile (kept in LDS for the epilogue: x = bits of 1 / a_scale of the row's utterance, y = its slot |
  // flags) and the staging plan of this thread's window rows.  ONE level of loads: the utterance slot of a row is computed
  // (row geometry: slot_G / slot_S), not loaded; masks, measured bounds and the first window rows themselves are all requested
  // unconditionally on clamped indices, and a masked row is zeroed by selection when it is staged.  (As a chain row -> mask ->
  // slot -> bound -> rows, each level behind its own s_waitcnt vmcnt(0) -- which, vmcnt being in order, also sat out the cold
  // weight fragments requested first -- the setup took 10 k cycles of a 64 k-cycle launch in the phase stamps.)
  int2* const rowinfo = reinterpret_cast<int2*>(rc_lds + A_OFF + 2 * A_BUF);      // 8 R bytes, then the slab
  constexpr int RI_KEEP = 1 << 29, RI_TRACK = 1 << 30;
  auto slot_of = [&](const long row) -> int {
    if (p.slot_S < 0) {      // compact geometry (ragged batches): by table; `row` is clamped to the buffer by every caller
      const int q = p.row_slot[row];
      return q < 0 ? 0 : (q >= p.slot_nb ? p.slot_nb - 1 : q);
    }
    if (p.slot_S <= 0) return 0;
    const int q = (int)((row - p.slot_G) / p.slot_S);
    return q < 0 ? 0 : (q >= p.slot_nb ? p.slot_nb - 1 : q);
  };
  const float* asrc[NI];
  int adst[NI];
  float ascale[NI];      // 0: the row reads as zero (outside the buffer, or masked)
  bool ok[NI];
  long arc[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int idx = tid + 512 * i;
    const int r = idx >> 3, c4 = idx & 7;
    const long ar = (long)m0 - 2 + r;
    ok[i] = r < WR && ar >= 0 && ar < p.a_rows;
    arc[i] = ar < 0 ? 0 : (ar < p.a_rows ? ar : p.a_rows - 1);
    asrc[i] = p.A + arc[i] * p.lda + 4 * c4;
    adst[i] = r < WR ? r * 64 + ((((c4 >> 1) ^ rg_key(r)) << 4) | ((c4 & 1) << 3)) : -1;
  }
  // The window rows of the next chunk travel in registers for a chunk's worth of steps.  Loaded by asm like the weight
  // fragments: as plain loads the compiler put its own s_waitcnt vmcnt(0) in front of store_A -- it cannot see the weight
  // loads issued since, so once per chunk that wait drained the fragments in flight (stamps: 1340 cycles per step against
  // the 900 of the same loop in rowblock_kernel.h).  The steps' counted waits already cover these loads (they are older
  // than the newest NWL fragment loads at the wait of step (c, 2)); landed_a() tells the compiler so.
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
    unsigned char* const base = rc_lds + A_OFF + buf * A_BUF;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (adst[i] >= 0) {
        const bool use = ascale[i] != 0.f;      // (a row that reads as zero may hold anything, NaN included: selected, never multiplied)
        const Split2 s0 = split2h_pair(use ? pa[i][0] * ascale[i] : 0.f, use ? pa[i][1] * ascale[i] : 0.f);
        const Split2 s1 = split2h_pair(use ? pa[i][2] * ascale[i] : 0.f, use ? pa[i][3] * ascale[i] : 0.f);
        *reinterpret_cast<rg_u32x2*>(base + adst[i]) = rg_u32x2{s0.h, s1.h};
        *reinterpret_cast<rg_u32x2*>(base + A_PLANE + adst[i]) = rg_u32x2{s0.l, s1.l};
      }
    }
  };
  load_A(0);      // the first window's rows travel with the facts below
  {
    typedef const __attribute__((address_space(1))) unsigned char* gbytes;
    const long mt = (long)m0 + (tid < R ? tid : 0);
    const long mc = mt < p.M ? mt : (long)p.M - 1;
    const int sl_o = slot_of(mc);
    const int keep_o = ((gbytes)(p.rowmask_out ? p.rowmask_out : rc_ones_page))[p.rowmask_out ? mc : 0];
    const int trk_o = ((gbytes)(p.row_mask ? p.row_mask : rc_ones_page))[p.row_mask ? mc : 0];
    const float am_o = p.amax_in[sl_o];
    int mk_i[NI];
    float am_i[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      mk_i[i] = ((gbytes)(p.rowmask_in ? p.rowmask_in : rc_ones_page))[p.rowmask_in ? arc[i] : 0];
      am_i[i] = p.amax_in[slot_of(arc[i])];
    }
    if (tid < R) {
      int y = 0;
      float inv = 0.f;
      if (mt < p.M) {
        inv = 1.0f / h3_scale_dev(am_o);
        y = sl_o;
        if (keep_o != 0) y |= RI_KEEP;
        if (p.amax_out && trk_o != 0) y |= RI_TRACK;
      }
      rowinfo[tid] = int2{(int)__float_as_uint(inv), y};
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) ascale[i] = (ok[i] && mk_i[i] != 0) ? h3_scale_dev(am_i[i]) : 0.f;
  }

  rg_f32x4 acc[RT][2], accr[RES ? RT : 1][2];      // accr: res_conv's product (RES)
#pragma unroll
  for (int mt = 0; mt < RT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      acc[mt][nt] = rg_f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (RES) accr[mt][nt] = rg_f32x4{0.f, 0.f, 0.f, 0.f};
    }

  // the epilogue's per-column constants, fetched here so that their latency is not paid behind the main loop
  const rg_f32x4 cs4 = *reinterpret_cast<const rg_f32x4*>(p.colscale + 4 * lane);
  rg_f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) b4 = *reinterpret_cast<const rg_f32x4*>(p.bias + 4 * lane);
  rg_f32x4 gg = {1.f, 1.f, 1.f, 1.f}, bb = {0.f, 0.f, 0.f, 0.f};
  if (p.ln) {
    gg = *reinterpret_cast<const rg_f32x4*>(p.ln_g + 4 * lane);
    bb = *reinterpret_cast<const rg_f32x4*>(p.ln_b + 4 * lane);
  }
  rg_f32x4 csr4 = {0.f, 0.f, 0.f, 0.f}, br4 = {0.f, 0.f, 0.f, 0.f};
  if constexpr (RES) {
    csr4 = *reinterpret_cast<const rg_f32x4*>(p.res_cs + 4 * lane);
    if (p.res_bias) br4 = *reinterpret_cast<const rg_f32x4*>(p.res_bias + 4 * lane);
  }
  stamp();      // 1: row facts and staging addresses set up
  rg_wait_vmcnt<0>();      // the first window's rows (and the first fragments, needed two lines down anyway)
  landed_a();
  store_A(0);
  if (NCH > 1) load_A(1);
  // both steps' fragments (and everything older) have landed once only the A loads of chunk 1 are outstanding
  if (NCH > 1) rg_wait_vmcnt<NI>(); else rg_wait_vmcnt<0>();
  landed_w(bq[0][0][0], bq[0][0][1], bq[0][1][0], bq[0][1][1]);
  landed_w(bq[1][0][0], bq[1][0][1], bq[1][1][0], bq[1][1][1]);
  stamp();      // 2: first window staged, first fragments landed

  // One step = (chunk c, tap j): block 0 (column block 0 x all row groups, A fragments at row offset j of the chunk's window),
  // reload bq[par][0] for step s + 2, counted wait, block 1, reload bq[par][1].  This wave's vector-memory operations in
  // program order: ... W0(s+1), W1(s+1) [A loads of chunk c + 2, behind step (c, 0)] | W0(s+2), <wait>, W1(s+2) ...; needed at
  // the wait: W0(s+1) and W1(s); behind W0(s+1): W1(s+1), the A loads where step s - 1 issued them, W0(s+2).  The window
  // changes per CHUNK, not per step: one workgroup barrier per chunk (three in the LDS-ring form, whose weight stages
  // needed one per step).  Past the end of the launch the W loads wrap around to weights that exist.
  auto step = [&](auto par_tag, const int c, auto jtag) {
    constexpr int par = decltype(par_tag)::value, j = decltype(jtag)::value;
    constexpr int jrow = j < 3 ? j : 2;      // res_conv's step reads the output row itself: tap 2's row offset
    if (j == 0) {
      rg_lds_barrier();      // every thread's plane stores of this chunk's window are complete; the other buffer is free
      if (c + 1 < NCH) {      // its registers were loaded a chunk ago and waited for at step (c - 1, 2) ...
        if (c == 0) rg_wait_vmcnt<0>();      // ... except chunk 1's, issued just ahead of the loop (nothing else is in flight yet)
        landed_a();
        store_A((c + 1) & 1);
      }
    }
    const unsigned char* const sa = rc_lds + A_OFF + (c & 1) * A_BUF;
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
    if constexpr (j == 3) blocks(accr); else blocks(acc);      // (res_conv's step: the second accumulator)
    advance_w();
    if (j == 0 && c + 2 < NCH) load_A(c + 2);      // behind this step's weight loads (the wait counts above rely on it)
  };
  typedef std::integral_constant<int, 0> J0;
  typedef std::integral_constant<int, 1> J1;
  typedef std::integral_constant<int, 2> J2;
  typedef std::integral_constant<int, 3> J3;
  if constexpr (RES) {
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {      // four steps per chunk: parities 0 1 0 1
      step(std::integral_constant<int, 0>{}, c, J0{});
      step(std::integral_constant<int, 1>{}, c, J1{});
      step(std::integral_constant<int, 0>{}, c, J2{});
      step(std::integral_constant<int, 1>{}, c, J3{});
    }
  } else {
#pragma unroll 1
    for (int c = 0; c < NCH; c += 2) {      // NCH is even: two chunks = six steps, parities 0 1 0 1 0 1
      step(std::integral_constant<int, 0>{}, c, J0{});
      step(std::integral_constant<int, 1>{}, c, J1{});
      step(std::integral_constant<int, 0>{}, c, J2{});
      step(std::integral_constant<int, 1>{}, c + 1, J0{});
      step(std::integral_constant<int, 0>{}, c + 1, J1{});
      step(std::integral_constant<int, 1>{}, c + 1, J2{});
    }
  }

  // ---- epilogue: the whole tile through ONE slab laid over the (now idle) weight ring -- a single chunk has nothing left to
  // prefetch -- then 2 RT rows per wave, RT at a time so that their loads, reductions and transcendentals overlap.  (In 32-row
  // passes through a slab of their own, two barriers each, the tail took 13 - 19 us of a 50 us launch.)
  stamp();      // 3: main loop done
  if (JV_ABLATE(p, 16)) return;
  const bool mish = p.act == ACT_MISH;      // (uniform) the one activation the estimator uses here
  // gain / offset of the following norm (ln2_out): requested here, in front of the slab writes and their barrier -- inside the row
  // pass each row group began by sitting out their L2 round trip
  rg_f32x4 g2 = {1.f, 1.f, 1.f, 1.f}, b2 = {0.f, 0.f, 0.f, 0.f};
  if (p.ln2_out) {
    g2 = *reinterpret_cast<const rg_f32x4*>(p.ln2_g + 4 * lane);
    b2 = *reinterpret_cast<const rg_f32x4*>(p.ln2_b + 4 * lane);
  }
  float* const slab = reinterpret_cast<float*>(rc_lds + SLAB_OFF);      // its own region: nothing to wait for before writing it
#pragma unroll
  for (int mt = 0; mt < RT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e) slab[(mt * 16 + kq * 4 + e) * RG_SLD + wave * 32 + nt * 16 + r16] = acc[mt][nt][e];
  // The row pass: rows [2 RT wave, + 2 RT) of the tile, RT at a time.  What a group of rows needs from global memory -- the
  // residual, the time-embedding vector of its utterance, the tracking slot's current maximum -- is requested a group AHEAD:
  // group 0's in front of the barrier behind the slab writes, group 1's before group 0 is computed (asked for inside the
  // group, as this pass first did, every group began by sitting out an L2 round trip with nothing to overlap it).
  constexpr int RPW = RT;
  struct RowIn { rg_f32x4 r[RPW], rv[RPW]; unsigned seen[RPW]; int slot[RPW]; bool ok[RPW], keep[RPW], tracked[RPW]; float inv[RPW]; };
  auto request = [&](auto ps_tag, RowIn& in) {
    constexpr int ps = decltype(ps_tag)::value;
#pragma unroll
    for (int jj = 0; jj < RPW; ++jj) {
      const int trow = wave * 2 * RT + ps * RT + jj;
      const long mrow = (long)m0 + trow;
      in.ok[jj] = mrow < p.M;
      const int2 ri = rowinfo[trow];
      in.slot[jj] = ri.y & (RI_KEEP - 1);
      in.inv[jj] = in.ok[jj] ? __uint_as_float((unsigned)ri.x) : 0.f;      // 1 / the power of two its window rows were staged with
      in.keep[jj] = in.ok[jj] && (ri.y & RI_KEEP) != 0;
      in.r[jj] = (p.res && in.ok[jj]) ? *reinterpret_cast<const rg_f32x4*>(p.res + mrow * p.ldr + 4 * lane) : rg_f32x4{0.f, 0.f, 0.f, 0.f};
      in.rv[jj] = (p.rowvec && in.ok[jj]) ? *reinterpret_cast<const rg_f32x4*>(p.rowvec + (long)in.slot[jj] * p.rowvec_ld + 4 * lane)
                                          : rg_f32x4{0.f, 0.f, 0.f, 0.f};
      in.tracked[jj] = in.ok[jj] && (ri.y & RI_TRACK) != 0;
      in.seen[jj] = 0xffffffffu;
      if (in.tracked[jj]) in.seen[jj] = *reinterpret_cast<const unsigned*>(p.amax_out + in.slot[jj]);
    }
  };
  auto rows = [&](auto ps_tag, const RowIn& in) {
    constexpr int ps = decltype(ps_tag)::value;
    rg_f32x4 v[RPW];
#pragma unroll
    for (int jj = 0; jj < RPW; ++jj) {
      const int trow = wave * 2 * RT + ps * RT + jj;
      v[jj] = *reinterpret_cast<const rg_f32x4*>(slab + trow * RG_SLD + 4 * lane) * (cs4 * in.inv[jj]) + b4;
    }
    if (p.ln) {
      float sum[RPW], sq[RPW];
#pragma unroll
      for (int jj = 0; jj < RPW; ++jj) sum[jj] = wave_sum((v[jj][0] + v[jj][1]) + (v[jj][2] + v[jj][3]));
#pragma unroll
      for (int jj = 0; jj < RPW; ++jj) {
        const rg_f32x4 d = v[jj] - sum[jj] * (1.f / 256.f);
        sq[jj] = wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]));
      }
      // 1 / sqrt(var + eps) of the group's rows in ONE evaluation, lane jj computing row jj's (rowblock_kernel.h)
      float var_l = sq[0];
#pragma unroll
      for (int jj = 1; jj < RPW; ++jj) var_l = lane == jj ? sq[jj] : var_l;
      const float rstd_l = 1.0f / sqrtf(var_l * (1.f / 256.f) + p.ln_eps);
#pragma unroll
      for (int jj = 0; jj < RPW; ++jj) {
        const float mean = sum[jj] * (1.f / 256.f);
        const float rstd = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rstd_l), jj));
        v[jj] = (v[jj] - mean) * rstd * gg + bb;
      }
    }
#pragma unroll
    for (int jj = 0; jj < RPW; ++jj) {
      const long mrow = (long)m0 + wave * 2 * RT + ps * RT + jj;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[jj][e] = in.keep[jj] ? (mish ? mish_fast(v[jj][e]) : act_apply(v[jj][e], p.act)) : 0.f;
      v[jj] = (v[jj] + in.rv[jj]) + in.r[jj];
      if (in.ok[jj]) *reinterpret_cast<rg_f32x4*>(p.out + mrow * p.ldo + 4 * lane) = v[jj];
    }
    if (p.ln2_out) {      // (uniform) the following block's norm1 of the stored rows -> operand planes (rowblock_kernel.h's row pass)
      float sum[RPW], sq[RPW];
#pragma unroll
      for (int jj = 0; jj < RPW; ++jj) sum[jj] = wave_sum((v[jj][0] + v[jj][1]) + (v[jj][2] + v[jj][3]));
#pragma unroll
      for (int jj = 0; jj < RPW; ++jj) {
        const rg_f32x4 d = v[jj] - sum[jj] * (1.f / 256.f);
        sq[jj] = wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]));
      }
      float var_l = sq[0];
#pragma unroll
      for (int jj = 1; jj < RPW; ++jj) var_l = lane == jj ? sq[jj] : var_l;
      const float rstd_l = 1.0f / sqrtf(var_l * (1.f / 256.f) + p.ln_eps);
#pragma unroll
      for (int jj = 0; jj < RPW; ++jj) {
        const long mrow = (long)m0 + wave * 2 * RT + ps * RT + jj;
        const float mean = sum[jj] * (1.f / 256.f);
        const float rstd = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rstd_l), jj));
        const rg_f32x4 y = (v[jj] - mean) * rstd * g2 + b2;
        const Split2 s0 = split2h_pair(y[0] * p.ln2_scale, y[1] * p.ln2_scale);
        const Split2 s1 = split2h_pair(y[2] * p.ln2_scale, y[3] * p.ln2_scale);
        if (in.ok[jj]) {
          unsigned short* const o2 = p.ln2_out + mrow * 256 + 4 * lane;
          *reinterpret_cast<rg_u32x2*>(o2) = rg_u32x2{s0.h, s1.h};
          *reinterpret_cast<rg_u32x2*>(o2 + p.ln2_plane) = rg_u32x2{s0.l, s1.l};
        }
      }
    }
    if (p.amax_out) {
#pragma unroll
      for (int jj = 0; jj < RPW; ++jj) {
        unsigned u = 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) u = max(u, __float_as_uint(v[jj][e]) & 0x7fffffffu);
        if (in.tracked[jj] && __builtin_amdgcn_ballot_w64(u > in.seen[jj]) != 0) {
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) u = max(u, (unsigned)__shfl_xor((int)u, o));
          if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(p.amax_out + in.slot[jj]), u);
        }
      }
    }
  };
  {
    RowIn in0, in1;
    request(std::integral_constant<int, 0>{}, in0);      // (rowinfo was written by this workgroup's threads ahead of the main loop's barriers)
    rg_lds_barrier();
    stamp();      // 4: slab written, barrier passed
    request(std::integral_constant<int, 1>{}, in1);
    rows(std::integral_constant<int, 0>{}, in0);
    stamp();      // 5: first row group done
    rows(std::integral_constant<int, 1>{}, in1);
    stamp();      // 6: second row group done
  }
  if constexpr (RES) {
    // ---- res_conv's rows: a second pass through the same slab -- res_out = acc_r * res_cs / a_scale + res_bias, fp32 rows that
    // block2's launch adds.  (A masked input row was staged as zeros: its output is the bias, as in a launch of its own.)
    rg_lds_barrier();      // every wave has read its rows of the first pass
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) slab[(mt * 16 + kq * 4 + e) * RG_SLD + wave * 32 + nt * 16 + r16] = accr[mt][nt][e];
    rg_lds_barrier();
#pragma unroll
    for (int jj = 0; jj < 2 * RT; ++jj) {
      const int trow = wave * 2 * RT + jj;
      const long mrow = (long)m0 + trow;
      const float inv = __uint_as_float((unsigned)rowinfo[trow].x);
      const rg_f32x4 v = *reinterpret_cast<const rg_f32x4*>(slab + trow * RG_SLD + 4 * lane) * (csr4 * inv) + br4;
      if (mrow < p.M) *(__attribute__((address_space(1))) rg_f32x4*)(p.res_out + mrow * 256 + 4 * lane) = v;
    }
  }
  rg_wait_vmcnt<0>();      // the wrapped-around W loads: bq stays reserved until they have landed (rowgemm_wd_kernel)
  landed_w(bq[0][0][0], bq[0][0][1], bq[0][1][0], bq[0][1][1]);
  landed_w(bq[1][0][0], bq[1][0][1], bq[1][1][0], bq[1][1][1]);
  asm volatile("" ::"v"(warm));
  stamp();      // 7 (6 without the res_conv pass): every store of this wave acknowledged
}


}  // namespace jv
