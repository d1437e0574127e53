// Row-owning fp16x3 GEMM for the estimator's transformer linears (jyutvoice/flow/transformer.py:355-443: attn1.to_out.0,
// ff.net.0.proj, ff.net.2, and the fused to_q/to_k/to_v), built for the shape that dominates the path: M ~ 19.5 K rows
// (2 B utterances x (T + 4) frames), K = 256 ... 1024, N = 256 ... 1536.
//
// Why another GEMM.  The tile kernels (conv_gemm_x6_kernel.h) cut M x N into 64x64 ... 160x128 tiles.  For the N = 256
// linears that left 1220 small tiles, each pulling 16 KB through its CU's load path per 24 MFMAs (A re-staged by every
// column tile, 0.125 B per multiply-add): round 1's PMC put the matrix pipe at 25 % with the waves waiting on staging.
// Here ONE 8-wave workgroup per CU owns R = 16 RT whole rows (RT = 5: 80 rows -> 244 workgroups for 19.5 K rows, one
// round on 256 CUs) and all N columns, 256 at a time:
//   * both operands arrive as fp16 planes by LDS-DMA (global_load_lds_dwordx4) into a 3-stage ring -- the producers
//     (LayerNorm, attention, this kernel's own epilogues) write the A planes, so the main loop has no VALU work at all;
//     waits are counted (s_waitcnt vmcnt(n)) and the one barrier per 32-deep K step is a bare s_barrier, so two steps
//     of DMA stay in flight across it;
//   * v_mfma_f32_16x16x32_f16: 16-row granularity is what lets 80-row tiles exist (32x32 tiles would mean 64 or 96 rows:
//     304 or 203 workgroups), and the 16x16 shape holds a higher clock under load (MI355X_MICROARCH.md);
//   * bytes per multiply-add: (1/256 + 1/80) x 4 B = 0.066, half the 64x64 tile's;
//   * the workgroup owns whole rows, so what follows the GEMM row-wise runs in its epilogue: bias, residual, the NEXT
//     LayerNorm (its result written as the next GEMM's pre-split operand), exact GELU, the measured-bound tracking.
// Precision: identical arithmetic to the tile kernels' fp16x3 (three products hh' + hl' + lh' of the 2-plane splits of
// A * a_scale and W * 2^e_n, fp32 accumulate, K in ascending 32-chunks): the same bits for the same operands.
#pragma once
#include <math.h>

#include <string>
#include <type_traits>

#include "jv_common.h"
#include "jv_device.h"

namespace jv {

typedef _Float16 rg_f16x8 __attribute__((ext_vector_type(8)));
typedef float rg_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int rg_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int rg_u32x2 __attribute__((ext_vector_type(2)));

enum RowGemmEpi : int {
  RG_PLAIN = 0,      // out = acc * colscale / a_scale (+ bias): fp32 rows
  RG_GELU_PL = 1,    // gelu(...) written as fp16 planes of value * out2_scale (ff.net.0 -> ff.net.2's operand)
  RG_RES = 2,        // ... + res -> fp32 rows (+ amax tracking)
  RG_RES_LN = 3,     // ... + res -> fp32 rows, then LayerNorm_256 of the row -> fp16 planes * ln_scale (the next GEMM's operand)
  RG_QKV = 4,        // N = 1536 = q | k | v: q -> fp32 rows [.,512] (out), k and v -> fp16 planes [2][rows][1024] (out2) of
                     // k * out2_scale, v * out2_scale2: the operands attention64_planes takes without splitting anything
};

struct RowGemmArgs {
  // A * a_scale as two fp16 planes [2][a_rows][lda2] (rows past a_rows - 1 are clamped: their outputs are never stored)
  const unsigned short* A2;
  long a2_plane, a_rows;
  int lda2;
  int M, K, N;      // K % 32 == 0, N % 256 == 0
  // W[n][:] * 2^e_n as two fp16 planes [2][n_rows][ldw], colscale[n] = 2^-e_n (registry.hip `half3`)
  const unsigned short* W2;
  long w2_plane;
  int ldw;
  const unsigned short* Wf;      // the same planes in fragment order (rowgemm_wd_kernel), plane stride wf_plane halves; null: W2 through LDS
  long wf_plane;
  const float* colscale;
  float a_scale;      // the power of two A was scaled with (a load-time bound, GemmW::a_scale)
  const float* bias;  // [N] or null
  float* out;         // RG_PLAIN / RG_RES / RG_RES_LN
  long ldo;
  const float* res;   // RG_RES / RG_RES_LN (may alias out)
  long ldr;
  unsigned short* out2;      // RG_GELU_PL: [2][rows][ldo2];  RG_RES_LN: the LayerNorm planes [2][rows][256]
  long out2_plane;
  int ldo2;
  float out2_scale, out2_scale2;
  const float *ln_g, *ln_b;
  float ln_eps;
  // measured-bound tracking of what RG_RES / RG_RES_LN store (ConvGemmArgs::amax_out): slot = row_slot[m], rows with
  // row_mask[m] == 0 are padding and not tracked
  float* amax_out;
  const int* row_slot;
  const unsigned char* row_mask;
  long alg_rows;      // profiler: real frames
  int nsplit;         // rowgemm_wa_kernel: the N / 256 column chunks are dealt to nsplit workgroups per row tile (grid.y); 0 / 1: one
  int rt;             // tile height in 16-row units, 2 .. 5; 0: rowgemm_tile(M).  A stand-alone launch need not take the fused block's:
                      // the L2 -> CU traffic of a launch is (row tiles) x (weight bytes), so few tall tiles x column split moves the least
  int ablate;         // tuning aid (JV_RG_ABLATE, tuning builds): 1 no loads in the loop, 2 no MFMAs, 4 no barrier/wait (LDS kernel),
                      // W-direct kernel also: 8 no LDS fragment reads, 16 no W loads, 32 no A DMA, 64 no row pass
};

constexpr int RG_SLD = 260;                      // slab row stride in floats (256 + 4: conflict-free b32 writes, aligned b128 reads)
constexpr int RG_SLAB_ROWS = 32;
constexpr int RG_SLAB_BYTES = RG_SLAB_ROWS * RG_SLD * 4;
template <int RT> constexpr int rg_stage_bytes() { return 2 * 16 * RT * 64 + 2 * 256 * 64; }
template <int RT> constexpr int rg_lds_bytes() { return 3 * rg_stage_bytes<RT>() + RG_SLAB_BYTES; }

// 16-byte slot key of a row inside its 16-row group: with 64-byte rows (32 fp16), lane l of a 16x16x32 fragment read
// takes row l & 15, k-slot l >> 4; XOR-ing bit 1 of the slot with bit 2 of the row puts the 16 lanes of every
// ds_read_b128 lane group on 16 distinct 16-byte bank slots (brute-forced over the four lane groups of MI355X_MICROARCH.md)
__device__ __forceinline__ int rg_key(int r) { return ((r >> 2) & 1) << 1; }

template <int N>
__device__ __forceinline__ void rg_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// Bare barriers (the "memory" clobber keeps the compiler from moving LDS / global accesses across them; s_barrier itself
// waits for no counter).  __syncthreads() would add s_waitcnt vmcnt(0): in the main loop that drains the two steps of
// LDS-DMA kept in flight, in the epilogue it waits for the row stores to be acknowledged.
__device__ __forceinline__ void rg_barrier() { asm volatile("s_barrier" ::: "memory"); }
__device__ __forceinline__ void rg_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int RT, int EPI>
__global__ __launch_bounds__(512, 2) void rowgemm_kernel(const RowGemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rg_lds[];
  constexpr int R = 16 * RT;
  constexpr int STAGE = rg_stage_bytes<RT>();
  constexpr int A_PLANE = R * 64, W_PLANE = 256 * 64, W_OFF = 2 * A_PLANE;
  constexpr int NPIECE = 2 * RT + 32;            // 1 KiB DMA pieces per stage: RT per A plane, 16 per W plane
  constexpr int PPW = (NPIECE + 7) / 8;          // issued per wave and step, at most
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.x * R;
  const int KS = p.K >> 5, NC = p.N >> 8;
  const int total = KS * NC;
  float* const slab = reinterpret_cast<float*>(rg_lds + 3 * STAGE);

  // ---- warm this XCD's L2 with the weight planes.  Every workgroup streams ALL of W, a new 32 KB tile per step that no
  // CU reads twice; in the estimator each launch has its own weights, cold in every cache, and with two steps (84 KB) in
  // flight per CU the loop then runs at 84 KB per HBM latency (traced: 25 % slower than the same kernel on L2-hot weights).
  // The workgroups that share an XCD (blockIdx % 8 under round-robin placement: a speed-only guess) each touch a slice of W
  // once, 128 bytes apart, so the whole matrix is on its way into that L2 while the first steps run.
  float warm = 0.f;
  {
    const long lpp = ((long)p.N * p.ldw * 2) >> 7;                 // 128-byte lines per plane
    const int grp = blockIdx.x >> 3, ngrp = (gridDim.x + 7) >> 3;
    const long per = (2 * lpp + ngrp - 1) / ngrp;
    const long l = (long)grp * per + tid;
    if (tid < per && l < 2 * lpp) {
      const int pl = l >= lpp;
      warm = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.W2 + (long)pl * p.w2_plane) + ((l - pl * lpp) << 7));
    }
    // (per <= 512 for every estimator shape at >= 96 workgroups; a larger W is simply warmed in part)
  }

  // ---- this wave's DMA pieces: per-lane source pointer at (chunk 0, k 0), wave-uniform LDS offset, per-chunk advance ----
  const unsigned short* src[PPW];
  int dst[PPW];
  long cadv[PPW];
  {
    const int prow = lane >> 2, pslot = (lane & 3) ^ rg_key(prow);      // the DMA writes lane-linear: swizzle the SOURCE k-slot
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int pc = wave + 8 * i;
      if (pc < 2 * RT) {
        const int pl = pc / RT, g = pc % RT;
        long row = (long)m0 + g * 16 + prow;
        row = row < p.a_rows ? row : p.a_rows - 1;
        src[i] = p.A2 + (long)pl * p.a2_plane + row * p.lda2 + 8 * pslot;
        dst[i] = pl * A_PLANE + g * 1024;
        cadv[i] = 0;
      } else {
        const int q = pc - 2 * RT, pl = (q >> 4) & 1, g = q & 15;
        src[i] = p.W2 + (long)pl * p.w2_plane + (long)(g * 16 + prow) * p.ldw + 8 * pslot;
        dst[i] = W_OFF + pl * W_PLANE + g * 1024;
        cadv[i] = 256L * p.ldw;
      }
    }
  }
  const int my_pieces = (NPIECE - wave + 7) / 8;      // pieces this wave really issues per step (wave-uniform)
  // The pieces of one step are issued ONE AT A TIME between the step's MFMA groups, not in a burst behind the barrier:
  // an LDS-DMA issue holds its wave for ~100 cycles, and in a burst both waves of a SIMD sit in theirs together while the
  // matrix pipe idles (ablation: the loop took 33 us where DMA alone and MFMA alone took 20 and 21).  `cur[i]` walks the
  // (chunk, k) sequence incrementally: + 32 halves per step, + cadv - K at a chunk boundary.
  const unsigned short* cur[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) cur[i] = src[i];
  int ik = 0;      // k-step of the next step to issue
  auto issue_piece = [&](auto itag, int stage) {
    constexpr int i = decltype(itag)::value;
    if (wave + 8 * i < NPIECE) {      // wave-uniform
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)cur[i],
                                       (__attribute__((address_space(3))) void*)(rg_lds + stage * STAGE + dst[i]), 16, 0, 0);
    }
  };
  auto advance = [&]() {      // after all pieces of a step have been issued
    const bool wrap = ++ik == KS;
    if (wrap) ik = 0;
#pragma unroll
    for (int i = 0; i < PPW; ++i) cur[i] += wrap ? cadv[i] - p.K + 32 : 32;
  };
  auto issue_all = [&](int stage) {
    issue_piece(std::integral_constant<int, 0>{}, stage);
    if constexpr (PPW > 1) issue_piece(std::integral_constant<int, 1>{}, stage);
    if constexpr (PPW > 2) issue_piece(std::integral_constant<int, 2>{}, stage);
    if constexpr (PPW > 3) issue_piece(std::integral_constant<int, 3>{}, stage);
    if constexpr (PPW > 4) issue_piece(std::integral_constant<int, 4>{}, stage);
    if constexpr (PPW > 5) issue_piece(std::integral_constant<int, 5>{}, stage);
    advance();
  };
  // all but this wave's `keep` youngest pieces have landed (keep = one step's worth, or 0)
  auto wait_landed = [&](bool next_in_flight) {
    if (!next_in_flight) {
      rg_wait_vmcnt<0>();
    } else if (my_pieces == PPW) {
      rg_wait_vmcnt<PPW>();
    } else {
      rg_wait_vmcnt<PPW - 1>();
    }
  };

  // per-lane fragment addresses inside a stage
  const int fslot = (kq ^ rg_key(r16)) << 4;
  const int a_off = r16 * 64 + fslot;                                   // + pl * A_PLANE + mt * 1024
  const int w_off = W_OFF + (wave * 32 + r16) * 64 + fslot;             // + pl * W_PLANE + nt * 1024

  rg_f32x4 acc[RT][2];
  issue_all(0);
  if (total > 1) issue_all(1);
  int s = 0;
  for (int c = 0; c < NC; ++c) {
    // the epilogue's per-column constants of this chunk, requested before its main loop so that their latency is not paid
    // behind it
    const int n0 = c * 256 + 4 * lane;      // this lane's four columns in the row pass
    rg_f32x4 cs4 = *reinterpret_cast<const rg_f32x4*>(p.colscale + n0);
    cs4 = cs4 * (1.0f / p.a_scale);        // powers of two: exact
    rg_f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) b4 = *reinterpret_cast<const rg_f32x4*>(p.bias + n0);
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = rg_f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < KS; ++ks, ++s) {
      // step s's pieces: mine have landed (all but the younger step's), then everyone's have -- and every wave is done
      // reading the stage that step s + 2 will overwrite (it was read in step s - 1)
      if (!JV_ABLATE(p, 4)) {
        wait_landed(s + 1 < total);
        rg_barrier();
      }
      const bool more = s + 2 < total && !JV_ABLATE(p, 1);
      const int nstage = (s + 2) % 3;
      if (JV_ABLATE(p, 2)) {
        if (more) issue_all(nstage);
        continue;
      }
      const unsigned char* const st = rg_lds + (s % 3) * STAGE;
      rg_u32x4 b[2][2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) b[nt][pl] = *reinterpret_cast<const rg_u32x4*>(st + w_off + pl * W_PLANE + nt * 1024);
      auto group = [&](auto mtag) {
        constexpr int mt = decltype(mtag)::value;
        rg_u32x4 a[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) a[pl] = *reinterpret_cast<const rg_u32x4*>(st + a_off + pl * A_PLANE + mt * 1024);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          rg_f32x4 t = acc[mt][nt];
          auto mm = [&](const rg_u32x4& x, const rg_u32x4& y) {
            t = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(rg_f16x8, x), __builtin_bit_cast(rg_f16x8, y), t, 0, 0, 0);
          };
          mm(a[1], b[nt][0]);      // smallest terms first, as the tile kernels do
          mm(a[0], b[nt][1]);
          mm(a[0], b[nt][0]);
          acc[mt][nt] = t;
        }
        // this group's share of the next-but-one step's DMA: pieces mt, mt + RT, ... of the wave's list
        if (more) {
          issue_piece(std::integral_constant<int, mt>{}, nstage);
          if constexpr (mt + RT < PPW) issue_piece(std::integral_constant<int, (mt + RT < PPW ? mt + RT : 0)>{}, nstage);
          if constexpr (mt + 2 * RT < PPW) issue_piece(std::integral_constant<int, (mt + 2 * RT < PPW ? mt + 2 * RT : 0)>{}, nstage);
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the interleave: the scheduler would hoist the DMA issues together
      };
      group(std::integral_constant<int, 0>{});
      if constexpr (RT > 1) group(std::integral_constant<int, 1>{});
      if constexpr (RT > 2) group(std::integral_constant<int, 2>{});
      if constexpr (RT > 3) group(std::integral_constant<int, 3>{});
      if constexpr (RT > 4) group(std::integral_constant<int, 4>{});
      if (more) advance();
    }

    // ---- epilogue of chunk c: 32 rows at a time through the slab (its own LDS region: the ring keeps streaming) ----
    // the rows [trow0, trow0 + RPW) of the tile, held in slab rows [srow0, ...) of `sl`, one wave per row
    auto rows_body = [&](auto rpw_tag, const float* sl, const int trow0, const int srow0) {
      constexpr int RPW = decltype(rpw_tag)::value;
      rg_f32x4 v[RPW];
      long mrow[RPW];
      bool ok[RPW];
#pragma unroll
      for (int j = 0; j < RPW; ++j) {
        const int trow = trow0 + j;
        mrow[j] = (long)m0 + trow;
        ok[j] = trow < R && mrow[j] < p.M;
        v[j] = *reinterpret_cast<const rg_f32x4*>(sl + (srow0 + j) * RG_SLD + 4 * lane) * cs4 + b4;
      }
      if constexpr (EPI == RG_PLAIN) {
#pragma unroll
        for (int j = 0; j < RPW; ++j)
          if (ok[j]) *reinterpret_cast<rg_f32x4*>(p.out + mrow[j] * p.ldo + n0) = v[j];
      } else if constexpr (EPI == RG_QKV) {
        if (c < 2) {      // q (chunk-uniform branch)
#pragma unroll
          for (int j = 0; j < RPW; ++j)
            if (ok[j]) *reinterpret_cast<rg_f32x4*>(p.out + mrow[j] * p.ldo + n0) = v[j];
        } else {
          const float sc = c < 4 ? p.out2_scale : p.out2_scale2;
#pragma unroll
          for (int j = 0; j < RPW; ++j) {
            if (!ok[j]) continue;
            const Split2 s0 = split2h_pair(v[j][0] * sc, v[j][1] * sc);
            const Split2 s1 = split2h_pair(v[j][2] * sc, v[j][3] * sc);
            unsigned short* o2 = p.out2 + mrow[j] * p.ldo2 + (n0 - 512);
            *reinterpret_cast<rg_u32x2*>(o2) = rg_u32x2{s0.h, s1.h};
            *reinterpret_cast<rg_u32x2*>(o2 + p.out2_plane) = rg_u32x2{s0.l, s1.l};
          }
        }
      } else if constexpr (EPI == RG_GELU_PL) {
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
          if (!ok[j]) continue;
          rg_f32x4 t = v[j];
#pragma unroll
          for (int e = 0; e < 4; ++e) t[e] = gelu_erf(t[e]);
          const Split2 s0 = split2h_pair(t[0] * p.out2_scale, t[1] * p.out2_scale);
          const Split2 s1 = split2h_pair(t[2] * p.out2_scale, t[3] * p.out2_scale);
          unsigned short* o2 = p.out2 + mrow[j] * p.ldo2 + n0;
          *reinterpret_cast<rg_u32x2*>(o2) = rg_u32x2{s0.h, s1.h};
          *reinterpret_cast<rg_u32x2*>(o2 + p.out2_plane) = rg_u32x2{s0.l, s1.l};
        }
      } else {
        // + residual -> fp32 rows (+ tracking) (-> LayerNorm -> planes)
        rg_f32x4 r[RPW];
        unsigned seen[RPW];
        bool tracked[RPW];
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
          r[j] = ok[j] ? *reinterpret_cast<const rg_f32x4*>(p.res + mrow[j] * p.ldr + 4 * lane) : rg_f32x4{0.f, 0.f, 0.f, 0.f};
          tracked[j] = false;
          seen[j] = 0xffffffffu;
          if (p.amax_out && ok[j]) {
            tracked[j] = !p.row_mask || p.row_mask[mrow[j]] != 0;
            // (a plain, cacheable load: the slot only grows, so a stale value is a valid lower bound)
            seen[j] = *reinterpret_cast<const unsigned*>(p.amax_out + (p.row_slot ? p.row_slot[mrow[j]] : 0));
          }
        }
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
          v[j] += r[j];
          if (ok[j]) *reinterpret_cast<rg_f32x4*>(p.out + mrow[j] * p.ldo + 4 * lane) = v[j];
        }
        if (p.amax_out) {
#pragma unroll
          for (int j = 0; j < RPW; ++j) {
            unsigned u = 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e) u = max(u, __float_as_uint(v[j][e]) & 0x7fffffffu);
            // wave-uniform branch (a wave holds whole rows): nothing to do once the slot holds a larger value
            if (tracked[j] && __builtin_amdgcn_ballot_w64(u > seen[j]) != 0) {
#pragma unroll
              for (int o = 32; o > 0; o >>= 1) u = max(u, (unsigned)__shfl_xor((int)u, o));
              if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(p.amax_out + (p.row_slot ? p.row_slot[mrow[j]] : 0)), u);
            }
          }
        }
        if constexpr (EPI == RG_RES_LN) {
          // LayerNorm over the row's 256 channels (two-pass, as rowops.hip's layernorm256_kernel), written as the next
          // GEMM's pre-split operand
          const rg_f32x4 gg = *reinterpret_cast<const rg_f32x4*>(p.ln_g + 4 * lane);
          const rg_f32x4 bb = *reinterpret_cast<const rg_f32x4*>(p.ln_b + 4 * lane);
          float sum[RPW], sq[RPW];
#pragma unroll
          for (int j = 0; j < RPW; ++j) sum[j] = wave_sum((v[j][0] + v[j][1]) + (v[j][2] + v[j][3]));
#pragma unroll
          for (int j = 0; j < RPW; ++j) {
            const rg_f32x4 d = v[j] - sum[j] * (1.f / 256.f);
            sq[j] = wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]));
          }
#pragma unroll
          for (int j = 0; j < RPW; ++j) {
            if (!ok[j]) continue;
            const float mean = sum[j] * (1.f / 256.f);
            const float rstd = 1.0f / sqrtf(sq[j] * (1.f / 256.f) + p.ln_eps);
            const rg_f32x4 y = (v[j] - mean) * rstd * gg + bb;
            const Split2 s0 = split2h_pair(y[0] * p.out2_scale, y[1] * p.out2_scale);
            const Split2 s1 = split2h_pair(y[2] * p.out2_scale, y[3] * p.out2_scale);
            unsigned short* o2 = p.out2 + mrow[j] * p.ldo2 + 4 * lane;
            *reinterpret_cast<rg_u32x2*>(o2) = rg_u32x2{s0.h, s1.h};
            *reinterpret_cast<rg_u32x2*>(o2 + p.out2_plane) = rg_u32x2{s0.l, s1.l};
          }
        }
      }
    };
    if constexpr (EPI == RG_RES || EPI == RG_RES_LN) {
      // one 256-column chunk: nothing left to prefetch, so the whole tile goes through ONE slab laid over the idle ring and
      // every wave takes its 2 RT rows RT at a time (loads, reductions and conversions of RT rows overlap).  In 32-row passes
      // through the small slab, two barriers each, this tail was 10 us of a 44 us launch.
      rg_lds_barrier();      // every wave is done reading the ring
      float* const big = reinterpret_cast<float*>(rg_lds);
#pragma unroll
      for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int e = 0; e < 4; ++e) big[(mt * 16 + kq * 4 + e) * RG_SLD + wave * 32 + nt * 16 + r16] = acc[mt][nt][e];
      rg_lds_barrier();
      rows_body(std::integral_constant<int, RT>{}, big, wave * 2 * RT, wave * 2 * RT);
      rows_body(std::integral_constant<int, RT>{}, big, wave * 2 * RT + RT, wave * 2 * RT + RT);
    } else {
#pragma unroll
      for (int ps = 0; ps < (RT + 1) / 2; ++ps) {
#pragma unroll
        for (int ml = 0; ml < 2; ++ml) {
          if (2 * ps + ml < RT) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
              for (int e = 0; e < 4; ++e)
                slab[(ml * 16 + kq * 4 + e) * RG_SLD + wave * 32 + nt * 16 + r16] = acc[2 * ps + ml][nt][e];
          }
        }
        rg_lds_barrier();
        rows_body(std::integral_constant<int, RG_SLAB_ROWS / 8>{}, slab, ps * 32 + wave * (RG_SLAB_ROWS / 8), wave * (RG_SLAB_ROWS / 8));
        rg_lds_barrier();      // the slab is rewritten by the next pass / chunk
      }
    }
  }
  asm volatile("" ::"v"(warm));      // the warm-up load is waited for here, not before
}

// ---- the same GEMM with the weight fragments loaded straight into registers ("W-direct") --------------------------------
// In the kernel above both operands go through LDS.  Per 32-deep step and CU that is 42 KB of DMA writes plus 112 KB of
// fragment reads (every wave reads all 80 rows of A, 10 KB, and its own 32 columns of W, 4 KB) = ~1200 LDS cycles at
// 128 B/clk against 960 matrix-pipe cycles (30 MFMAs x 16 clk x 2 waves per SIMD): LDS-bound, and the ablations agree (DMA
// alone and MFMA alone each take 60 % of the loop they make together).  But W is not shared inside the workgroup -- each
// wave owns 32 columns -- so staging it in LDS buys nothing except asynchrony.  Here the weights are stored a second time in
// FRAGMENT ORDER (RowGemmArgs::Wf: for each plane, 16-column block and 32-deep k-step, 1 KB holding lane l's 8 halves at
// l * 16 B; registry.hip packs it at load time), each wave loads its four fragments of a step with four fully coalesced
// global_load_dwordx4 two steps ahead into a register double buffer, and LDS carries only A: 10 KB written and 80 KB read
// per step, ~700 cycles, under the matrix pipe's 960.  The ring is A-only (3 x 10 KB), which leaves room for a slab of the
// whole tile (83 KB) beside it: every chunk's epilogue is one pass, and the ring keeps streaming underneath it.
constexpr int RGW_NST = 3;
template <int RT> constexpr int rgw_stage_bytes() { return 2 * 16 * RT * 64; }
template <int RT> constexpr int rgw_lds_bytes() { return RGW_NST * rgw_stage_bytes<RT>() + 16 * RT * RG_SLD * 4; }

template <int RT, int EPI>
__global__ __launch_bounds__(512, 2) void rowgemm_wd_kernel(const RowGemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rg_lds[];
  constexpr int R = 16 * RT;
  constexpr int STAGE = rgw_stage_bytes<RT>();
  constexpr int A_PLANE = R * 64;
  constexpr int NPIECE = 2 * RT;                 // 1 KiB DMA pieces per stage: RT per A plane
  constexpr int PPW = (NPIECE + 7) / 8;          // issued per wave and step, at most
  constexpr int NWL = 4;                         // weight loads per wave and step: 2 column blocks x 2 planes
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.x * R;
  // (the residual epilogues own whole 256-wide rows: one chunk, known at compile time, so that nothing of the main loop is
  // live in their register-hungry row pass)
  const int KS = p.K >> 5, NC = (EPI == RG_RES || EPI == RG_RES_LN) ? 1 : p.N >> 8;
  const int total = KS * NC;
  float* const big = reinterpret_cast<float*>(rg_lds + RGW_NST * STAGE);

  // warm this XCD's L2 with the weights (see rowgemm_kernel above)
  float warm = 0.f;
  {
    const long lpp = ((long)p.N * p.K * 2) >> 7;                   // 128-byte lines per plane
    const int grp = blockIdx.x >> 3, ngrp = (gridDim.x + 7) >> 3;
    const long per = (2 * lpp + ngrp - 1) / ngrp;
    const long l = (long)grp * per + tid;
    if (tid < per && l < 2 * lpp) {
      const int pl = l >= lpp;
      warm = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.Wf + (long)pl * p.wf_plane) + ((l - pl * lpp) << 7));
    }
  }

  // ---- A: this wave's DMA pieces ----
  const unsigned short* cur[PPW];
  int dst[PPW];
  {
    const int prow = lane >> 2, pslot = (lane & 3) ^ rg_key(prow);
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int pc = wave + 8 * i;
      const int pl = pc / RT, g = pc % RT;
      long row = (long)m0 + g * 16 + prow;
      row = row < p.a_rows ? row : p.a_rows - 1;
      cur[i] = p.A2 + (long)(pl & 1) * p.a2_plane + row * p.lda2 + 8 * pslot;
      dst[i] = (pl & 1) * A_PLANE + g * 1024;
    }
  }
  const int my_pieces = (NPIECE - wave + 7) / 8 > 0 ? (NPIECE - wave + 7) / 8 : 0;      // wave-uniform
  int ik = 0;
  auto issue_piece = [&](auto itag, int stage) {
    constexpr int i = decltype(itag)::value;
    if (wave + 8 * i < NPIECE) {      // wave-uniform
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)cur[i],
                                       (__attribute__((address_space(3))) void*)(rg_lds + stage * STAGE + dst[i]), 16, 0, 0);
    }
  };
  auto advance = [&]() {      // A walks k within the row and starts over for every chunk
    const bool wrap = ++ik == KS;
    if (wrap) ik = 0;
#pragma unroll
    for (int i = 0; i < PPW; ++i) cur[i] += wrap ? 32 - p.K : 32;
  };
  // ---- W: fragment-ordered, [plane][KS][N/16][64 lanes][8 halves] (k-step major: pack_wfrag); this wave's column blocks are c*16 + wave*2 + nt ----
  // per-lane pointers of the four fragments (column block nt, plane pl) at (chunk 0, step 0); `woff` walks the (chunk,
  // k-step) sequence in halves.  (A scalar base with 32-bit lane offsets -- global_load_dwordx4 v, v_off, s[base] -- would
  // save the 64-bit adds, but from inline asm it faulted: the base reached the asm through v_readfirstlane, and a VMEM
  // instruction reading an SGPR that a VALU instruction has just written needs wait states the compiler's hazard
  // recognizer does not insert around inline asm.)
  const unsigned short* wbase[2][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) wbase[nt][pl] = p.Wf + (long)pl * p.wf_plane + (long)(wave * 2 + nt) * 512 + lane * 8;
  long woff = 0;
  int wk = 0;
  // The register double buffer is loaded and waited for by hand (inline asm): left to the compiler, the wait in front of
  // the first MFMA of a step becomes s_waitcnt vmcnt(0) -- its bookkeeping gives up on loads carried around the loop --
  // and the two steps of lookahead are gone.  The asm load is invisible to that bookkeeping, so (a) nothing but the MFMAs
  // below may read bq, which the "+v" operands of wait_w order behind the counted wait, and (b) the register allocator
  // must not copy bq while a load is in flight: tools/check_rowgemm_isa.py checks the built code object for both.
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
  int wc = 0;
  const long wstep = 512L * 16 * NC;      // a k-step on: N / 16 column blocks x 1 KB (512 halves) per plane
  auto advance_w = [&]() {      // (chunk wc, step wk) sits (wk N/16 + 16 wc) KB into a plane; back to the start after the last chunk
    const bool wrap = ++wk == KS;
    if (wrap) {
      wk = 0;
      const bool end = ++wc == NC;
      if (end) wc = 0;
      woff += 16L * 512 - (KS - 1) * wstep - (end ? 16L * 512 * NC : 0L);
    } else {
      woff += wstep;
    }
  };

  const int fslot = (kq ^ rg_key(r16)) << 4;
  const int a_off = r16 * 64 + fslot;

  rg_f32x4 acc[RT][2];
  // A fragments, double-buffered in registers by step parity: the reads of step s + 1 are issued in the middle of step s
  rg_u32x4 af[2][RT][2];
  auto read_a = [&](auto par_tag, int stage) {
    constexpr int par = decltype(par_tag)::value;
    const unsigned char* const st = rg_lds + stage * STAGE;
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) af[par][mt][pl] = *reinterpret_cast<const rg_u32x4*>(st + a_off + pl * A_PLANE + mt * 1024);
  };
  // prologue: A of steps 0, 1, 2 into the three stages, W of steps 0 and 1 into the two register buffers (K % 64 == 0: there
  // are at least two steps); everything is waited for once, then step 0's fragments are read
  issue_piece(std::integral_constant<int, 0>{}, 0);
  if constexpr (PPW > 1) issue_piece(std::integral_constant<int, 1>{}, 0);
  advance();
  load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
  advance_w();
  issue_piece(std::integral_constant<int, 0>{}, 1);
  if constexpr (PPW > 1) issue_piece(std::integral_constant<int, 1>{}, 1);
  advance();
  load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
  load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
  advance_w();
  issue_piece(std::integral_constant<int, 0>{}, 2);
  if constexpr (PPW > 1) issue_piece(std::integral_constant<int, 1>{}, 2);
  advance();
  rg_wait_vmcnt<0>();
  landed_w(bq[0][0][0], bq[0][0][1], bq[0][1][0], bq[0][1][1]);
  landed_w(bq[1][0][0], bq[1][0][1], bq[1][1][0], bq[1][1][1]);
  rg_barrier();
  read_a(std::integral_constant<int, 0>{}, 0);

  int s = 0, s3 = 0;      // step, step % 3
  bool waited = false;
  for (int c = 0; c < NC; ++c) {
    const int n0 = c * 256 + 4 * lane;
    rg_f32x4 cs4 = *reinterpret_cast<const rg_f32x4*>(p.colscale + n0);
    cs4 = cs4 * (1.0f / p.a_scale);
    rg_f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) b4 = *reinterpret_cast<const rg_f32x4*>(p.bias + n0);
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = rg_f32x4{0.f, 0.f, 0.f, 0.f};

    // One 32-deep step, software-pipelined across the workgroup barrier; `par` = s & 1 picks the register buffers.
    //   block 0: column block 0 x all row groups (fragments af[par], bq[par][0] -- both in registers since the middle of
    //            step s - 1), then bq[par][0] is reloaded for step s + 2;
    //   middle:  counted wait + barrier: A of step s + 1 has landed for every wave (and this wave's W0 of step s + 1 and W1
    //            of step s), and every wave has finished reading stage s % 3 (read in the middle of step s - 1); step s + 1's
    //            fragments are requested into af[par ^ 1] -- their LDS latency hides behind block 1 -- and step s + 3's A
    //            pieces go into stage s % 3, one per MFMA group;
    //   block 1: column block 1, then bq[par][1] is reloaded for step s + 2.
    // With the barrier at the head of the step instead (first version), each step began with wait -> barrier -> LDS reads
    // -> first MFMA in all eight waves at once and an idle matrix pipe: an ablated kernel with no loads, no MFMAs and no
    // LDS reads still took 0.5 us per step, which added to the 0.35 us of MFMA work instead of hiding under it.
    // The counted wait: program order of this wave's memory operations is
    //   ... W0(s+1), A(s+2), W1(s+1) | W0(s+2), <wait of step s>, A(s+3), W1(s+2) | ...
    // and what must have landed is A(s+1), W0(s+1), W1(s) -- the youngest of them W0(s+1), with A(s+2), W1(s+1), W0(s+2)
    // behind it: at most my_pieces + 4 outstanding (vmcnt counts loads and LDS-DMA alike and retires them in order).
    // Every step issues its loads unconditionally -- past the end of the launch they wrap around to operands that exist and
    // land where nothing reads them -- so that the loop body has no tail cases: one counted wait, no branches but the
    // my_pieces one.  (Peeling a branch-free steady state off a general tail instead made the register allocator give the
    // two copies different registers for bq and move the buffer between them with a load in flight.)
    auto step = [&](auto par_tag) {
      constexpr int par = decltype(par_tag)::value;
      const bool more_a = !JV_ABLATE(p, 1) && !JV_ABLATE(p, 32), more_w = !JV_ABLATE(p, 1) && !JV_ABLATE(p, 16);
      auto group = [&](auto mtag, auto nttag) {
        constexpr int mt = decltype(mtag)::value, nt = decltype(nttag)::value;
        if (!JV_ABLATE(p, 2)) {
          rg_f32x4 t = acc[mt][nt];
          auto mm = [&](const rg_u32x4& x, const rg_u32x4& y) {
            t = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(rg_f16x8, x), __builtin_bit_cast(rg_f16x8, y), t, 0, 0, 0);
          };
          mm(af[par][mt][1], bq[par][nt][0]);      // smallest terms first, as the tile kernels do
          mm(af[par][mt][0], bq[par][nt][1]);
          mm(af[par][mt][0], bq[par][nt][0]);
          acc[mt][nt] = t;
        }
        if constexpr (nt == 1 && mt < PPW) {
          if (more_a) issue_piece(std::integral_constant<int, (mt < PPW ? mt : 0)>{}, s3);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      auto block = [&](auto nttag) {
        group(std::integral_constant<int, 0>{}, nttag);
        if constexpr (RT > 1) group(std::integral_constant<int, 1>{}, nttag);
        if constexpr (RT > 2) group(std::integral_constant<int, 2>{}, nttag);
        if constexpr (RT > 3) group(std::integral_constant<int, 3>{}, nttag);
        if constexpr (RT > 4) group(std::integral_constant<int, 4>{}, nttag);
        if (more_w) load_w(par_tag, nttag);
        __builtin_amdgcn_sched_barrier(0);
      };
      block(std::integral_constant<int, 0>{});
      // (the first step after a chunk's epilogue: what it needs was waited for BEFORE the epilogue's stores were issued,
      // below -- counting here would wait for those stores, vmcnt being in order)
      if (!waited) {
        if (my_pieces == PPW) rg_wait_vmcnt<NWL + PPW>();
        else rg_wait_vmcnt<NWL + PPW - 1>();
      }
      waited = false;
      landed_w(bq[par][1][0], bq[par][1][1], bq[par ^ 1][0][0], bq[par ^ 1][0][1]);
      if (!JV_ABLATE(p, 4)) rg_barrier();
      const int s3n = s3 == 2 ? 0 : s3 + 1;      // (s + 1) % 3
      if (!JV_ABLATE(p, 8)) read_a(std::integral_constant<int, par ^ 1>{}, s3n);
      __builtin_amdgcn_sched_barrier(0);
      block(std::integral_constant<int, 1>{});
      if (more_a) advance();
      if (more_w) advance_w();
      ++s;
      s3 = s3n;
    };
    for (int ks = 0; ks < KS; ks += 2) {
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
    }
    if (c + 1 < NC) {
      // the next step's middle needs A(s + 1), W0(s + 1), W1(s); behind the youngest of them, W0(s + 1), only A(s + 2) and
      // W1(s + 1) have been issued so far
      if (my_pieces == PPW) rg_wait_vmcnt<2 + PPW>();
      else rg_wait_vmcnt<2 + PPW - 1>();
      waited = true;
    }

    // residual epilogues: the rows' residual values, tracking slots and masks are requested BEFORE the accumulators go
    // through the slab, so their latency (the residual was written by the previous launch: L2 or further) hides behind the
    // slab pass and its barrier instead of stalling each half of the row pass
    constexpr int NRW = 2 * RT;      // rows per wave
    constexpr bool RESID = EPI == RG_RES || EPI == RG_RES_LN;
    rg_f32x4 rpre[RESID ? NRW : 1];
    unsigned seenpre[RESID ? NRW : 1];
    bool trkpre[RESID ? NRW : 1];
    if constexpr (RESID) {
#pragma unroll
      for (int j = 0; j < NRW; ++j) {
        // rows past M are clamped to the last one (their results are never stored or tracked): unconditional loads, issued
        // back to back -- behind a per-row branch the compiler waited for each before requesting the next
        const long mr0 = (long)m0 + wave * NRW + j;
        const long mr = mr0 < p.M ? mr0 : (long)p.M - 1;
        rpre[j] = *reinterpret_cast<const rg_f32x4*>(p.res + mr * p.ldr + 4 * lane);
      }
#pragma unroll
      for (int j = 0; j < NRW; ++j) {
        trkpre[j] = false;
        seenpre[j] = 0xffffffffu;
      }
      if (p.amax_out) {      // masks and slots of all rows first, then the slots' current maxima: two latencies, not 2 NRW
        int slot[NRW];
#pragma unroll
        for (int j = 0; j < NRW; ++j) {
          const long mr0 = (long)m0 + wave * NRW + j;
          const long mr = mr0 < p.M ? mr0 : (long)p.M - 1;
          trkpre[j] = mr0 < p.M && (!p.row_mask || p.row_mask[mr] != 0);
          slot[j] = p.row_slot ? p.row_slot[mr] : 0;
        }
        // (plain, cacheable loads: a slot only grows, so a stale value is a valid lower bound)
#pragma unroll
        for (int j = 0; j < NRW; ++j) seenpre[j] = *reinterpret_cast<const unsigned*>(p.amax_out + slot[j]);
      }
    }
    auto rows_body = [&](auto rpw_tag, const float* sl, const int trow0, const int srow0, auto pre_tag) {
      constexpr int RPW = decltype(rpw_tag)::value;
      constexpr int PRE = decltype(pre_tag)::value;      // first index into rpre / seenpre / trkpre
      rg_f32x4 v[RPW];
      long mrow[RPW];
      bool ok[RPW];
#pragma unroll
      for (int j = 0; j < RPW; ++j) {
        const int trow = trow0 + j;
        mrow[j] = (long)m0 + trow;
        ok[j] = trow < R && mrow[j] < p.M;
        v[j] = *reinterpret_cast<const rg_f32x4*>(sl + (srow0 + j) * RG_SLD + 4 * lane) * cs4 + b4;
      }
      if constexpr (EPI == RG_PLAIN) {
#pragma unroll
        for (int j = 0; j < RPW; ++j)
          if (ok[j]) *reinterpret_cast<rg_f32x4*>(p.out + mrow[j] * p.ldo + n0) = v[j];
      } else if constexpr (EPI == RG_QKV) {
        if (c < 2) {      // q (chunk-uniform branch)
#pragma unroll
          for (int j = 0; j < RPW; ++j)
            if (ok[j]) *reinterpret_cast<rg_f32x4*>(p.out + mrow[j] * p.ldo + n0) = v[j];
        } else {
          const float sc = c < 4 ? p.out2_scale : p.out2_scale2;
#pragma unroll
          for (int j = 0; j < RPW; ++j) {
            if (!ok[j]) continue;
            const Split2 s0 = split2h_pair(v[j][0] * sc, v[j][1] * sc);
            const Split2 s1 = split2h_pair(v[j][2] * sc, v[j][3] * sc);
            unsigned short* o2 = p.out2 + mrow[j] * p.ldo2 + (n0 - 512);
            *reinterpret_cast<rg_u32x2*>(o2) = rg_u32x2{s0.h, s1.h};
            *reinterpret_cast<rg_u32x2*>(o2 + p.out2_plane) = rg_u32x2{s0.l, s1.l};
          }
        }
      } else if constexpr (EPI == RG_GELU_PL) {
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
          if (!ok[j]) continue;
          rg_f32x4 t = v[j];
#pragma unroll
          for (int e = 0; e < 4; ++e) t[e] = gelu_erf(t[e]);
          const Split2 s0 = split2h_pair(t[0] * p.out2_scale, t[1] * p.out2_scale);
          const Split2 s1 = split2h_pair(t[2] * p.out2_scale, t[3] * p.out2_scale);
          unsigned short* o2 = p.out2 + mrow[j] * p.ldo2 + n0;
          *reinterpret_cast<rg_u32x2*>(o2) = rg_u32x2{s0.h, s1.h};
          *reinterpret_cast<rg_u32x2*>(o2 + p.out2_plane) = rg_u32x2{s0.l, s1.l};
        }
      } else {
        // + residual -> fp32 rows (+ tracking) (-> LayerNorm -> planes)
        rg_f32x4 r[RPW];
        unsigned seen[RPW];
        bool tracked[RPW];
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
          r[j] = rpre[RESID ? PRE + j : 0];
          tracked[j] = trkpre[RESID ? PRE + j : 0];
          seen[j] = seenpre[RESID ? PRE + j : 0];
        }
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
          v[j] += r[j];
          if (ok[j]) *reinterpret_cast<rg_f32x4*>(p.out + mrow[j] * p.ldo + 4 * lane) = v[j];
        }
        if (p.amax_out) {
#pragma unroll
          for (int j = 0; j < RPW; ++j) {
            unsigned u = 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e) u = max(u, __float_as_uint(v[j][e]) & 0x7fffffffu);
            // wave-uniform branch (a wave holds whole rows): nothing to do once the slot holds a larger value
            if (tracked[j] && __builtin_amdgcn_ballot_w64(u > seen[j]) != 0) {
#pragma unroll
              for (int o = 32; o > 0; o >>= 1) u = max(u, (unsigned)__shfl_xor((int)u, o));
              if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(p.amax_out + (p.row_slot ? p.row_slot[mrow[j]] : 0)), u);
            }
          }
        }
        if constexpr (EPI == RG_RES_LN) {
          // LayerNorm over the row's 256 channels (two-pass, as rowops.hip's layernorm256_kernel), written as the next
          // GEMM's pre-split operand
          const rg_f32x4 gg = *reinterpret_cast<const rg_f32x4*>(p.ln_g + 4 * lane);
          const rg_f32x4 bb = *reinterpret_cast<const rg_f32x4*>(p.ln_b + 4 * lane);
          float sum[RPW], sq[RPW];
#pragma unroll
          for (int j = 0; j < RPW; ++j) sum[j] = wave_sum((v[j][0] + v[j][1]) + (v[j][2] + v[j][3]));
#pragma unroll
          for (int j = 0; j < RPW; ++j) {
            const rg_f32x4 d = v[j] - sum[j] * (1.f / 256.f);
            sq[j] = wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]));
          }
#pragma unroll
          for (int j = 0; j < RPW; ++j) {
            if (!ok[j]) continue;
            const float mean = sum[j] * (1.f / 256.f);
            const float rstd = 1.0f / sqrtf(sq[j] * (1.f / 256.f) + p.ln_eps);
            const rg_f32x4 y = (v[j] - mean) * rstd * gg + bb;
            const Split2 s0 = split2h_pair(y[0] * p.out2_scale, y[1] * p.out2_scale);
            const Split2 s1 = split2h_pair(y[2] * p.out2_scale, y[3] * p.out2_scale);
            unsigned short* o2 = p.out2 + mrow[j] * p.ldo2 + 4 * lane;
            *reinterpret_cast<rg_u32x2*>(o2) = rg_u32x2{s0.h, s1.h};
            *reinterpret_cast<rg_u32x2*>(o2 + p.out2_plane) = rg_u32x2{s0.l, s1.l};
          }
        }
      }
    };

    // the whole tile through its own slab, one pass: every wave takes its 2 RT rows RT at a time.  No barrier after the
    // row pass: the slab is next written a whole main loop (>= 2 barriers) later.
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) big[(mt * 16 + kq * 4 + e) * RG_SLD + wave * 32 + nt * 16 + r16] = acc[mt][nt][e];
    rg_lds_barrier();
    if (!JV_ABLATE(p, 64)) {
      rows_body(std::integral_constant<int, RT>{}, big, wave * 2 * RT, wave * 2 * RT, std::integral_constant<int, 0>{});
      rows_body(std::integral_constant<int, RT>{}, big, wave * 2 * RT + RT, wave * 2 * RT + RT, std::integral_constant<int, RT>{});
    }
  }
  // The wrapped-around loads of the last two steps land after the loop: no LDS-DMA may outlive the workgroup, and bq must
  // stay RESERVED until they have landed -- the compiler does not know about the asm loads in flight, and once it saw bq
  // dead after the last step (the single-chunk residual epilogues) it kept a pointer there, which such a load then
  // overwrote (a memory fault).  The empty asm below makes every bq register live up to this point.
  rg_wait_vmcnt<0>();
  landed_w(bq[0][0][0], bq[0][0][1], bq[0][1][0], bq[0][1][1]);
  landed_w(bq[1][0][0], bq[1][0][1], bq[1][1][0], bq[1][1][1]);
  asm volatile("" ::"v"(warm));
}

// ---- W-direct with the A tile RESIDENT in LDS: the multi-chunk launches (q|k|v and ff.net.0: K = 256, N = 1536 / 1024) ------
// rowgemm_wd_kernel streams A through its 3-stage ring once per 256-column chunk: the q|k|v launch fetched 114 MB where
// its operands are 32 (PMC FETCH_SIZE: the A slices of the ~30 workgroups of an XCD, 2.4 MB, do not survive in a 4 MB L2
// next to the 120 MB the launch writes).  With K = 256 the whole A tile is 8 steps x 10 KB = 80 KB: it is loaded ONCE,
// in the prologue, and every chunk reads it in place.  What that removes besides the traffic:
//   * the A DMA and its counted waits from the main loop (what is waited for is W alone: vmcnt(NWL));
//   * the workgroup barrier from the main loop -- it existed to publish DMA'd stages and to protect the ring -- so the
//     eight waves drift apart and one wave's fragment reads / address arithmetic / chunk epilogue run under another's MFMAs.
// LDS: 80 KB of A + eight 16 x 36-float patches (one per wave, 18 KB) through which each wave transposes its own 32 columns
// of a chunk for the row-wise stores: no barrier in the chunk epilogues either -- the prologue's is the only one.
template <int RT> constexpr int rgwa_lds_bytes(int KS) { return KS * rgw_stage_bytes<RT>() + 8 * 16 * 36 * 4; }

template <int RT, int EPI>
__global__ __launch_bounds__(512, 2) void rowgemm_wa_kernel(const RowGemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rg_lds[];
  constexpr int R = 16 * RT;
  constexpr int STAGE = rgw_stage_bytes<RT>();
  constexpr int A_PLANE = R * 64;
  constexpr int NPIECE = 2 * RT;
  constexpr int PPW = (NPIECE + 7) / 8;
  constexpr int NWL = 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.x * R;
  const int KS = p.K >> 5, NC = p.N >> 8;
  // Column split (RowGemmArgs::nsplit, grid.y): this workgroup owns the chunks [c0, c1) of its rows.  A batch of 3 - 10
  // utterances makes fewer row tiles than the chip has CUs and every workgroup streams the whole weight matrix through its
  // CU's 64 B / clk L2 port -- which, not the matrix pipe, then sets the launch's length (at RT = 2 a step's 32 KB of weights
  // are 512 cycles against 384 of MFMA issue).  The chunks are independent given the rows' operand planes: dealt to the
  // idle CUs, each workgroup streams its share only.  Same K order per output: the same bits.
  const int c0 = (int)blockIdx.y * NC / (int)gridDim.y, c1 = ((int)blockIdx.y + 1) * NC / (int)gridDim.y;
  float* const slab = reinterpret_cast<float*>(rg_lds + KS * STAGE);

  float warm = 0.f;      // L2 warm-up of the weights (rowgemm_kernel)
  if (gridDim.y == 1) {
    const long lpp = ((long)p.N * p.K * 2) >> 7;
    const int grp = blockIdx.x >> 3, ngrp = (gridDim.x + 7) >> 3;
    const long per = (2 * lpp + ngrp - 1) / ngrp;
    const long l = (long)grp * per + tid;
    if (tid < per && l < 2 * lpp) {
      const int pl = l >= lpp;
      warm = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.Wf + (long)pl * p.wf_plane) + ((l - pl * lpp) << 7));
    }
  }

  // ---- the whole A tile, once: step ks of every chunk reads stage ks ----
  {
    const int prow = lane >> 2, pslot = (lane & 3) ^ rg_key(prow);
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int pc = wave + 8 * i;
      if (pc < NPIECE) {      // wave-uniform
        const int pl = pc / RT, g = pc % RT;
        long row = (long)m0 + g * 16 + prow;
        row = row < p.a_rows ? row : p.a_rows - 1;
        const unsigned short* src = p.A2 + (long)pl * p.a2_plane + row * p.lda2 + 8 * pslot;
        const int dst = pl * A_PLANE + g * 1024;
        for (int ks = 0; ks < KS; ++ks)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 32 * ks),
                                           (__attribute__((address_space(3))) void*)(rg_lds + ks * STAGE + dst), 16, 0, 0);
      }
    }
  }

  // ---- W: as in rowgemm_wd_kernel (fragment order, register double buffer loaded by inline asm, counted waits) ----
  const unsigned short* wbase[2][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) wbase[nt][pl] = p.Wf + (long)pl * p.wf_plane + (long)(wave * 2 + nt) * 512 + lane * 8;
  long woff = 16L * 512 * c0;      // (chunk c, step ks) sits (ks N/16 + 16 c) KB into a plane
  int wk = 0, wc = c0;
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
  const long wstep = 512L * (p.N >> 4);      // a k-step on: N / 16 column blocks x 1 KB per plane
  auto advance_w = [&]() {      // back to this workgroup's first chunk after its last
    const bool wrap = ++wk == KS;
    if (wrap) {
      wk = 0;
      const bool end = ++wc == c1;
      if (end) wc = c0;
      woff += 16L * 512 - (KS - 1) * wstep - (end ? 16L * 512 * (c1 - c0) : 0L);
    } else {
      woff += wstep;
    }
  };
  load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
  advance_w();
  load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
  load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
  advance_w();
  rg_wait_vmcnt<0>();
  landed_w(bq[0][0][0], bq[0][0][1], bq[0][1][0], bq[0][1][1]);
  landed_w(bq[1][0][0], bq[1][0][1], bq[1][1][0], bq[1][1][1]);
  rg_barrier();      // every wave's A pieces are in LDS: the only barrier ahead of the epilogues

  const int fslot = (kq ^ rg_key(r16)) << 4;
  const int a_off = r16 * 64 + fslot;
  rg_f32x4 acc[RT][2];
  rg_u32x4 af[2][RT][2];
  auto read_a = [&](auto par_tag, int stage) {
    constexpr int par = decltype(par_tag)::value;
    const unsigned char* const st = rg_lds + stage * STAGE;
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) af[par][mt][pl] = *reinterpret_cast<const rg_u32x4*>(st + a_off + pl * A_PLANE + mt * 1024);
  };
  read_a(std::integral_constant<int, 0>{}, 0);

  int ks1 = 1;      // (step + 1) % KS: the stage the next step reads
  bool waited = false;
  for (int c = c0; c < c1; ++c) {
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = rg_f32x4{0.f, 0.f, 0.f, 0.f};

    // One step: block 0, reload bq[par][0] for step s + 2, counted wait, fragments of step s + 1 requested, block 1, reload
    // bq[par][1].  This wave's memory operations in program order: ... W0(s+1), W1(s+1) | W0(s+2), <wait>, W1(s+2) | ...;
    // needed at the wait: W0(s+1) and W1(s); behind W0(s+1): W1(s+1) and W0(s+2) = NWL loads.  (The chunk epilogue's loads
    // and stores only add younger operations: the wait is then stricter than needed, never weaker.)  Past the end of the
    // launch the loads wrap around to weights that exist (rowgemm_wd_kernel).
    auto step = [&](auto par_tag) {
      constexpr int par = decltype(par_tag)::value;
      auto block = [&](auto nttag) {
        constexpr int nt = decltype(nttag)::value;
#pragma unroll
        for (int mt = 0; mt < RT; ++mt) {
          rg_f32x4 t = acc[mt][nt];
          auto mm = [&](const rg_u32x4& x, const rg_u32x4& y) {
            t = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(rg_f16x8, x), __builtin_bit_cast(rg_f16x8, y), t, 0, 0, 0);
          };
          mm(af[par][mt][1], bq[par][nt][0]);      // smallest terms first, as the tile kernels do
          mm(af[par][mt][0], bq[par][nt][1]);
          mm(af[par][mt][0], bq[par][nt][0]);
          acc[mt][nt] = t;
        }
        __builtin_amdgcn_sched_barrier(0);
        load_w(par_tag, nttag);
        __builtin_amdgcn_sched_barrier(0);
      };
      block(std::integral_constant<int, 0>{});
      if (!waited) rg_wait_vmcnt<NWL>();      // (first step of a chunk: waited for ahead of the epilogue's stores, below)
      waited = false;
      landed_w(bq[par][1][0], bq[par][1][1], bq[par ^ 1][0][0], bq[par ^ 1][0][1]);
      read_a(std::integral_constant<int, par ^ 1>{}, ks1);
      __builtin_amdgcn_sched_barrier(0);
      block(std::integral_constant<int, 1>{});
      advance_w();
      ks1 = ks1 + 1 == KS ? 0 : ks1 + 1;
    };
    for (int ks = 0; ks < KS; ks += 2) {
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
    }
    if (c + 1 < c1) {      // what the next step's wait needs -- W0(s + 1), W1(s) -- has only W1(s + 1) behind it so far
      rg_wait_vmcnt<2>();
      waited = true;
    }

    // ---- the chunk's epilogue, per wave: no workgroup barrier, so one wave's stores and GELU run under the others' MFMAs.
    // A wave owns 32 columns of the chunk; 16 rows at a time go through its private 16 x 36-float patch of LDS (MFMA layout
    // in: lane = column, 4 rows; rows out: 8 lanes x 16 B per row), so that every store instruction writes whole 128-byte
    // (fp32) / 64-byte (fp16 plane) row segments.  LDS executes one wave's accesses in order: its reads see its writes.
    {
      float* const ws = slab + wave * (16 * 36);
      const int prow = lane >> 3, pc4 = (lane & 7) * 4;      // this lane's row of an 8-row pass, its 4 columns of the 32
      const int nw = c * 256 + wave * 32 + pc4;
      rg_f32x4 cw = *reinterpret_cast<const rg_f32x4*>(p.colscale + nw);
      cw = cw * (1.0f / p.a_scale);
      rg_f32x4 bw = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) bw = *reinterpret_cast<const rg_f32x4*>(p.bias + nw);
#pragma unroll
      for (int mt = 0; mt < RT; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int e = 0; e < 4; ++e) ws[(kq * 4 + e) * 36 + nt * 16 + r16] = acc[mt][nt][e];
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          const int trow = mt * 16 + ps * 8 + prow;
          const long mrow = (long)m0 + trow;
          rg_f32x4 v = *reinterpret_cast<const rg_f32x4*>(ws + (ps * 8 + prow) * 36 + pc4) * cw + bw;
          if (mrow >= p.M) continue;
          if (EPI == RG_PLAIN || (EPI == RG_QKV && c < 2)) {
            *reinterpret_cast<rg_f32x4*>(p.out + mrow * p.ldo + nw) = v;
          } else {
            const float sc = EPI == RG_GELU_PL ? p.out2_scale : (c < 4 ? p.out2_scale : p.out2_scale2);
            if constexpr (EPI == RG_GELU_PL) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
            }
            const Split2 s0 = split2h_pair(v[0] * sc, v[1] * sc);
            const Split2 s1 = split2h_pair(v[2] * sc, v[3] * sc);
            unsigned short* o2 = p.out2 + mrow * p.ldo2 + (EPI == RG_GELU_PL ? nw : nw - 512);
            *reinterpret_cast<rg_u32x2*>(o2) = rg_u32x2{s0.h, s1.h};
            *reinterpret_cast<rg_u32x2*>(o2 + p.out2_plane) = rg_u32x2{s0.l, s1.l};
          }
        }
      }
    }
  }
  // the wrapped-around W loads of the last two steps: bq stays reserved until they have landed (rowgemm_wd_kernel)
  rg_wait_vmcnt<0>();
  landed_w(bq[0][0][0], bq[0][0][1], bq[0][1][0], bq[0][1][1]);
  landed_w(bq[1][0][0], bq[1][0][1], bq[1][1][0], bq[1][1][1]);
  asm volatile("" ::"v"(warm));
}

// ---- the feed-forward pair in ONE launch: out = h + ff.net.2(gelu(ff.net.0(x))) (+ the next LayerNorm) ----------------------
// (jyutvoice/flow/transformer.py:425-443: FeedForward = Linear(256, 1024) -> GELU -> Linear(1024, 256), then the residual)
// Unfused, ff.net.0 writes its 80 x 1024 hidden tile as planes (80 MB per launch over the batch) and ff.net.2 streams it
// back through an LDS ring: 160 MB of HBM traffic, one launch ramp and one tail per block more than the arithmetic needs.
// With the weights out of LDS (W-direct) the tile's two operands fit it exactly: the LayerNorm planes of the 16 RT rows
// (K = 256: 8 stages, 80 KB at RT = 5) stay resident, and the hidden tile is produced 256 columns at a time straight into
// a second resident operand image (another 8 stages): 160 KB.  Per hidden chunk c:
//   phase 1   acc1 = X W1[256 c ..][:]^T             8 steps, A = X (resident), wave w owns hidden columns 32 w .. + 32
//   GELU      acc1 -> * colscale / a_scale + bias -> exact GELU -> * h_scale -> fp16 planes, written from the MFMA layout
//             into stage w of the H image (the wave's 32 columns ARE k-step w of phase 2); barrier
//   phase 2   acc2 += H W2[:, 256 c ..]^T            8 steps, A = H, wave w owns output columns 32 w .. + 32; barrier
// then ff.net.2's residual (+ LayerNorm) epilogue through a slab laid over both images.  The W fragments of both linears
// arrive through ONE register double buffer (the step sequence is phase 1, phase 2, phase 1, ... and a step's loads only need
// an address), waits are counted on W alone, and there is no barrier inside a phase.  K order of every sum is the unfused
// kernels': bits identical to ff.net.0 -> planes -> ff.net.2 (tests/test_gpu_pipeline.py compares the two paths with torch.equal).
struct RowFfnArgs {
  const unsigned short* A2;      // LayerNorm planes [2][a_rows][lda2] of x * a_scale1 (K = 256)
  long a2_plane, a_rows;
  int lda2, M;
  const unsigned short* W1f;     // ff.net.0 (N = 1024, K = 256) in fragment order, plane stride w1f_plane halves
  long w1f_plane;
  const float *cs1, *b1;
  float a_scale1, h_scale;       // h_scale: the power of two the hidden planes are scaled with (= ff.net.2's a_scale)
  const unsigned short* W2f;     // ff.net.2 (N = 256, K = 1024) in fragment order
  long w2f_plane;
  const float *cs2, *b2;
  float* out;                    // fp32 rows
  long ldo;
  const float* res;
  long ldr;
  unsigned short* out2;          // LN != 0: LayerNorm planes [2][rows][ldo2] of the stored row * out2_scale
  long out2_plane;
  int ldo2, ln;
  float out2_scale;
  const float *ln_g, *ln_b;
  float ln_eps;
  float* amax_out;
  const int* row_slot;
  const unsigned char* row_mask;
  long alg_rows;
};

template <int RT> constexpr int rgf_lds_bytes() { return 16 * rgw_stage_bytes<RT>() > 16 * RT * RG_SLD * 4 ? 16 * rgw_stage_bytes<RT>() : 16 * RT * RG_SLD * 4; }

template <int RT>
__global__ __launch_bounds__(512, 2) void rowffn_kernel(const RowFfnArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rg_lds[];
  constexpr int R = 16 * RT;
  constexpr int STAGE = rgw_stage_bytes<RT>();
  constexpr int A_PLANE = R * 64;
  constexpr int NPIECE = 2 * RT;
  constexpr int PPW = (NPIECE + 7) / 8;
  constexpr int NWL = 4;
  constexpr int KS = 8, NCH = 4;      // K = 256 per phase, 4 hidden chunks of 256
  constexpr int H_OFF = KS * STAGE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.x * R;

  float warm = 0.f;      // L2 warm-up of both weight matrices (2 planes x 512 KB each: 16384 lines)
  {
    const long lines = 16384;
    const int grp = blockIdx.x >> 3, ngrp = (gridDim.x + 7) >> 3;
    const long per = (lines + ngrp - 1) / ngrp;
    const long l = (long)grp * per + tid;
    if (tid < per && l < lines) {
      const int which = (int)(l >> 13), pl = (int)(l >> 12) & 1;      // 4096 lines per plane
      const unsigned short* base = which ? p.W2f + (long)pl * p.w2f_plane : p.W1f + (long)pl * p.w1f_plane;
      warm = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + ((l & 4095) << 7));
    }
  }

  // ---- X: the whole A tile, once ----
  {
    const int prow = lane >> 2, pslot = (lane & 3) ^ rg_key(prow);
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int pc = wave + 8 * i;
      if (pc < NPIECE) {
        const int pl = pc / RT, g = pc % RT;
        long row = (long)m0 + g * 16 + prow;
        row = row < p.a_rows ? row : p.a_rows - 1;
        const unsigned short* src = p.A2 + (long)pl * p.a2_plane + row * p.lda2 + 8 * pslot;
        const int dst = pl * A_PLANE + g * 1024;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 32 * ks),
                                           (__attribute__((address_space(3))) void*)(rg_lds + ks * STAGE + dst), 16, 0, 0);
      }
    }
  }

  // ---- W of both linears through one register double buffer ----
  const unsigned short* w1base[2][2];
  const unsigned short* w2base[2][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      w1base[nt][pl] = p.W1f + (long)pl * p.w1f_plane + (long)(wave * 2 + nt) * 512 + lane * 8;
      w2base[nt][pl] = p.W2f + (long)pl * p.w2f_plane + (long)(wave * 2 + nt) * 512 + lane * 8;
    }
  long off1 = 0, off2 = 0;      // halves: where the NEXT step to load sits in W1f / W2f
  int wph = 0, wk = 0, wcc = 0; // its phase, k-step, chunk
  rg_u32x4 bq[2][2][2];
#pragma unroll
  for (int i = 0; i < 8; ++i) bq[i >> 2][(i >> 1) & 1][i & 1] = rg_u32x4{0u, 0u, 0u, 0u};
  auto load_frag = [](rg_u32x4& dst, const unsigned short* ptr) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(dst) : "v"(ptr) : "memory");
  };
  auto load_w = [&](auto par_tag, auto nttag) {
    constexpr int par = decltype(par_tag)::value, nt = decltype(nttag)::value;
    load_frag(bq[par][nt][0], wph ? w2base[nt][0] + off2 : w1base[nt][0] + off1);
    load_frag(bq[par][nt][1], wph ? w2base[nt][1] + off2 : w1base[nt][1] + off1);
  };
  auto landed_w = [](rg_u32x4& b00, rg_u32x4& b01, rg_u32x4& b10, rg_u32x4& b11) {
    asm volatile("" : "+v"(b00), "+v"(b01), "+v"(b10), "+v"(b11)::"memory");
  };
  auto advance_w = [&]() {
    // k-step major fragments: W1f (64 column blocks): block 16 c + 2 wave + nt, k-step ks -> (64 ks + 16 c) 512 halves past the
    // base; W2f (16 blocks): k-step 8 c + ks -> 16 (8 c + ks) 512
    if (wph == 0) off1 += 64L * 512; else off2 += 16L * 512;
    if (++wk == KS) {
      wk = 0;
      if (wph == 0) {
        off1 += (16L - 8L * 64) * 512;      // next chunk's blocks: + 16 blocks, less the 8 steps just walked
        wph = 1;
      } else {
        wph = 0;
        if (++wcc == NCH) { wcc = 0; off1 = 0; off2 = 0; }      // past the end: wrap around to weights that exist
      }
    }
  };
  load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
  advance_w();
  load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
  load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
  advance_w();
  rg_wait_vmcnt<0>();
  landed_w(bq[0][0][0], bq[0][0][1], bq[0][1][0], bq[0][1][1]);
  landed_w(bq[1][0][0], bq[1][0][1], bq[1][1][0], bq[1][1][1]);
  rg_barrier();      // X is in LDS for every wave

  const int fslot = (kq ^ rg_key(r16)) << 4;
  const int a_off = r16 * 64 + fslot;
  rg_f32x4 acc1[RT][2], acc2[RT][2];
#pragma unroll
  for (int mt = 0; mt < RT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc2[mt][nt] = rg_f32x4{0.f, 0.f, 0.f, 0.f};

  // one 32-deep step of either phase: block 0, reload, counted wait (W alone: see rowgemm_wa_kernel), the NEXT step's A
  // fragments requested into the other register set (their LDS latency hides behind block 1), block 1, reload.  `have`: this
  // step's fragments were requested by the previous step; `st_next` = null where the next step's operand does not exist
  // yet (the last step of phase 1: H is only complete behind the barrier)
  rg_u32x4 af[2][RT][2];
  auto read_a = [&](auto par_tag, const unsigned char* st) {
    constexpr int par = decltype(par_tag)::value;
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) af[par][mt][pl] = *reinterpret_cast<const rg_u32x4*>(st + a_off + pl * A_PLANE + mt * 1024);
  };
  auto step = [&](auto par_tag, rg_f32x4 (&acc)[RT][2], const unsigned char* st, const unsigned char* st_next, const bool have) {
    constexpr int par = decltype(par_tag)::value;
    if (!have) read_a(par_tag, st);
    __builtin_amdgcn_sched_barrier(0);
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
    block(std::integral_constant<int, 0>{});
    rg_wait_vmcnt<NWL>();
    landed_w(bq[par][1][0], bq[par][1][1], bq[par ^ 1][0][0], bq[par ^ 1][0][1]);
    if (st_next) read_a(std::integral_constant<int, par ^ 1>{}, st_next);
    __builtin_amdgcn_sched_barrier(0);
    block(std::integral_constant<int, 1>{});
    advance_w();
  };

  const float inv1 = 1.0f / p.a_scale1;
  for (int c = 0; c < NCH; ++c) {
    // ---- phase 1: the hidden chunk ----
    const int hc0 = c * 256 + wave * 32 + r16;      // this lane's hidden columns: hc0 (nt = 0), hc0 + 16 (nt = 1)
    float csl[2], bl[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      csl[nt] = p.cs1[hc0 + 16 * nt] * inv1;
      bl[nt] = p.b1 ? p.b1[hc0 + 16 * nt] : 0.f;
    }
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc1[mt][nt] = rg_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int ks = 0; ks < KS; ks += 2) {
      step(std::integral_constant<int, 0>{}, acc1, rg_lds + ks * STAGE, rg_lds + (ks + 1) * STAGE, ks > 0 || c > 0);
      step(std::integral_constant<int, 1>{}, acc1, rg_lds + (ks + 1) * STAGE, ks + 2 < KS ? rg_lds + (ks + 2) * STAGE : nullptr, true);
    }
    // ---- GELU -> planes, into stage `wave` of H (rows mt 16 + 4 kq + e, k = 16 nt + r16 of that stage) ----
    if (c > 0) rg_lds_barrier();      // every wave is done reading the previous chunk's H
    {
      unsigned char* const hs = rg_lds + H_OFF + wave * STAGE;
      const int key = (kq & 1) << 1;      // rg_key(row): bit 2 of the row = bit 0 of kq
#pragma unroll
      for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int slot = (2 * nt + (r16 >> 3)) ^ key;
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            const float g0 = gelu_erf(acc1[mt][nt][e] * csl[nt] + bl[nt]) * p.h_scale;
            const float g1 = gelu_erf(acc1[mt][nt][e + 1] * csl[nt] + bl[nt]) * p.h_scale;
            const Split2 sp = split2h_pair(g0, g1);
            const int row = mt * 16 + kq * 4 + e;
            unsigned char* d = hs + row * 64 + (slot << 4) + (r16 & 7) * 2;
            *reinterpret_cast<unsigned short*>(d) = (unsigned short)(sp.h & 0xffffu);
            *reinterpret_cast<unsigned short*>(d + 64) = (unsigned short)(sp.h >> 16);
            *reinterpret_cast<unsigned short*>(d + A_PLANE) = (unsigned short)(sp.l & 0xffffu);
            *reinterpret_cast<unsigned short*>(d + A_PLANE + 64) = (unsigned short)(sp.l >> 16);
          }
        }
    }
    rg_lds_barrier();      // H complete
    // ---- phase 2 ----
#pragma unroll 1
    for (int ks = 0; ks < KS; ks += 2) {
      step(std::integral_constant<int, 0>{}, acc2, rg_lds + H_OFF + ks * STAGE, rg_lds + H_OFF + (ks + 1) * STAGE, ks > 0);
      // (the last step of the chunk requests the first X stage of the next chunk's phase 1: X is always there)
      step(std::integral_constant<int, 1>{}, acc2, rg_lds + H_OFF + (ks + 1) * STAGE,
           ks + 2 < KS ? rg_lds + H_OFF + (ks + 2) * STAGE : (c + 1 < NCH ? rg_lds : nullptr), true);
    }
  }

  // ---- ff.net.2's epilogue: + bias + residual -> rows (+ tracking) (-> LayerNorm -> planes); rowgemm_wd_kernel's ----
  constexpr int NRW = 2 * RT;
  rg_f32x4 rpre[NRW];
  unsigned seenpre[NRW];
  bool trkpre[NRW];
#pragma unroll
  for (int j = 0; j < NRW; ++j) {
    const long mr0 = (long)m0 + wave * NRW + j;
    const long mr = mr0 < p.M ? mr0 : (long)p.M - 1;
    rpre[j] = *reinterpret_cast<const rg_f32x4*>(p.res + mr * p.ldr + 4 * lane);
    trkpre[j] = false;
    seenpre[j] = 0xffffffffu;
  }
  if (p.amax_out) {
    int slot[NRW];
#pragma unroll
    for (int j = 0; j < NRW; ++j) {
      const long mr0 = (long)m0 + wave * NRW + j;
      const long mr = mr0 < p.M ? mr0 : (long)p.M - 1;
      trkpre[j] = mr0 < p.M && (!p.row_mask || p.row_mask[mr] != 0);
      slot[j] = p.row_slot ? p.row_slot[mr] : 0;
    }
#pragma unroll
    for (int j = 0; j < NRW; ++j) seenpre[j] = *reinterpret_cast<const unsigned*>(p.amax_out + slot[j]);
  }
  rg_f32x4 cs4 = *reinterpret_cast<const rg_f32x4*>(p.cs2 + 4 * lane);
  cs4 = cs4 * (1.0f / p.h_scale);
  rg_f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
  if (p.b2) b4 = *reinterpret_cast<const rg_f32x4*>(p.b2 + 4 * lane);
  rg_f32x4 gg = {1.f, 1.f, 1.f, 1.f}, bb = {0.f, 0.f, 0.f, 0.f};
  if (p.ln) {
    gg = *reinterpret_cast<const rg_f32x4*>(p.ln_g + 4 * lane);
    bb = *reinterpret_cast<const rg_f32x4*>(p.ln_b + 4 * lane);
  }
  rg_lds_barrier();      // every wave is done with X and H: the slab goes over them
  float* const big = reinterpret_cast<float*>(rg_lds);
#pragma unroll
  for (int mt = 0; mt < RT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e) big[(mt * 16 + kq * 4 + e) * RG_SLD + wave * 32 + nt * 16 + r16] = acc2[mt][nt][e];
  rg_lds_barrier();
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    rg_f32x4 v[RT];
    long mrow[RT];
    bool ok[RT];
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      const int trow = wave * NRW + ps * RT + j;
      mrow[j] = (long)m0 + trow;
      ok[j] = mrow[j] < p.M;
      v[j] = *reinterpret_cast<const rg_f32x4*>(big + trow * RG_SLD + 4 * lane) * cs4 + b4;
      v[j] += rpre[ps * RT + j];
      if (ok[j]) *reinterpret_cast<rg_f32x4*>(p.out + mrow[j] * p.ldo + 4 * lane) = v[j];
    }
    if (p.amax_out) {
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        unsigned u = 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) u = max(u, __float_as_uint(v[j][e]) & 0x7fffffffu);
        if (trkpre[ps * RT + j] && __builtin_amdgcn_ballot_w64(u > seenpre[ps * RT + j]) != 0) {
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) u = max(u, (unsigned)__shfl_xor((int)u, o));
          if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(p.amax_out + (p.row_slot ? p.row_slot[mrow[j]] : 0)), u);
        }
      }
    }
    if (p.ln) {
      float sum[RT], sq[RT];
#pragma unroll
      for (int j = 0; j < RT; ++j) sum[j] = wave_sum((v[j][0] + v[j][1]) + (v[j][2] + v[j][3]));
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        const rg_f32x4 d = v[j] - sum[j] * (1.f / 256.f);
        sq[j] = wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]));
      }
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        if (!ok[j]) continue;
        const float mean = sum[j] * (1.f / 256.f);
        const float rstd = 1.0f / sqrtf(sq[j] * (1.f / 256.f) + p.ln_eps);
        const rg_f32x4 y = (v[j] - mean) * rstd * gg + bb;
        const Split2 s0 = split2h_pair(y[0] * p.out2_scale, y[1] * p.out2_scale);
        const Split2 s1 = split2h_pair(y[2] * p.out2_scale, y[3] * p.out2_scale);
        unsigned short* o2 = p.out2 + mrow[j] * p.ldo2 + 4 * lane;
        *reinterpret_cast<rg_u32x2*>(o2) = rg_u32x2{s0.h, s1.h};
        *reinterpret_cast<rg_u32x2*>(o2 + p.out2_plane) = rg_u32x2{s0.l, s1.l};
      }
    }
  }
  rg_wait_vmcnt<0>();      // the wrapped-around W loads: bq stays reserved until they have landed
  landed_w(bq[0][0][0], bq[0][0][1], bq[0][1][0], bq[0][1][1]);
  landed_w(bq[1][0][0], bq[1][0][1], bq[1][1][0], bq[1][1][1]);
  asm volatile("" ::"v"(warm));
}

}  // namespace jv
