// One row-owning launch for the tail of a BasicTransformerBlock and the head of the next one
// (jyutvoice/flow/transformer.py:355-443): on the same 16 RT rows,
//   phase A   h  += attn1.to_out.0(att) + bias                      (K = 512: the attention planes stream through a ring)
//             x   = LayerNorm3(h)                                    -> fp16 planes, written INTO the resident operand image
//   phase B   out = h + ff.net.2(gelu(ff.net.0(x)))                  (rowffn_kernel's loop: hidden tile in LDS)
//             x'  = LayerNorm1_next(out)                             -> planes, into the operand image again          [QKV]
//   phase C   q | k | v = to_q/k/v_next(x')                          (rowgemm_wa_kernel's loop, per-wave epilogues)    [QKV]
// Unfused these are three launches (rowgemm_wd<res,ln>, rowffn, rowgemm_wa<qkv>) that own the same rows: each pays a launch
// ramp and a tail, and the LayerNorm planes (20 MB at 19.5 K rows) travel to HBM and straight back twice per block.  Here
// they never leave LDS.  What stays in HBM between the phases is the residual stream itself: phase A stores h and phase
// B's epilogue reads the same values back (the same lane reads what it wrote) -- at 2 waves per SIMD a wave has 256
// registers and the feed-forward loop uses 242 of them, so the 40 the rows would need do not exist.
//
// LDS (32 RT KB; 160 KB at RT = 5), in STAGE = 2 KB x RT units (one 32-deep k-step of the 16 RT rows as two fp16 planes):
//   [0, 8)    X: the K = 256 operand image (LayerNorm planes), rowgemm_wa_kernel's layout
//   [8, 16)   phase A: the 3-stage ring of the attention planes, then the epilogue slab; phase B: the hidden image H, then
//             the epilogue slab; phase C: the eight per-wave transposition patches
// The slab is the whole tile as fp32, UNPADDED (16 RT rows x 1 KB = 8 stages exactly): instead of rowgemm's 260-float rows
// the 64-byte groups of a row are XOR-ed with bits 2..3 of the row, which keeps the MFMA-layout writes on 64 distinct banks
// and the row pass's 16-byte reads aligned.
//
// The weight fragments of all four linears arrive through ONE register double buffer (rowgemm_wd_kernel explains the
// hand-counted waits): the loader walks a sequence of 8-step segments -- Wo k 0..7, Wo k 8..15, then per hidden chunk
// W1[chunk] and W2[k-steps of the chunk], then the six 256-column chunks of Wqkv -- two steps ahead of the MFMAs, and
// wraps around to the first segment past the end (those loads land in registers nothing reads).
// Arithmetic: every sum in the K order of the separate kernels, the same epilogue expressions -- bit-identical results
// (tests/test_gpu_pipeline.py::test_fused_block_equals_separate_launches).
#pragma once
#include "rowgemm_kernel.h"

namespace jv {

struct RowBlockArgs {
  // ---- phase A: attn1.to_out.0 (N = 256, K = 512) + residual + LayerNorm3
  // Every row buffer must be readable up to row ceil(M / tile) * tile - 1 (the host checks a_rows): rows past M - 1 are read
  // unclamped (their results are never stored or tracked) and the strides are compile-time -- h [., 256], attention planes
  // [., 512], q [., 512], K/V planes [., 1024] -- so that a row's address is a constant offset from the tile's first row
  const unsigned short* A2;      // attention planes [2][a_rows][512] of att * a_scale_o
  long a2_plane, a_rows;
  int M;
  const unsigned short* Wof;     // fragment order, plane stride wof_plane halves
  long wof_plane;
  const float *cso, *bo;
  float a_scale_o, inv_a_scale_o;      // (the reciprocals of the powers of two, from the host: an IEEE division is ten instructions)
  float* h;                      // [rows, 256] fp32: residual in, h + to_out(att) out (in place)
  const float *ln3_g, *ln3_b;
  // ---- phase B: ff.net.0 (N = 1024, K = 256) -> GELU -> ff.net.2 (N = 256, K = 1024) + residual
  const unsigned short* W1f;
  long w1f_plane;
  const float *cs1, *b1;
  float a_scale1, h_scale;       // a_scale1: what LayerNorm3's planes are scaled with; h_scale: the hidden planes
  float inv_a_scale1, inv_h_scale;
  const unsigned short* W2f;
  long w2f_plane;
  const float *cs2, *b2;
  float* out;                    // fp32 rows (h itself, or the skip / concat buffer for a stage's last block)
  long ldo;
  // measured-bound tracking of what phases A and B store (RowGemmArgs::amax_out)
  float *amax_h, *amax_out;
  const int* row_slot;
  const unsigned char* row_mask;
  // ---- phase C (QKV): the next block's norm1 and to_q | to_k | to_v (N = 1536, K = 256)
  const float *ln1_g, *ln1_b;
  const unsigned short* Wqf;
  long wqf_plane;
  const float* csq;
  float a_scale_q, inv_a_scale_q;      // what LayerNorm1's planes are scaled with
  float* q;                      // fp32 rows [., 512]
  unsigned short* kv2;           // planes [2][rows][1024]: k * k_scale in columns 0..511, v * v_scale in 512..1023
  long kv2_plane;
  float k_scale, v_scale;
  // ---- not QKV, a next block exists (mid-size batches, flow.hip `qkv_split`): LayerNorm1_next(out) still runs in phase B's
  // epilogue and its planes go to HBM, [2][rows][256] at plane stride ln_out_plane, for a column-split q|k|v launch
  unsigned short* ln_out;
  long ln_out_plane;
  long alg_rows;
  unsigned long long* stamps;      // tuning aid (JV_RB_STAMPS, tuning builds): [workgroup][48] s_memtime at the phase boundaries
  // tuning aid (JV_RB_ABLATE, tuning builds; results are wrong by design): 1 GELU pass without its arithmetic, 2 no GELU pass at all,
  // 4 phase C without global stores, 8 phase C without chunk epilogues, 16 residual epilogues without their row passes
  int ablate;
};

template <int RT> constexpr int rb_lds_bytes() { return 16 * rgw_stage_bytes<RT>(); }

// element (row, col) of the swizzled slab, in floats
__device__ __forceinline__ int rb_slab(int row, int col) { return row * 256 + (col ^ (((row >> 2) & 3) << 4)); }

template <int RT, bool QKV, bool STAG>
__global__ __launch_bounds__(512, 2) void rowblock_kernel(const RowBlockArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rg_lds[];
  constexpr int R = 16 * RT;
  constexpr int STAGE = rgw_stage_bytes<RT>();
  constexpr int A_PLANE = R * 64;
  constexpr int NPIECE = 2 * RT;                 // 1 KiB DMA pieces per ring stage
  constexpr int PPW = (NPIECE + 7) / 8;
  constexpr int NWL = 4;
  constexpr int KSA = 16, KS = 8, NCH = 4, NCQ = 6;
  constexpr int H_OFF = KS * STAGE;
  constexpr int NQUAD = QKV ? 4 + 4 * NCH + 2 * NCQ : 4 + 4 * NCH;      // 4-step runs of weight fragments (the walker below)
  constexpr int NRW = 2 * RT;                    // rows per wave in the row passes
  constexpr float LN_EPS = 1e-5f;                // nn.LayerNorm's default (transformer.py:355-443 builds its norms with it)
  constexpr long LDH = 256, LDQ = 512, LDKV = 1024;
  static_assert(8 * STAGE == R * 1024, "the unpadded slab is exactly the upper half of LDS");
  constexpr int NPATCH = 8 * 2 * 16 * 36 * 4 <= 8 * STAGE ? 2 : 1;      // transposition patches per wave in phase C (one at RT = 2)
  static_assert(8 * NPATCH * 16 * 36 * 4 <= 8 * STAGE, "the per-wave patches of phase C fit the upper half");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.x * R;
  const unsigned lane4 = 4u * (unsigned)lane;      // this lane's four columns in the row passes
  int stamp_i = 0;
  auto stamp = [&]() {      // (compiled out unless JV_TUNING)
    if (JV_STAMP(p)) {
      // the LAST wave's arrival (wave 0 alone reads fast: the older wave of a SIMD wins the matrix pipe's arbitration)
      if (lane == 0 && stamp_i < 32) atomicMax(&p.stamps[(long)blockIdx.x * 48 + stamp_i], (unsigned long long)__builtin_amdgcn_s_memtime());
      ++stamp_i;
    }
  };
  stamp();      // 0: start
  int xstamp_i = 32;
  auto xstamp = [&]() {      // detail stamps inside the epilogues: slots 32..47
    if (JV_STAMP(p)) {
      if (lane == 0 && xstamp_i < 48) atomicMax(&p.stamps[(long)blockIdx.x * 48 + xstamp_i], (unsigned long long)__builtin_amdgcn_s_memtime());
      ++xstamp_i;
    }
  };
  unsigned char* const hreg = rg_lds + H_OFF;
  float* const slab = reinterpret_cast<float*>(hreg);

  // ---- L2 warm-up of the weights (rowgemm_kernel): the workgroups that share an XCD touch every 128-byte line of a matrix
  // once.  Wo at kernel start; W1 in the middle of phase A; W2 while phase A's epilogue runs; Wqkv at the last GELU pass -- a
  // matrix touched a hundred microseconds ahead would be gone from a 4 MB L2 by the time its phase starts.
  const int wgrp = blockIdx.x >> 3, wngrp = (gridDim.x + 7) >> 3;
  auto warm_lines = [&](const unsigned short* base, long plane_halves, long lines_per_plane, int round) -> float {
    const long per = (2 * lines_per_plane + wngrp - 1) / wngrp;
    const long i = tid + 512L * round;
    const long l = (long)wgrp * per + i;
    float v = 0.f;
    if (i < per && l < 2 * lines_per_plane) {
      const int pl = l >= lines_per_plane;
      // (explicitly GLOBAL loads and stores throughout this kernel: through a lambda's pointer parameter the compiler lost the
      // address space and emitted flat_load / flat_store, which count on lgkmcnt as well -- every LDS wait and every barrier
      // then also waited for the rows' residual loads and stores in flight)
      v = *(const __attribute__((address_space(1))) float*)(reinterpret_cast<const char*>(base + (long)pl * plane_halves) + ((l - pl * lines_per_plane) << 7));
    }
    return v;
  };
  float warm = warm_lines(p.Wof, p.wof_plane, 2048, 0);      // 256 x 512 halves = 256 KB per plane
  float warm1 = 0.f, warm2 = 0.f, warm3 = 0.f;

  // Per-row facts of the tracking (mask, slot, the slots' current maxima): lane j < NRW loads those of the wave's row j and
  // the row passes broadcast them with v_readlane.  Loaded HERE, at kernel start, where the dependent chain (row -> slot ->
  // maximum) hides behind the prologue's own waits: in front of an epilogue it was two L2 round trips with nothing to
  // overlap them (3 - 4.6 us per epilogue in the phase stamps).  A maximum read early is merely stale, i.e. a valid lower
  // bound.  (Asked for row by row these are wave-uniform values: the compiler then made 2 NRW scalar-path loads of them,
  // each behind its own branch and s_waitcnt vmcnt(0).)
  struct RowFacts { int slot; unsigned seen; int trk; };
  RowFacts facts_h{0, 0xffffffffu, 0}, facts_o{0, 0xffffffffu, 0};
  if (p.amax_h || p.amax_out) {
    const long mr0 = (long)m0 + wave * NRW + (lane < NRW ? lane : 0);
    const long mr = mr0 < p.M ? mr0 : (long)p.M - 1;
    const int trk = (mr0 < p.M && (!p.row_mask || p.row_mask[mr] != 0)) ? 1 : 0;
    const int slot = p.row_slot ? p.row_slot[mr] : 0;
    // (plain, cacheable loads: a slot only grows)
    if (p.amax_h) facts_h = RowFacts{slot, *reinterpret_cast<const unsigned*>(p.amax_h + slot), trk};
    if (p.amax_out) facts_o = RowFacts{slot, *reinterpret_cast<const unsigned*>(p.amax_out + slot), trk};
  }

  // ---- phase A operand: the attention planes through a 3-stage ring in the upper half (rowgemm_wd_kernel) ----
  const unsigned short* cur[PPW];
  int dst[PPW];
  {
    const int prow = lane >> 2, pslot = (lane & 3) ^ rg_key(prow);
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int pc = wave + 8 * i;
      const int pl = pc / RT, g = pc % RT;
      const long row = (long)m0 + g * 16 + prow;
      cur[i] = p.A2 + (long)(pl & 1) * p.a2_plane + row * 512 + 8 * pslot;
      dst[i] = (pl & 1) * A_PLANE + g * 1024;
    }
  }
  const int my_pieces = (NPIECE - wave + 7) / 8 > 0 ? (NPIECE - wave + 7) / 8 : 0;      // wave-uniform
  auto issue_piece = [&](auto itag, int stage) {
    constexpr int i = decltype(itag)::value;
    if (wave + 8 * i < NPIECE) {      // wave-uniform
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)cur[i],
                                       (__attribute__((address_space(3))) void*)(hreg + stage * STAGE + dst[i]), 16, 0, 0);
    }
  };
  auto advance_a = [&]() {
#pragma unroll
    for (int i = 0; i < PPW; ++i) cur[i] += 32;
  };

  // ---- W: the walker.  The launch's weight fragments are a sequence of QUADS, runs of four 32-deep steps inside one matrix:
  // Wo k 0..15 (4 quads), then the feed-forward's 16 (W1[chunk] k 0..7 = 2 quads, W2[k-steps of the chunk] = 2 quads, per hidden
  // chunk), then the six 256-column chunks of Wqkv (2 quads each).  Fragment (nt, pl) of step j of a quad sits at
  //   base + pl * plane + (((k0 + j) * NBm + blk0 + 2 wave + nt) * 64 + lane) * 8 halves      (k-step major, NBm = the matrix's N / 16)
  // STAG: waves 0..3 walk the feed-forward's quads in THEIR order of use (the staggered schedule below):
  //   W1[0] | W2[c] first half, W1[c + 1], W2[c] second half (c = 0..2) | W2[3]
  const bool hx = STAG && wave < 4;      // wave-uniform: the leading half of the workgroup
  const unsigned short* wp[2][2];
  int wq = 0, wj = 0, wstep = 16 * 512;      // wstep: halves from one k-step's fragments to the next's = NBm x 512
  auto quad_ptrs = [&](const int q) {      // wave-uniform
    const unsigned short* base;
    long plane;
    int blk0, nbm, k0;
    if (q < 4) {
      base = p.Wof; plane = p.wof_plane; blk0 = 0; nbm = 16; k0 = 4 * q;
    } else if (q < 4 + 4 * NCH) {
      const int f = q - 4;
      int c = f >> 2, h = f & 1;
      bool first = (f & 2) == 0;      // W1 (the chunk's first linear) or W2
      if (hx) {
        if (f < 2) { c = 0; first = true; h = f; }
        else if (f >= 4 * NCH - 2) { c = NCH - 1; first = false; h = f - (4 * NCH - 2); }
        else {
          const int g = f - 2, r = g & 3;
          c = g >> 2;
          if (r == 0) { first = false; h = 0; }
          else if (r == 3) { first = false; h = 1; }
          else { first = true; c += 1; h = r - 1; }
        }
      }
      if (first) { base = p.W1f; plane = p.w1f_plane; blk0 = 16 * c; nbm = 16 * NCH; k0 = 4 * h; }
      else { base = p.W2f; plane = p.w2f_plane; blk0 = 0; nbm = 16; k0 = 8 * c + 4 * h; }
    } else {
      const int g = q - 4 - 4 * NCH;
      base = p.Wqf; plane = p.wqf_plane; blk0 = 16 * (g >> 1); nbm = 16 * NCQ; k0 = 4 * (g & 1);
    }
    wstep = nbm * 512;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        wp[nt][pl] = base + (long)pl * plane + ((long)k0 * nbm + blk0 + 2 * wave + nt) * 512 + lane * 8;
  };
  quad_ptrs(0);
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
  auto advance_w = [&]() {
    if (++wj == 4) {
      wj = 0;
      if (++wq == NQUAD) wq = 0;      // past the end: wrap around to weights that exist
      quad_ptrs(wq);
    } else {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) wp[nt][pl] += wstep;
    }
  };

  const int fslot = (kq ^ rg_key(r16)) << 4;
  const int a_off = r16 * 64 + fslot;
  rg_u32x4 af[2][RT][2];
  auto read_a = [&](auto par_tag, const unsigned char* st) {
    constexpr int par = decltype(par_tag)::value;
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) af[par][mt][pl] = *reinterpret_cast<const rg_u32x4*>(st + a_off + pl * A_PLANE + mt * 1024);
  };
  auto mfma_block = [&](auto par_tag, auto nttag, rg_f32x4 (&acc)[RT][2], auto mt_tag) {
    constexpr int par = decltype(par_tag)::value, nt = decltype(nttag)::value, mt = decltype(mt_tag)::value;
    rg_f32x4 t = acc[mt][nt];
    auto mm = [&](const rg_u32x4& x, const rg_u32x4& y) {
      t = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(rg_f16x8, x), __builtin_bit_cast(rg_f16x8, y), t, 0, 0, 0);
    };
    mm(af[par][mt][1], bq[par][nt][0]);      // smallest terms first, as everywhere
    mm(af[par][mt][0], bq[par][nt][1]);
    mm(af[par][mt][0], bq[par][nt][0]);
    acc[mt][nt] = t;
  };

  // ---- prologue: ring stages 0..2, W of steps 0 and 1; everything waited for once ----
#pragma unroll
  for (int st = 0; st < 3; ++st) {
    issue_piece(std::integral_constant<int, 0>{}, st);
    if constexpr (PPW > 1) issue_piece(std::integral_constant<int, 1>{}, st);
    advance_a();
    if (st < 2) {
      if (st == 0) {
        load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
      } else {
        load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
        load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
      }
      advance_w();
    }
  }
  rg_wait_vmcnt<0>();
  landed_w(bq[0][0][0], bq[0][0][1], bq[0][1][0], bq[0][1][1]);
  landed_w(bq[1][0][0], bq[1][0][1], bq[1][1][0], bq[1][1][1]);
  // A warm-up value is "used" (an empty asm) where the wait costs nothing and is dead from then on.  Kept alive to the end
  // of the kernel it crossed the feed-forward loop -- in scratch: the compiler spilled it right behind the load, i.e. put an
  // s_waitcnt vmcnt(0) on a load that is an L2 miss BY DESIGN in front of phase A's epilogue (6 k cycles in the stamps).
  asm volatile("" ::"v"(warm));
  rg_barrier();
  stamp();      // 1: prologue done
  read_a(std::integral_constant<int, 0>{}, hreg);

  rg_f32x4 acc1[RT][2], acc2[RT][2];
#pragma unroll
  for (int mt = 0; mt < RT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc1[mt][nt] = rg_f32x4{0.f, 0.f, 0.f, 0.f};

  // ================================ phase A: acc1 = att Wo^T (16 steps through the ring) ================================
  // rowgemm_wd_kernel's step.  This wave's memory operations in program order:
  //   ... W0(s+1), A(s+2), W1(s+1) | W0(s+2), <wait of step s>, A(s+3), W1(s+2) | ...
  // needed at the wait: A(s+1), W0(s+1), W1(s); behind W0(s+1): A(s+2) (if it exists), W1(s+1), W0(s+2).  The ring is NOT fed
  // past the end of the phase (its stages become the slab), so the last two steps count the W loads alone.
  {
    int s = 0, s3 = 0;
    auto stepA = [&](auto par_tag) {
      constexpr int par = decltype(par_tag)::value;
      const bool more_a = s + 3 < KSA;
      auto group = [&](auto mtag, auto nttag) {
        constexpr int mt = decltype(mtag)::value, nt = decltype(nttag)::value;
        mfma_block(par_tag, nttag, acc1, mtag);
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
        load_w(par_tag, nttag);
        __builtin_amdgcn_sched_barrier(0);
      };
      block(std::integral_constant<int, 0>{});
      if (s + 2 < KSA) {
        if (my_pieces == PPW) rg_wait_vmcnt<NWL + PPW>();
        else rg_wait_vmcnt<NWL + PPW - 1>();
      } else {
        rg_wait_vmcnt<NWL>();
      }
      landed_w(bq[par][1][0], bq[par][1][1], bq[par ^ 1][0][0], bq[par ^ 1][0][1]);
      rg_barrier();
      const int s3n = s3 == 2 ? 0 : s3 + 1;
      if (s + 1 < KSA) read_a(std::integral_constant<int, par ^ 1>{}, hreg + s3n * STAGE);
      __builtin_amdgcn_sched_barrier(0);
      block(std::integral_constant<int, 1>{});
      if (more_a) advance_a();
      advance_w();
      ++s;
      s3 = s3n;
    };
#pragma unroll 1
    for (int ks = 0; ks < KSA; ks += 2) {
      // W1 is touched HERE, half a phase (~6 us) ahead of its first fragment loads: touched at kernel start it had left the
      // L2 again by then -- 7.5 MB of attention planes and residual rows per XCD pass through in phase A -- and the stamps
      // read 5.9 k cycles between the end of this loop and the epilogue's first instruction, the wait for those fragments
      if (ks == KSA / 2) warm1 = warm_lines(p.W1f, p.w1f_plane, 4096, 0);     // 1024 x 256 halves = 512 KB per plane
      stepA(std::integral_constant<int, 0>{});
      stepA(std::integral_constant<int, 1>{});
    }
  }
  stamp();      // 2: phase A loop done
  // what the first step of phase B waits for in its middle -- W0(17), W1(16) -- has only W1(17) behind it so far; counted
  // HERE, ahead of the epilogue's own loads and stores (vmcnt retires in order: counting behind them would wait for them)
  rg_wait_vmcnt<2>();

  // the row pass shared by the two residual epilogues: rows [wave NRW, + NRW) of the tile, RT at a time, from the slab:
  //   v = slab * cs + bias + res -> dst rows (+ tracking); LN: LayerNorm_256(v) * sc -> fp16 planes into the operand image X
  auto row_pass = [&](const rg_f32x4 cs4, const rg_f32x4 b4, const rg_f32x4 (&rpre)[NRW], const RowFacts facts, float* dstp,
                      const long ldd, float* amax, const bool ln, const rg_f32x4 gg, const rg_f32x4 bb, const float sc,
                      unsigned short* const gplanes) {      // gplanes: the LayerNorm planes to HBM as well (RowBlockArgs::ln_out)
    auto rows = [&](auto ps_tag) {
      constexpr int ps = decltype(ps_tag)::value;
      rg_f32x4 v[RT];
      bool ok[RT];
      int drow0 = m0 + wave * NRW + ps * RT;      // (laundered: see prefetch_rows)
      asm volatile("" : "+s"(drow0));
      float* const dbase = dstp + (long)drow0 * ldd;
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        const int trow = wave * NRW + ps * RT + j;
        ok[j] = m0 + trow < p.M;
        v[j] = *reinterpret_cast<const rg_f32x4*>(slab + rb_slab(trow, 4 * lane)) * cs4 + b4;
        v[j] += rpre[ps * RT + j];
        if (ok[j]) *(__attribute__((address_space(1))) rg_f32x4*)(dbase + j * ldd + lane4) = v[j];
      }
      if (amax) {
#pragma unroll
        for (int j = 0; j < RT; ++j) {
          unsigned u = 0u;
#pragma unroll
          for (int e = 0; e < 4; ++e) u = max(u, __float_as_uint(v[j][e]) & 0x7fffffffu);
          const int trk = __builtin_amdgcn_readlane(facts.trk, ps * RT + j);
          const unsigned seen = (unsigned)__builtin_amdgcn_readlane((int)facts.seen, ps * RT + j);
          if (trk && __builtin_amdgcn_ballot_w64(u > seen) != 0) {      // wave-uniform: nothing to do once the slot holds a larger value
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) u = max(u, (unsigned)__shfl_xor((int)u, o));
            if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(amax + __builtin_amdgcn_readlane(facts.slot, ps * RT + j)), u);
          }
        }
      }
      if (ln) {
        float sum[RT], sq[RT];
#pragma unroll
        for (int j = 0; j < RT; ++j) sum[j] = wave_sum((v[j][0] + v[j][1]) + (v[j][2] + v[j][3]));
#pragma unroll
        for (int j = 0; j < RT; ++j) {
          const rg_f32x4 d = v[j] - sum[j] * (1.f / 256.f);
          sq[j] = wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]));
        }
        // 1 / sqrt(var + eps) of the RT rows in ONE evaluation: lane j computes row j's (the correctly rounded division and
        // square root are ~30 instructions; per row, on wave-uniform values, that was a third of this pass), then broadcast
        float var_l = sq[0];
#pragma unroll
        for (int j = 1; j < RT; ++j) var_l = lane == j ? sq[j] : var_l;
        const float rstd_l = 1.0f / sqrtf(var_l * (1.f / 256.f) + LN_EPS);
#pragma unroll
        for (int j = 0; j < RT; ++j) {
          const int trow = wave * NRW + ps * RT + j;
          const float mean = sum[j] * (1.f / 256.f);
          const float rstd = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rstd_l), j));
          const rg_f32x4 y = (v[j] - mean) * rstd * gg + bb;
          const Split2 s0 = split2h_pair(y[0] * sc, y[1] * sc);
          const Split2 s1 = split2h_pair(y[2] * sc, y[3] * sc);
          // columns 4 lane .. + 3 = k-step lane >> 3, 16-byte slot (lane & 7) >> 1 (swizzled), its half lane & 1
          unsigned char* d = rg_lds + (lane >> 3) * STAGE + trow * 64 + (((((lane & 7) >> 1) ^ rg_key(trow))) << 4) + (lane & 1) * 8;
          *reinterpret_cast<rg_u32x2*>(d) = rg_u32x2{s0.h, s1.h};
          *reinterpret_cast<rg_u32x2*>(d + A_PLANE) = rg_u32x2{s0.l, s1.l};
          if constexpr (!QKV) {
            if (gplanes && ok[j]) {
              unsigned short* const o2 = gplanes + (long)(drow0 + j) * 256 + lane4;
              *(__attribute__((address_space(1))) rg_u32x2*)(o2) = rg_u32x2{s0.h, s1.h};
              *(__attribute__((address_space(1))) rg_u32x2*)(o2 + p.ln_out_plane) = rg_u32x2{s0.l, s1.l};
            }
          }
        }
      }
    };
    rows(std::integral_constant<int, 0>{});
    rows(std::integral_constant<int, 1>{});
  };
  // the rows' residual values, requested BEFORE the accumulators go through the slab (rowgemm_wd_kernel): one base address
  // and constant row offsets (asked for with a run-time stride and a clamp per row, the ten addresses were 150 scalar
  // instructions in front of the epilogue's barrier)
  auto prefetch_rows = [&](const float* res, rg_f32x4 (&rpre)[NRW]) {
    // The wave's first row, laundered through an SGPR: the two epilogues read the same rows of h, and with a common
    // subexpression the compiler kept the ten 64-bit lane addresses alive across the feed-forward loop -- in scratch, each
    // reload followed by s_waitcnt vmcnt(0), i.e. by a wait for the residual loads just issued: six L2 round trips in
    // series in front of the epilogue (5 k cycles in the phase stamps).  A uniform base + the unsigned lane offset is the
    // scalar-base form of the load: no address registers at all.
    int row0 = m0 + wave * NRW;
    asm volatile("" : "+s"(row0));
    const float* const base = res + (long)row0 * LDH;
#pragma unroll
    for (int j = 0; j < NRW; ++j) rpre[j] = *(const __attribute__((address_space(1))) rg_f32x4*)(base + j * LDH + lane4);
  };
  auto acc_to_slab = [&](const rg_f32x4 (&acc)[RT][2]) {
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) slab[rb_slab(mt * 16 + kq * 4 + e, wave * 32 + nt * 16 + r16)] = acc[mt][nt][e];
  };

  // ---- phase A's epilogue: h += ... ; x = LayerNorm3(h) -> X ----
  {
    warm2 = warm_lines(p.W2f, p.w2f_plane, 4096, 0);           // 256 x 1024 halves = 512 KB per plane
    // (the per-column constants FIRST: vmcnt retires in order, so the first use of one of them must not have the rows'
    // residual loads in front of it -- it then waited for all ten, ~2 us, before the slab pass had even begun)
    rg_f32x4 cs4 = *reinterpret_cast<const rg_f32x4*>(p.cso + 4 * lane);
    rg_f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bo) b4 = *reinterpret_cast<const rg_f32x4*>(p.bo + 4 * lane);
    const rg_f32x4 gg = *reinterpret_cast<const rg_f32x4*>(p.ln3_g + 4 * lane);
    const rg_f32x4 bb = *reinterpret_cast<const rg_f32x4*>(p.ln3_b + 4 * lane);
    rg_f32x4 rpre[NRW];
    prefetch_rows(p.h, rpre);
    xstamp();      // 18: loads requested
    rg_lds_barrier();      // every wave is done reading the ring: the slab goes over it
    xstamp();      // 19
    acc_to_slab(acc1);
    rg_lds_barrier();
    xstamp();      // 20: slab written
    cs4 = cs4 * p.inv_a_scale_o;      // (first use of a loaded value: HERE, behind the barriers that hide the loads' ~2 k cycles, not in front of them)
    if (!JV_ABLATE(p, 16)) row_pass(cs4, b4, rpre, facts_h, p.h, LDH, p.amax_h, true, gg, bb, p.a_scale1, nullptr);
    xstamp();      // 21: row pass done (this wave)
    asm volatile("" ::"v"(warm2), "v"(warm1));      // (issued a whole row pass ago / in the middle of phase A)
    rg_lds_barrier();      // X is complete, the slab has been read: the upper half is free for H
  }
  stamp();      // 3: epilogue A done

  // ================================ phase B: the feed-forward pair (rowffn_kernel's loop) ================================
  // one 32-deep step: block 0, reload, counted wait (W alone), the NEXT step's A fragments requested into the other register
  // set, block 1, reload.  `have`: this step's fragments were requested by the previous step; st_next = null where the next
  // step's operand does not exist yet.  `waited`: the wait was done ahead of an epilogue's memory operations.
  bool waited = true;
  auto step = [&](auto par_tag, rg_f32x4 (&acc)[RT][2], const unsigned char* st, const unsigned char* st_next, const bool have) {
    constexpr int par = decltype(par_tag)::value;
    if (!have) read_a(par_tag, st);
    __builtin_amdgcn_sched_barrier(0);
    auto block = [&](auto nttag) {
      mfma_block(par_tag, nttag, acc, std::integral_constant<int, 0>{});
      if constexpr (RT > 1) mfma_block(par_tag, nttag, acc, std::integral_constant<int, 1>{});
      if constexpr (RT > 2) mfma_block(par_tag, nttag, acc, std::integral_constant<int, 2>{});
      if constexpr (RT > 3) mfma_block(par_tag, nttag, acc, std::integral_constant<int, 3>{});
      if constexpr (RT > 4) mfma_block(par_tag, nttag, acc, std::integral_constant<int, 4>{});
      __builtin_amdgcn_sched_barrier(0);
      load_w(par_tag, nttag);
      __builtin_amdgcn_sched_barrier(0);
    };
    block(std::integral_constant<int, 0>{});
    if (!waited) rg_wait_vmcnt<NWL>();
    waited = false;
    landed_w(bq[par][1][0], bq[par][1][1], bq[par ^ 1][0][0], bq[par ^ 1][0][1]);
    if (st_next) read_a(std::integral_constant<int, par ^ 1>{}, st_next);
    __builtin_amdgcn_sched_barrier(0);
    block(std::integral_constant<int, 1>{});
    advance_w();
  };

#pragma unroll
  for (int mt = 0; mt < RT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc2[mt][nt] = rg_f32x4{0.f, 0.f, 0.f, 0.f};
  const float inv1 = p.inv_a_scale1;
  // the pieces of a hidden chunk: the first linear's per-column constants, its 8 steps over X into acc1, the GELU pass that
  // turns acc1 into stage `wave` of the hidden image H, and a run of 4 steps of the second linear over H into acc2
  float csl[2], bl[2];
  auto load_cs = [&](const int c) {
    const int hc0 = c * 256 + wave * 32 + r16;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      csl[nt] = p.cs1[hc0 + 16 * nt] * inv1;
      bl[nt] = p.b1 ? p.b1[hc0 + 16 * nt] : 0.f;
    }
  };
  auto p1_run = [&](const bool have) {      // have: stage 0 of X was requested by the step before
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc1[mt][nt] = rg_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int ks = 0; ks < KS; ks += 2) {
      step(std::integral_constant<int, 0>{}, acc1, rg_lds + ks * STAGE, rg_lds + (ks + 1) * STAGE, ks > 0 || have);
      step(std::integral_constant<int, 1>{}, acc1, rg_lds + (ks + 1) * STAGE, ks + 2 < KS ? rg_lds + (ks + 2) * STAGE : nullptr, true);
    }
  };
  auto p2_run = [&](const int k0, const int n, const bool have, const unsigned char* const after) {      // stages k0 .. k0 + n - 1 of H
#pragma unroll 1
    for (int ks = k0; ks < k0 + n; ks += 2) {
      step(std::integral_constant<int, 0>{}, acc2, hreg + ks * STAGE, hreg + (ks + 1) * STAGE, ks > k0 || have);
      step(std::integral_constant<int, 1>{}, acc2, hreg + (ks + 1) * STAGE, ks + 2 < k0 + n ? hreg + (ks + 2) * STAGE : after, true);
    }
  };
  auto gelu_pass = [&]() {      // GELU -> planes, into stage `wave` of H (rowffn_kernel)
    if (JV_ABLATE(p, 2)) return;
    // A lane holds ONE hidden column (k) of rows 4 kq + e: written as it stands that is 2-byte LDS stores (80 per lane and
    // chunk, two lanes per bank word).  Neighbouring lanes (columns k, k + 1) trade one row of each pair instead -- the
    // even lane takes row e of both columns, the odd lane row e + 1 -- and store whole dwords: half the LDS instructions.
    unsigned char* const hs = hreg + wave * STAGE;
    const int key = (kq & 1) << 1;
    const int odd = r16 & 1;
    // v_perm_b32 selectors (bytes of {hi operand, lo operand} = {partner, mine}): even: (mine.lo16 | partner.lo16 << 16),
    // odd: (partner.hi16 | mine.hi16 << 16)
    const unsigned sel = odd ? 0x03020706u : 0x05040100u;
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int slot = (2 * nt + (r16 >> 3)) ^ key;
        float g4[4];
#ifdef JV_GELU_SCALAR      // (A/B builds: one value per instruction, as rowffn_kernel does)
#pragma unroll
        for (int e = 0; e < 4; ++e) g4[e] = JV_ABLATE(p, 1) ? acc1[mt][nt][e] : gelu_erf(acc1[mt][nt][e] * csl[nt] + bl[nt]) * p.h_scale;
#else
#pragma unroll
        for (int e = 0; e < 4; e += 2) {      // two rows per instruction (jv_device.h gelu_erf2: the same operations, packed)
          const jv_f32x2 x2 = jv_f32x2{acc1[mt][nt][e], acc1[mt][nt][e + 1]} * jv_f32x2{csl[nt], csl[nt]} + jv_f32x2{bl[nt], bl[nt]};
          const jv_f32x2 g2 = gelu_erf2(x2) * jv_f32x2{p.h_scale, p.h_scale};
          g4[e] = JV_ABLATE(p, 1) ? acc1[mt][nt][e] : g2[0];
          g4[e + 1] = JV_ABLATE(p, 1) ? acc1[mt][nt][e + 1] : g2[1];
        }
#endif
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const Split2 sp = JV_ABLATE(p, 1) ? Split2{__float_as_uint(g4[e]), __float_as_uint(g4[e + 1])}
                                            : split2h_pair(g4[e], g4[e + 1]);      // (row e | row e + 1 << 16) of this lane's column
          const unsigned ph = (unsigned)__builtin_amdgcn_update_dpp(0, (int)sp.h, 0xB1, 0xf, 0xf, true);      // the neighbour's
          const unsigned pl = (unsigned)__builtin_amdgcn_update_dpp(0, (int)sp.l, 0xB1, 0xf, 0xf, true);
          const unsigned wh = __builtin_amdgcn_perm(ph, sp.h, sel), wl = __builtin_amdgcn_perm(pl, sp.l, sel);
          const int row = mt * 16 + kq * 4 + e + odd;
          unsigned char* d = hs + row * 64 + (slot << 4) + (r16 & 6) * 2;
          *reinterpret_cast<unsigned*>(d) = wh;
          *reinterpret_cast<unsigned*>(d + A_PLANE) = wl;
        }
      }
  };
  // q|k|v's weights are touched ~30 us ahead of phase C: younger than every W load in flight and followed by a GELU pass
  // without a counted wait, so they stall nothing (in phase B's epilogue they came too late: its first fragment loads,
  // issued two steps earlier, were cold misses the epilogue then sat waiting for)
  auto warm_q = [&]() {
    if constexpr (QKV) warm3 = warm_lines(p.Wqf, p.wqf_plane, 6144, 0);      // 1536 x 256 halves = 768 KB per plane
  };
  if constexpr (!STAG) {
    // ---- every wave in the same phase: first linear -> GELU -> second linear, chunk by chunk ----
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
      load_cs(c);
      p1_run(c > 0);
      stamp();      // 4 + 3 c: phase 1 done
      if (c == NCH - 1) warm_q();
      if (c > 0) rg_lds_barrier();      // every wave is done reading the previous chunk's H
      gelu_pass();
      rg_lds_barrier();      // H complete
      stamp();      // 5 + 3 c: GELU done
      p2_run(0, KS, false, c + 1 < NCH ? rg_lds : nullptr);
      stamp();      // 6 + 3 c: phase 2 done
    }
  } else {
    // ---- STAGGERED: the two waves of a SIMD (w and w + 4) never run the GELU pass together.  With every wave in the same
    // phase the matrix pipe idles through each GELU pass (4 x ~6 k cycles a launch, MI355X_MICROARCH.md "Two waves per SIMD"
    // item 9); here waves 0..3 (X, which produce stages 0..3 of H) run half a chunk AHEAD of waves 4..7 (Y, stages 4..7), so one
    // half's GELU pass runs under the other half's MFMAs.  The second linear must take H's stages in ascending order (the sum's
    // K order is the unfused kernels'), so X fills the time until Y's stages exist with the NEXT chunk's first linear:
    //           X (waves 0..3)                          Y (waves 4..7)
    //   start   first(0), GELU(0)                      first(0)
    //   I1(c)   second(c) k 0..3, first(c + 1)         GELU(c)                                    | barrier: stages 4..7 of H(c) exist
    //   I2(c)   second(c) k 4..7                       second(c) k 0..3                           | barrier: stages 0..3 of H(c) are free
    //   I3(c)   GELU(c + 1)                            second(c) k 4..7, first(c + 1)             | barrier: stages 0..3 of H(c + 1) exist
    // Every barrier separates a write of H stages from their reads or their reads from the next write (three per chunk
    // instead of two).  The weight walker above hands each half its fragments in its own order of use.
    // (written as ONE loop body whose two role phases swap the halves, so that the GELU pass and the step bodies exist once)
    load_cs(0);
    p1_run(false);      // both halves: first(0); X, the older wave of each SIMD, wins the matrix pipe and gets ahead by itself
    stamp();      // 4
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
      const bool more = c + 1 < NCH;
#pragma unroll 1
      for (int ph = 0; ph < 2; ++ph) {
        // ph 0 = I3(c - 1): X GELU(c)                    | Y second(c - 1) k 4..7, first(c)    (c = 0: Y has nothing left to do)
        // ph 1 = I1(c):     X second(c) k 0..3, first(c + 1) | Y GELU(c)
        if (hx == (ph == 0)) {
          if (c == NCH - 1) warm_q();
          gelu_pass();
        } else {
          const bool do_p2 = hx || c > 0;
          const bool do_p1 = hx ? more : c > 0;
          if (do_p2) p2_run(hx ? 0 : 4, 4, false, do_p1 ? rg_lds : nullptr);
          if (do_p1) {
            load_cs(hx ? c + 1 : c);
            p1_run(true);
          }
        }
        rg_lds_barrier();
        stamp();      // 5 + 3 c, 6 + 3 c
      }
      // I2(c): X second(c) k 4..7 | Y second(c) k 0..3
      p2_run(hx ? 4 : 0, 4, false, nullptr);
      if (more) rg_lds_barrier();      // (the last one is the epilogue's)
      stamp();      // 7 + 3 c
    }
    if (!hx) p2_run(4, 4, false, nullptr);      // Y's last four steps
  }
  xstamp();      // B0: loop done
  rg_wait_vmcnt<2>();      // what the next step's middle needs (if there is one), ahead of the epilogue's own memory operations
  xstamp();      // B1: fragments of the next steps landed
  if constexpr (QKV) asm volatile("" ::"v"(warm3));      // (issued at the last GELU pass)

  // ---- phase B's epilogue: out = h + ... (+ tracking); QKV: x' = LayerNorm1_next(out) -> X ----
  {
    rg_f32x4 cs4 = *reinterpret_cast<const rg_f32x4*>(p.cs2 + 4 * lane);
    rg_f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (p.b2) b4 = *reinterpret_cast<const rg_f32x4*>(p.b2 + 4 * lane);
    rg_f32x4 gg = {1.f, 1.f, 1.f, 1.f}, bb = {0.f, 0.f, 0.f, 0.f};
    const bool ln_b = QKV || p.ln_out != nullptr;      // (not QKV: the planes go to HBM for a column-split q|k|v launch)
    if (ln_b) {
      gg = *reinterpret_cast<const rg_f32x4*>(p.ln1_g + 4 * lane);
      bb = *reinterpret_cast<const rg_f32x4*>(p.ln1_b + 4 * lane);
    }
    rg_f32x4 rpre[NRW];
    prefetch_rows(p.h, rpre);
    xstamp();      // 22: loads requested
    rg_lds_barrier();      // every wave is done with X and H
    xstamp();      // 23
    acc_to_slab(acc2);
    rg_lds_barrier();
    xstamp();      // 24: slab written
    cs4 = cs4 * p.inv_h_scale;
    if (!JV_ABLATE(p, 16))
      row_pass(cs4, b4, rpre, facts_o, p.out, QKV ? LDH : p.ldo, p.amax_out, ln_b, gg, bb, ln_b ? p.a_scale_q : 1.f,
               QKV ? nullptr : p.ln_out);      // (q|k|v follows only where out is the trunk itself)
    xstamp();      // 25: row pass done
    if constexpr (QKV) rg_lds_barrier();      // X is complete, the slab has been read: the patches go over it
  }
  stamp();      // 16: epilogue B done

  // ================================ phase C: q | k | v of the next block (rowgemm_wa_kernel's loop) ================================
  if constexpr (QKV) {
    waited = true;
    // Phase C has no workgroup barrier: six independent 256-column chunks, each 8 steps of MFMAs and then a per-wave epilogue.
    // A SIMD hosts waves w and w + 4 and its arbiter serves the OLDER wave first (tools/probes/wave_prio.hip: of two waves
    // that start a run of independent MFMAs together, wave w finishes at the single-wave rate and wave w + 4 runs in what
    // is left), so the pair drifts out of phase by itself -- one wave's epilogue under the other's MFMAs -- and an explicit
    // s_setprio for waves 0..3 changed nothing (121.1 against 121.0 us per launch, same box).
    // (two patches per wave, alternating over the row groups: group mt + 1 is written while group mt's rows are still
    // being read back -- with one patch every group was an LDS write -> read round trip in series)
    float* const ws0 = slab + wave * (NPATCH * 16 * 36);
    const int prow = lane >> 3, pc4 = (lane & 7) * 4;
    read_a(std::integral_constant<int, 0>{}, rg_lds);
#pragma unroll 1
    for (int c = 0; c < NCQ; ++c) {
#pragma unroll
      for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc1[mt][nt] = rg_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
      for (int ks = 0; ks < KS; ks += 2) {
        step(std::integral_constant<int, 0>{}, acc1, nullptr, rg_lds + (ks + 1) * STAGE, true);
        // (the last step of a chunk requests stage 0 again: the next chunk's first step)
        step(std::integral_constant<int, 1>{}, acc1, nullptr, ks + 2 < KS ? rg_lds + (ks + 2) * STAGE : (c + 1 < NCQ ? rg_lds : nullptr), true);
      }
      stamp();      // 17 + 2 c: chunk loop done
      if (c + 1 < NCQ) {
        rg_wait_vmcnt<2>();
        waited = true;
      }
      // the chunk's epilogue, per wave, through its private 16 x 36-float patch (rowgemm_wa_kernel): q -> fp32 rows, k / v -> planes
      const int nw = c * 256 + wave * 32 + pc4;
      const float sc = c < 4 ? p.k_scale : p.v_scale;
      if (JV_ABLATE(p, 8)) continue;
      if (c < 2) {      // q: fp32 rows, 8 lanes x 16 B per row segment
        rg_f32x4 cw = *reinterpret_cast<const rg_f32x4*>(p.csq + nw);
        cw = cw * p.inv_a_scale_q;
#pragma unroll
        for (int mt = 0; mt < RT; ++mt) {
          float* const ws = ws0 + (mt & (NPATCH - 1)) * (16 * 36);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) ws[(kq * 4 + e) * 36 + nt * 16 + r16] = acc1[mt][nt][e];
#pragma unroll
          for (int ps = 0; ps < 2; ++ps) {
            const int trow = mt * 16 + ps * 8 + prow;
            const long mrow = (long)m0 + trow;
            const rg_f32x4 v = *reinterpret_cast<const rg_f32x4*>(ws + (ps * 8 + prow) * 36 + pc4) * cw;
            if (mrow >= p.M || JV_ABLATE(p, 4) && v[0] != 123.456f) continue;
            *(__attribute__((address_space(1))) rg_f32x4*)(p.q + mrow * LDQ + nw) = v;
          }
        }
      } else {
        // k / v: fp16 planes.  A lane takes EIGHT columns of a row (4 lanes per row segment, 16 rows per pass): 16 bytes per
        // plane and lane -- with four columns per lane the planes left as 8-byte stores, twice the store instructions, and
        // the chunk epilogue is bound by store issue (MI355X_MICROARCH.md: 8-byte accesses run at 0.54 - 0.70 of the 16-byte rate)
        const int prow16 = lane >> 2, pc8 = (lane & 3) * 8;
        const int nw8 = c * 256 + wave * 32 + pc8;
        rg_f32x4 cw0 = *reinterpret_cast<const rg_f32x4*>(p.csq + nw8), cw1 = *reinterpret_cast<const rg_f32x4*>(p.csq + nw8 + 4);
        cw0 = cw0 * (p.inv_a_scale_q * sc);      // (powers of two: the same bits as scaling the product)
        cw1 = cw1 * (p.inv_a_scale_q * sc);
#pragma unroll
        for (int mt = 0; mt < RT; ++mt) {
          float* const ws = ws0 + (mt & (NPATCH - 1)) * (16 * 36);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) ws[(kq * 4 + e) * 36 + nt * 16 + r16] = acc1[mt][nt][e];
          const long mrow = (long)m0 + mt * 16 + prow16;
          const rg_f32x4 v0 = *reinterpret_cast<const rg_f32x4*>(ws + prow16 * 36 + pc8) * cw0;
          const rg_f32x4 v1 = *reinterpret_cast<const rg_f32x4*>(ws + prow16 * 36 + pc8 + 4) * cw1;
          if (mrow >= p.M || JV_ABLATE(p, 4) && v0[0] != 123.456f) continue;
          const Split2 s0 = split2h_pair(v0[0], v0[1]), s1 = split2h_pair(v0[2], v0[3]);
          const Split2 s2 = split2h_pair(v1[0], v1[1]), s3 = split2h_pair(v1[2], v1[3]);
          unsigned short* const o2 = p.kv2 + mrow * LDKV + (nw8 - 512);
          *(__attribute__((address_space(1))) rg_u32x4*)(o2) = rg_u32x4{s0.h, s1.h, s2.h, s3.h};
          *(__attribute__((address_space(1))) rg_u32x4*)(o2 + p.kv2_plane) = rg_u32x4{s0.l, s1.l, s2.l, s3.l};
        }
      }
      stamp();      // 18 + 2 c: chunk epilogue done
    }
  }
  // the wrapped-around W loads of the last two steps: bq stays reserved until they have landed (rowgemm_wd_kernel)
  rg_wait_vmcnt<0>();
  landed_w(bq[0][0][0], bq[0][0][1], bq[0][1][0], bq[0][1][1]);
  landed_w(bq[1][0][0], bq[1][0][1], bq[1][1][0], bq[1][1][1]);
  stamp();      // last: drained
}

}  // namespace jv
