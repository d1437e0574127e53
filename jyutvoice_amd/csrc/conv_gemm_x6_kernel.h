// conv_gemm with the contraction on the 16-bit matrix cores at fp32 accuracy: "bf16x6" (NP = 3, below) for any fp32
// operand, "fp16x3" (NP = 2, see the kernel's template comment and DESIGN.md 5) where the caller has proven the range.
//
// Every fp32 operand is split into three bf16 planes x = h + m + l (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m):
// 24 significant bits) and the product is formed from the six partial products whose weight is >= 2^-16:
//     x*y ~= h h' + (h m' + m h') + (h l' + l h' + m m')          dropped: m l' + l m' + l l'  (relative 2^-24, 2^-32)
// Each partial product of two bf16 values is exact in fp32 and the MFMA accumulates in fp32, so the result carries
// fp32-level error (measured against fp64 in tests/test_gpu_ops.py with the same bounds as the fp32-MFMA kernel), while
// v_mfma_f32_32x32x16_bf16 delivers a 32x32x16 block in 32 cycles against 8 x 64 cycles for v_mfma_f32_32x32x2_f32:
// 6 x 32 = 192 vs 512 cycles per 16 k, i.e. 2.67x less matrix-pipe time for the same fp32-accurate contraction.
//
// Weights are split once at load time (registry.hip, three bf16 planes in HBM, 6 B per weight) and travel global -> LDS by
// LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write); activations are split while the A window is staged
// into LDS through registers (after the Snake / LeakyReLU / mask prologue), once per workgroup and K-chunk.
// Both LDS images are unpadded 64-byte rows (32 bf16) whose 16-byte slots are XOR-swizzled with swz(row) -- for W on
// the SOURCE address, since the DMA writes lane-linear -- which keeps every ds_read_b128 lane group (8 consecutive k of
// one row per lane) on 64 distinct banks.  Everything outside the main loop (tile order, masks, epilogue) is shared with
// conv_gemm.hip.
// (The kernel template and its launch chain live in this header so that each tile shape is compiled in its own
// translation unit -- conv_gemm_x6_t*.hip -- in parallel; conv_gemm_x6.hip keeps the tile choice.)
#pragma once
#include <math.h>
#include <stdlib.h>

#include <stdio.h>

#include <string>
#include <vector>

#include "conv_gemm_epilogue.h"
#include "jv_common.h"
#include "jv_device.h"
#include "jv_ops.h"

namespace jv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int ROWB = 64;      // bytes per LDS row: 32 bf16, 16-byte slots swizzled

// Slot key of a row.  (row >> 2) & 3 separates the rows that share a 64-byte quadrant of the 256-byte bank row inside every
// ds_read_b128 lane group; the (row >> 1) & 1 term separates rows r and r + 2 inside the 8-lane groups of ds_write_b128,
// whose bank period is 128 bytes (without it every A-plane store was a 2-way conflict: SQ_LDS_BANK_CONFLICT 25 %).
__device__ __attribute__((aligned(64))) const float jv_zero_page[16] = {};

__device__ __forceinline__ int swz(int row) { return ((row >> 2) & 3) ^ ((row >> 1) & 1); }

__device__ __forceinline__ void split3x8(const float (&x)[8], u32x4& h, u32x4& m, u32x4& l) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {          // pair-wise: one v_cvt_pk_bf16_f32 per plane and pair
    const Split3 t = split3_pair(x[2 * e], x[2 * e + 1]);
    h[e] = t.h;
    m[e] = t.m;
    l[e] = t.l;
  }
}

// The 64x64 plain variants are held to 96 registers so five workgroups fit a CU: the N = 256 GEMMs of the estimator
// make 1220 such tiles, which then all run in one resident wave (1280 slots) instead of 1024 + a 20 % tail.
// NA2: window rows staged per thread (rows tid>>1 + 128 i, 16 k each): 1 when the A window fits 128 rows, else 2; 0 when it
// fits 64 rows: four threads per row, 8 k each, so that all four waves share the split instead of two doing all of it.
// NWB: weight buffers in LDS.  2: the DMA for step s + 1 is issued behind step s's barriers and lands during its MFMAs;
// 1: the DMA is issued between the two barriers of its own step and lands while the wave splits and stores its A rows.
// NP: planes per operand.  3: bf16 (h, m, l), six products, any fp32 operand.  2: fp16 (h, l) of A * a_scale and of the
// per-row scaled W (p.W2), three products h h' + h l' + l h' (dropped: l l', relative 2^-22) -- half the matrix-pipe,
// LDS and DMA work for operands whose range a load-time bound has proven (ConvGemmArgs::W2).
template <int BM, int BN, int WM, int WN, int PRO, int EPI, int NA2, int NWB, int NP>
__global__ __launch_bounds__(256, (BM == 64 && BN == 64 && PRO == PRO_NONE && EPI != 4) ? 5 : 2) void conv_gemm_x6_kernel(const ConvGemmArgs p, const int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int MT = WM / 32, NT = WN / 32;
  // fp16x3 scale of A: the caller's (proven at load time), or -- per staged row, below -- derived from the measured
  // maximum of the row's utterance (p.amax_in)
  const float a_scale = NP == 2 ? p.a_scale : 1.f;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
  static_assert(NWB == 1 || NWB == 2, "one or two weight buffers");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int r32 = lane & 31, half = lane >> 5;
  constexpr bool AQ = NA2 == 0;                  // quarter-row staging
  constexpr bool APL = NA2 == 3;                 // A arrives as fp16 planes (p.A2) by LDS-DMA, double-buffered like W
  static_assert(!APL || (NP == 2 && NWB == 2 && PRO == PRO_NONE), "pre-split A: fp16x3, two buffers, plain prologue");
  static_assert(NP == 2 || NP == 3, "two fp16 or three bf16 planes");
  constexpr int NR = (AQ || APL) ? 1 : NA2, NG = AQ ? 1 : 2;
  const int arow = AQ ? tid >> 2 : tid >> 1, kpart = AQ ? tid & 3 : tid & 1;
  const int koff = (AQ ? 8 : 16) * kpart;

  int m0, n0;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    n0 = (lid % tiles_n) * BN;
    m0 = (lid / tiles_n) * BM;
  }
  const float* A = p.A;
  float* out = p.out;
  unsigned long long t_start = 0, t_loop = 0;
  if (JV_STAMP(p)) t_start = __builtin_amdgcn_s_memtime();

  const int ntaps = p.ntaps, dil = p.tap_dil;
  const int win = BM + (ntaps - 1) * dil;
  unsigned char* const ldsA = reinterpret_cast<unsigned char*>(smem);            // [NP][win][64 B]
  unsigned char* const ldsW = ldsA + (APL ? 2 : 1) * NP * win * ROWB;            // [NWB][NP][BN][64 B]

  // Rows outside the matrix or masked out read as zero.  Instead of predicating every load of the loop (exec-mask
  // branches, ~16 scalar instructions per step), such a row's pointer is aimed once at a page of zeros and its per-step
  // advance set to 0: the loop body is branch-free.
  const float* asrc[NR];
  int astep[NR];
  float ascale[NR];      // fp16x3: the power of two this thread's window row is staged with
#pragma unroll
  for (int i = 0; i < (APL ? 0 : NR); ++i) {
    const int r = arow + 128 * i;
    const long ar = (long)m0 + p.tap_row0 + r;
    bool ok = (r < win) && (ar >= 0) && (ar < p.a_rows);
    if (ok && p.rowmask_in) ok = p.rowmask_in[ar] != 0;
    asrc[i] = ok ? A + ar * p.lda + koff : jv_zero_page;
    astep[i] = ok ? 1 : 0;
    // every tap row of a real output row belongs to that row's utterance (the gaps between utterances are at least a
    // window wide) or is masked to zero, so staging with the input row's slot and un-scaling with the output row's agree
    ascale[i] = (NP == 2 && p.amax_in) ? h3_scale_dev(p.amax_in[amax_slot(p, ar)] + p.a_extra) : a_scale;
  }
  int2* const rowtab = reinterpret_cast<int2*>(reinterpret_cast<unsigned char*>(smem) + p.rowtab_off);
  rowtab_fill<BM, NP == 2>(p, rowtab, m0);      // read by the epilogue, many barriers from here

  f32x4 pa[NR][2 * NG];

  auto load_A = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const float* src = asrc[i] + c0 * astep[i];
#pragma unroll
      for (int v = 0; v < 2 * NG; ++v)      // explicitly GLOBAL: through a generic pointer (A or the zero page) these are flat_load,
                                            // which counts on lgkmcnt too -- every wait for an LDS fragment then also waits for the
                                            // prefetch of the next step's A rows
        pa[i][v] = *(const __attribute__((address_space(1))) f32x4*)(src + 4 * v);
    }
  };
  auto store_A = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int r = arow + 128 * i;
      if (r < win) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {         // groups of 8 consecutive k
          float x[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = pa[i][2 * g + (e >> 2)][e & 3];
          if (PRO == PRO_SNAKE) {
            const float* al = p.pro_alpha + c0 + koff + 8 * g;
            float arg[8], inv[8];
            bool big = false;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float a = al[e];
              arg[e] = x[e] * a;
              inv[e] = 1.0f / (a + 1e-9f);
              big = big || fabsf(arg[e]) > 32768.f;      // false for NaN, which sin2_small propagates
            }
            if (snake_args_small(big)) {
#pragma unroll
              for (int e = 0; e < 8; ++e) x[e] = x[e] + inv[e] * sin2_small(arg[e]);
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float sn = sinf(arg[e]);
                x[e] = x[e] + inv[e] * (sn * sn);
              }
            }
          } else if (PRO == PRO_LRELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = x[e] > 0.f ? x[e] : x[e] * p.pro_slope;
          }
          unsigned char* dst = ldsA + r * ROWB + ((((AQ ? kpart : 2 * kpart + g)) ^ swz(r)) << 4);
          if constexpr (NP == 2) {
            u32x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const Split2 t = split2h_pair(x[2 * e] * ascale[i], x[2 * e + 1] * ascale[i]);
              h[e] = t.h;
              l[e] = t.l;
            }
            *reinterpret_cast<u32x4*>(dst) = h;
            *reinterpret_cast<u32x4*>(dst + win * ROWB) = l;
          } else {
            u32x4 h, m, l;
            split3x8(x, h, m, l);
            *reinterpret_cast<u32x4*>(dst) = h;
            *reinterpret_cast<u32x4*>(dst + win * ROWB) = m;
            *reinterpret_cast<u32x4*>(dst + 2 * win * ROWB) = l;
          }
        }
      }
    }
  };
  // NP * BN / 16 one-KiB pieces (16 weight rows x 64 B of one plane) per step, dealt round-robin to the four waves.
  // Lane L of a piece lands in LDS row L >> 2, slot L & 3, and therefore fetches k-slot (L & 3) ^ swizzle(row).
  auto dma_W = [&](int kb, int buf) {      // kb: first k column of the step (tap * Cin + chunk * 32)
    constexpr int PIECES = NP * BN / 16;
    const unsigned short* const planes = NP == 2 ? p.W2 : p.W3;
    const long pstride = NP == 2 ? p.w2_plane : p.w3_plane;
#pragma unroll
    for (int i = 0; i < PIECES / 4; ++i) {
      const int pc = wave + 4 * i;
      const int plane = pc / (BN / 16), g16 = pc % (BN / 16);
      const int row = g16 * 16 + (lane >> 2);
      const int kslot = (lane & 3) ^ swz(row);
      const int n = min(n0 + row, p.n_rows_w - 1);      // rows past the weight matrix feed columns that are never stored
      const unsigned short* src = planes + (long)plane * pstride + (long)n * p.ldw + kb + 8 * kslot;
      unsigned char* dst = ldsW + ((buf * NP + plane) * BN + g16 * 16) * ROWB;      // wave-uniform; the DMA adds lane * 16
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };

  // pre-split A: NP * BM / 16 one-KiB pieces per step, rows past the buffer clamped (their outputs are never stored)
  auto dma_A = [&](int kb, int buf) {
    constexpr int PIECES = NP * BM / 16;
#pragma unroll
    for (int i = 0; i < PIECES / 4; ++i) {
      const int pc = wave + 4 * i;
      const int plane = pc / (BM / 16), g16 = pc % (BM / 16);
      const int row = g16 * 16 + (lane >> 2);
      const int kslot = (lane & 3) ^ swz(row);
      const long ar = min((long)m0 + row, p.a_rows - 1);
      const unsigned short* src = p.A2 + (long)plane * p.a2_plane + ar * p.lda2 + kb + 8 * kslot;
      unsigned char* dst = ldsA + ((buf * NP + plane) * BM + g16 * 16) * ROWB;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;

  // split-K (ConvGemmArgs::ksplit): this workgroup contracts the chunks [c_lo, c_hi) and writes its own partial
  int c_lo = 0, c_hi = p.Cin >> 5;
  if (p.ksplit > 1) {
    const int z = blockIdx.y, nch = p.Cin >> 5;
    c_lo = z * nch / p.ksplit;
    c_hi = (z + 1) * nch / p.ksplit;
    out += (long)z * p.split_stride;
  }
  const int nsteps = (c_hi - c_lo) * ntaps;
  if constexpr (APL) dma_A(c_lo * 32, 0);
  else load_A(c_lo * 32);
  if constexpr (NWB == 2) dma_W(c_lo * 32, 0);
  if (JV_STAMP(p)) t_loop = __builtin_amdgcn_s_memtime();
  int c = c_lo, j = 0;
  for (int s = 0; s < nsteps; ++s) {
    if (!JV_ABLATE(p, 4)) __syncthreads();      // every wave is done reading the previous step's LDS images
    if constexpr (NWB == 1) dma_W(j * p.Cin + c * 32, 0);
    if constexpr (!APL) {
      if (!JV_ABLATE(p, 2)) {
        if (j == 0) store_A(c * 32);
      }
      // __syncthreads() drains vmcnt before its barrier, which retires this step's weight DMA in every wave: LDS-DMA
      // data may be read only after the issuing waves' vmcnt wait AND a barrier the reader has passed
      if (!JV_ABLATE(p, 4)) __syncthreads();
    }
    int j2 = j + 1, c2 = c;
    if (j2 == ntaps) { j2 = 0; c2 = c + 1; }
    if (s + 1 < nsteps && !JV_ABLATE(p, 1)) {
      if constexpr (APL) dma_A(c2 * 32, (s + 1) & 1);      // one barrier per step: both operands land a step ahead
      else if (j2 == 0) load_A(c2 * 32);
      // buffer (s + 1) & 1 was last read in step s - 1, which every wave finished before the barriers above
      if constexpr (NWB == 2) dma_W(j2 * p.Cin + c2 * 32, (s + 1) & 1);
    }
    const unsigned char* la = ldsA + (APL ? (s & 1) * NP * win * ROWB : 0) + (wm * WM + r32 + j * dil) * ROWB;
    const unsigned char* lw = ldsW + (NWB == 2 ? (s & 1) * NP * BN * ROWB : 0) + (wn * WN + r32) * ROWB;
    // WM, WN and the 32-row fragment steps are multiples of 16 rows, so only r32 (and the tap's row offset) enter the keys
    const int swzw = swz(r32), swza = swz(r32 + j * dil);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {            // two k-steps of 16 per 32-channel chunk
      u32x4 a[MT][NP], b[NT][NP];
      const int koffa = ((2 * ks + half) ^ swza) << 4, koffw = ((2 * ks + half) ^ swzw) << 4;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
          a[mt][pl] = *reinterpret_cast<const u32x4*>(la + (pl * win + mt * 32) * ROWB + koffa);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
          b[nt][pl] = *reinterpret_cast<const u32x4*>(lw + (pl * BN + nt * 32) * ROWB + koffw);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          f32x16 t = acc[mt][nt];
          if (JV_ABLATE(p, 8)) continue;
          if constexpr (NP == 2) {
            auto mm = [&](const u32x4& x, const u32x4& y) {
              t = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), t, 0, 0, 0);
            };
            mm(a[mt][1], b[nt][0]);      // smallest terms first
            mm(a[mt][0], b[nt][1]);
            mm(a[mt][0], b[nt][0]);
          } else {
            auto mm = [&](const u32x4& x, const u32x4& y) {
              t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), t, 0, 0, 0);
            };
            mm(a[mt][2], b[nt][0]);      // smallest terms first
            mm(a[mt][0], b[nt][2]);
            mm(a[mt][1], b[nt][1]);
            mm(a[mt][1], b[nt][0]);
            mm(a[mt][0], b[nt][1]);
            mm(a[mt][0], b[nt][0]);
          }
          acc[mt][nt] = t;
        }
    }
    j = j2;
    c = c2;
  }
  __syncthreads();
  conv_epilogue<WM, WN, EPI, NP == 2>(p, out, acc, smem, m0, n0, wm, wn, t_start, t_loop, rowtab, a_scale);
}

namespace {

template <int BM, int BN>
size_t x6_lds_bytes(const ConvGemmArgs& a, int nwb = 1) {
  const int win = BM + (a.ntaps - 1) * a.tap_dil;
  return (size_t)(a.W2 ? 2 : 3) * ((a.A2 ? 2 : 1) * win + nwb * BN) * ROWB;
}

template <int BM, int BN, int WM, int WN, int PRO, int EPI, int NA2, int NWB, int NP>
int x6_launch4(const ConvGemmArgs& a_in, hipStream_t st) {
  static bool raised[64] = {};      // per device: the attribute belongs to the kernel's image on the current device
  int dev = 0;
  JV_HIP(hipGetDevice(&dev));
  if (!raised[dev & 63]) {
    JV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_x6_kernel<BM, BN, WM, WN, PRO, EPI, NA2, NWB, NP>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    raised[dev & 63] = true;
  }
  ConvGemmArgs a = a_in;
  size_t lds = x6_lds_bytes<BM, BN>(a, NWB);
  const size_t need = (size_t)4 * 32 * (WN + 4) * sizeof(float);
  if (lds < need) lds = need;
  a.rowtab_off = (int)lds;                     // per-row table (conv_gemm_epilogue.h) behind the images and the slabs
  lds += (size_t)BM * sizeof(int2);
  const int tiles_m = cdiv(a.M, BM), tiles_n = cdiv(a.N, BN);
  const bool prof = prof_on();
  if (prof) prof_begin(st);
  if (tuning_env("JV_STAMPS")) {   // diagnostic only: synchronous, prints phase shares of this launch
    static unsigned long long* dbuf = nullptr;
    const unsigned nwg = tiles_m * tiles_n;
    if (!dbuf) (void)hipMalloc(reinterpret_cast<void**>(&dbuf), sizeof(unsigned long long) * 4 * 65536);
    ConvGemmArgs b = a;
    b.stamps = nwg <= 65536 ? dbuf : nullptr;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, st);
    hipLaunchKernelGGL((conv_gemm_x6_kernel<BM, BN, WM, WN, PRO, EPI, NA2, NWB, NP>), dim3(nwg, a.ksplit > 1 ? a.ksplit : 1), dim3(256), lds, st, b, tiles_n);      // (the splits of a tile share its stamp slot: the last writer's)
    (void)hipEventRecord(e1, st);
    (void)hipStreamSynchronize(st);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (b.stamps) {
      std::vector<unsigned long long> h((size_t)nwg * 4);
      (void)hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost);
      double pro = 0, loop = 0, epi = 0;
      unsigned long long t0 = ~0ull, t1 = 0;
      for (unsigned i = 0; i < nwg; ++i) {
        pro += (double)(h[4 * i + 1] - h[4 * i]); loop += (double)(h[4 * i + 2] - h[4 * i + 1]); epi += (double)(h[4 * i + 3] - h[4 * i + 2]);
        if (h[4 * i] < t0) t0 = h[4 * i];
        if (h[4 * i + 3] > t1) t1 = h[4 * i + 3];
      }
      const double span = (double)(t1 - t0);
      fprintf(stderr, "[stamps x6] %dx%d K %d taps %d split %d N %d grid %u: prologue %.0f  loop %.0f  epilogue %.0f ticks avg per workgroup; span %.0f ticks, %.1f us by events (%.1f ticks/us); mean resident workgroups %.1f\n",
              BM, BN, a.Cin, a.ntaps, a.ksplit, a.N, nwg, pro / nwg, loop / nwg, epi / nwg, span, ms * 1e3, span / (ms * 1e3), (pro + loop + epi) / span);
    }
    return JV_OK;
  }
  hipLaunchKernelGGL((conv_gemm_x6_kernel<BM, BN, WM, WN, PRO, EPI, NA2, NWB, NP>), dim3(tiles_m * tiles_n, a.ksplit > 1 ? a.ksplit : 1),
                     dim3(256), lds, st, a, tiles_n);
  if (prof) {
    static const std::string name = std::string(NP == 2 ? "conv_gemm_h3<" : "conv_gemm_x6<") + std::to_string(BM) + "x" + std::to_string(BN) +
                                    (PRO == PRO_SNAKE ? ",snake" : PRO == PRO_LRELU ? ",lrelu" : "") +
                                    (NA2 == 3 ? ",dmaA" : "") +
                                    ((EPI & 7) == 1 ? ",gelu" : EPI == 2 ? ",res" : EPI == 4 ? ",generic" : "") + ">";
    const double rows = (double)(a.alg_rows > 0 ? a.alg_rows : a.M);
    const double k = (double)(a.alg_k > 0 ? a.alg_k : a.ntaps * a.Cin);
    const double bytes = 4.0 * (rows * a.Cin + (double)a.N * k + rows * a.N * (1 + (a.res1 ? 1 : 0) + (a.res2 ? 1 : 0)));
    prof_end(st, name.c_str(), 2.0 * rows * a.N * k, bytes);
  }
  JV_HIP(hipGetLastError());
  return JV_OK;
}

template <int BM, int BN, int WM, int WN, int PRO, int EPI, int NA2, int NWB>
int x6_launch3(const ConvGemmArgs& a, hipStream_t st) {
  if (a.W2) return x6_launch4<BM, BN, WM, WN, PRO, EPI, NA2, NWB, 2>(a, st);
  return x6_launch4<BM, BN, WM, WN, PRO, EPI, NA2, NWB, 3>(a, st);
}

// Weight planes always travel by LDS-DMA: two W buffers when both fit beside the A window in half a CU's LDS (the DMA of
// step s + 1 then runs behind step s's MFMAs), otherwise one.  The 64-row tiles always take one: a second buffer would cost
// the 64x64 tile its fifth workgroup per CU.
template <int BM, int BN, int WM, int WN, int PRO, int EPI>
int x6_launch2(const ConvGemmArgs& a, hipStream_t st) {
  if constexpr (PRO == PRO_NONE) {
    if (a.A2) return x6_launch4<BM, BN, WM, WN, PRO, EPI, 3, 2, 2>(a, st);      // conv_gemm_x6() checked the preconditions
  }
  const int win = BM + (a.ntaps - 1) * a.tap_dil;
  if (win > (BM >= 128 ? 256 : 128)) return fail(JV_ERR_ARG, "conv_gemm_x6: window too tall for this tile variant");
  if constexpr (BM >= 128) {
    const bool two = x6_lds_bytes<BM, BN>(a, 2) <= 80 * 1024;
    if (win > 128 || BM > 128) return two ? x6_launch3<BM, BN, WM, WN, PRO, EPI, 2, 2>(a, st) : x6_launch3<BM, BN, WM, WN, PRO, EPI, 2, 1>(a, st);
    return two ? x6_launch3<BM, BN, WM, WN, PRO, EPI, 1, 2>(a, st) : x6_launch3<BM, BN, WM, WN, PRO, EPI, 1, 1>(a, st);
  } else {
    // Few tiles (small batches): occupancy cannot hide the weight DMA's latency, LDS is plentiful -> two W buffers
    const long tiles = (long)cdiv(a.M, BM) * cdiv(a.N, BN);
    const bool two = (tiles <= 1024 && x6_lds_bytes<BM, BN>(a, 2) <= 64 * 1024) ||
                     (a.W2 && 5 * x6_lds_bytes<BM, BN>(a, 2) <= 160 * 1024 && !dyn_env("JV_H3_NWB1"));   // still five per CU
    if (win <= 64) return two ? x6_launch3<BM, BN, WM, WN, PRO, EPI, 0, 2>(a, st) : x6_launch3<BM, BN, WM, WN, PRO, EPI, 0, 1>(a, st);
    return two ? x6_launch3<BM, BN, WM, WN, PRO, EPI, 1, 2>(a, st) : x6_launch3<BM, BN, WM, WN, PRO, EPI, 1, 1>(a, st);
  }
}

template <int BM, int BN, int WM, int WN, int PRO>
int x6_launch1(const ConvGemmArgs& a, hipStream_t st) {
  const bool lean = !(a.N & 3) && !(a.ldo & 3) && (!a.res1 || !(a.ldr1 & 3)) && !a.res2 && !a.rowvec && !a.rowmask_out &&
                    !a.accumulate && a.out_scale == 1.f;
  if (lean && a.act == ACT_NONE)
    return a.res1 ? x6_launch2<BM, BN, WM, WN, PRO, 2>(a, st) : x6_launch2<BM, BN, WM, WN, PRO, 0>(a, st);
  if (lean && a.act == ACT_GELU && PRO == PRO_NONE && !a.res1)
    return a.out2 ? x6_launch2<BM, BN, WM, WN, PRO_NONE, 9>(a, st) : x6_launch2<BM, BN, WM, WN, PRO_NONE, 1>(a, st);
  if (a.out2) return fail(JV_ERR_ARG, "conv_gemm_x6: plane output exists for the lean GELU epilogue only");
  return x6_launch2<BM, BN, WM, WN, PRO, 4>(a, st);
}

template <int BM, int BN, int WM, int WN>
int x6_launch(const ConvGemmArgs& a, hipStream_t st) {
  switch (a.pro) {
    case PRO_NONE: return x6_launch1<BM, BN, WM, WN, PRO_NONE>(a, st);
    case PRO_SNAKE: return x6_launch1<BM, BN, WM, WN, PRO_SNAKE>(a, st);
    case PRO_LRELU: return x6_launch1<BM, BN, WM, WN, PRO_LRELU>(a, st);
    default: return fail(JV_ERR_ARG, "conv_gemm: unknown prologue");
  }
}

}  // namespace

}  // namespace jv
