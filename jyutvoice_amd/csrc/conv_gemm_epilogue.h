// Shared epilogue of the conv_gemm kernels (fp32-MFMA and bf16x6-MFMA main loops produce the same 32x32 accumulator
// tiles): accumulators -> per-wave LDS slab (32 rows per pass) -> float4-coalesced finish.  (Storing straight from the
// MFMA layout -- one dword per lane and instruction, 128 contiguous bytes per half-wave -- needs no LDS but 4x the store
// instructions; with the DMA-staged main loop it measured 10-17 % slower on the N >= 1024 GEMMs, whose epilogue is bound by
// store issue: qkv 142 -> 127 us, ff1 96 -> 79 us.)
#pragma once
#include <type_traits>

#include "jv_common.h"
#include "jv_device.h"

namespace jv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- per-row table of a tile (LDS, [BM] int2, filled at kernel start by rowtab_fill) -------------------------------
// The measured fp16x3 bound (ConvGemmArgs::amax_in / amax_out) is kept PER UTTERANCE: slot(row) = (row - amax_G) / amax_S
// clamped to [0, amax_nb) (one slot when amax_S == 0), so that an utterance's scales -- hence its results, bit for bit --
// do not depend on what else is in the batch.  x = bits of 1 / a_scale of the row's slot (1.0 unless the launch derives
// its scale from amax_in); y = the slot, with ROW_UNTRACKED set when the row's values must not enter amax_out (rows past
// M, and rows that amax_mask marks as padding: their values are never read as real frames).
constexpr int ROW_UNTRACKED = 1 << 30;
__device__ __forceinline__ int amax_slot(const ConvGemmArgs& p, long row) {
  if (p.amax_rows) {      // compact geometry: by table (a window row outside the buffer is masked anyway: any slot will do)
    const long r = row < 0 ? 0 : (row < p.a_rows ? row : p.a_rows - 1);
    const int s = p.amax_rows[r];
    return s < 0 ? 0 : (s >= p.amax_nb ? p.amax_nb - 1 : s);
  }
  if (p.amax_S <= 0) return 0;
  const long s = (row - p.amax_G) / p.amax_S;
  return (int)(s < 0 ? 0 : (s >= p.amax_nb ? p.amax_nb - 1 : s));
}
template <int BM, bool SC>
__device__ __forceinline__ void rowtab_fill(const ConvGemmArgs& p, int2* tab, const int m0) {
  const int t = threadIdx.x;
  if (t < BM && (p.amax_out || (SC && p.amax_in))) {
    const long m = (long)m0 + t;
    const int slot = amax_slot(p, m);
    float inv = 1.f;
    if (SC && p.amax_in) inv = 1.0f / h3_scale_dev(p.amax_in[slot] + p.a_extra);
    const bool tracked = p.amax_out && m < p.M && (!p.amax_mask || p.amax_mask[m] != 0);
    tab[t] = int2{(int)__float_as_uint(inv), slot | (tracked ? 0 : ROW_UNTRACKED)};
  }
}
// fold this wave's maxima (rmax[k] = max |value| the lane stored in tile row ridx[k]) into the slots of the tile rows
// [first, last] the wave covered: one wave-level reduction and at most one atomic per slot, and none at all once the slot
// already holds a larger value (wave-uniform branch; `seen0` = the slot of row `first` as prefetched by the caller)
template <int NR>
__device__ __forceinline__ void amax_flush(float* amax_out, const int2* tab, const int first, const int last,
                                           const int (&ridx)[NR], const unsigned (&rmax)[NR], const unsigned seen0) {
  const int lo = __builtin_amdgcn_readfirstlane(tab[first].y & (ROW_UNTRACKED - 1));
  const int hi = __builtin_amdgcn_readfirstlane(tab[last].y & (ROW_UNTRACKED - 1));
  for (int s = lo; s <= hi; ++s) {
    unsigned u = 0u;
#pragma unroll
    for (int k = 0; k < NR; ++k)
      if (tab[ridx[k]].y == s) u = max(u, rmax[k]);      // untracked rows carry the flag bit and never match
    // (a plain, cacheable load on purpose: thousands of waves polling one L2 line with a volatile load serialise there;
    // the slot only grows, so a stale value is a valid lower bound)
    const unsigned seen = s == lo ? seen0 : *reinterpret_cast<const unsigned*>(amax_out + s);
    if (__builtin_amdgcn_ballot_w64(u > seen) != 0) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) u = max(u, (unsigned)__shfl_xor((int)u, o));
      if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(amax_out + s), u);
    }
  }
}

template <int I, int N, class F>
__device__ __forceinline__ void epilogue_passes(F& f) {      // f(0), f(1), ... f(N-1) with compile-time indices
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    epilogue_passes<I + 1, N>(f);
  }
}

// EPI: bit 0 = exact GELU, bit 1 = + res1, bit 3 = write fp16 planes of value * out2_scale to p.out2 instead of fp32 rows
// (lean float4 epilogues); 4 = generic (any activation, mask, row vector,
// second residual, scaling, accumulation, ragged N).  Must be entered by all 256 threads after the main loop's last
// barrier; `smem` is reused for the slabs.
// SC: the accumulators are first multiplied by colscale[n] / a_scale (fp16x3 main loop, ConvGemmArgs::W2).
template <int WM, int WN, int EPI, bool SC = false>
__device__ __forceinline__ void conv_epilogue(const ConvGemmArgs& p, float* out, f32x16 (&acc)[WM / 32][WN / 32], float* smem,
                                              const int m0, const int n0, const int wm, const int wn,
                                              unsigned long long t_start, unsigned long long t_loop,
                                              const int2* tab, const float a_scale = 1.f) {
  constexpr int MT = WM / 32, NT = WN / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, half = lane >> 5;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  unsigned long long t_epi = 0, t_loop2 = 0, t_p0 = 0;
  // Per-row operands (mask, residuals, previous value) of four rows are fetched together before any arithmetic: a
  // rolled row loop would expose one global-load latency per row (measured: 33 K cycles per 128x128 tile).
  if (JV_STAMP(p)) t_epi = __builtin_amdgcn_s_memtime();
  if (JV_ABLATE(p, 16)) return;
  constexpr int ES = WN + 4;
  constexpr int C4 = WN / 4;                     // float4 columns per row
  constexpr int RPI = 64 / C4;                   // rows per wave-instruction
  constexpr int NIT = 32 / RPI;                  // row groups per 32-row pass
  constexpr int UN = NIT < 4 ? NIT : 4;
  float* slab = smem + wave * 32 * ES;
  const int col = (lane % C4) * 4, rsub = lane / C4;
  const int n = n0 + wn * WN + col;
  const bool nin = n < p.N;
  const bool vec = (n + 3 < p.N) && !(p.ldo & 3) && (!p.res1 || !(p.ldr1 & 3)) && (!p.res2 || !(p.ldr2 & 3));
  f32x4 bb = zero4;
  if (p.bias && nin) {
    if (n + 3 < p.N) bb = *reinterpret_cast<const f32x4*>(p.bias + n);
    else
      for (int e = 0; e < 4; ++e) bb[e] = (n + e < p.N) ? p.bias[n + e] : 0.f;
  }
  f32x4 cs = {1.f, 1.f, 1.f, 1.f};
  const bool rowscale = SC && p.amax_in != nullptr;      // measured bound: 1 / a_scale is per row (rowtab), else uniform
  if constexpr (SC) {
    const float inv = rowscale ? 1.f : 1.0f / a_scale;        // powers of two: exact
#pragma unroll
    for (int e = 0; e < 4; ++e) cs[e] = (n + e < p.N) ? p.colscale[n + e] * inv : 0.f;
  }
  // p.amax_out: max |value| over everything this thread stores, kept as the bit pattern of the non-negative float --
  // unsigned order is numeric order there, with inf and NaN on top, so the integer max propagates them
  auto amax4 = [](const f32x4& t) {
    unsigned a = 0u;
#pragma unroll
    for (int e = 0; e < 4; ++e) a = max(a, __float_as_uint(t[e]) & 0x7fffffffu);
    return a;
  };
  auto pass = [&](auto mt_tag) {
    constexpr int mt = decltype(mt_tag)::value;
    // The slab is private to the wave and LDS executes one wave's accesses in program order: no barrier between a
    // pass's stores, its loads and the next pass's stores.
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) slab[((e & 3) + 8 * (e >> 2) + 4 * half) * ES + nt * 32 + r32] = acc[mt][nt][e];
    if (JV_STAMP(p) && mt == 0) t_loop2 = __builtin_amdgcn_s_memtime();
    if constexpr (EPI != 4) {
      // lean path (host guarantees N % 4 == 0, 16-byte aligned rows, no mask / row vector / second residual / scaling):
      // out = act(acc + bias) (+ res1).  All residual loads of the pass are issued before any arithmetic.
      constexpr bool E_GELU = (EPI & 1) != 0, E_RES = (EPI & 2) != 0, E_PL = (EPI & 8) != 0;
      const int mrow = m0 + wm * WM + mt * 32 + rsub;
      const int trow0 = wm * WM + mt * 32;      // first tile row of the pass (index into the row table)
      unsigned seen0 = 0xffffffffu;
      if (p.amax_out) seen0 = *reinterpret_cast<const unsigned*>(p.amax_out + (tab[trow0].y & (ROW_UNTRACKED - 1)));
      int ridx[NIT];
      unsigned rmax[NIT];
      f32x4 x[NIT], r[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        ridx[it] = trow0 + it * RPI + rsub;
        rmax[it] = 0u;
        x[it] = *reinterpret_cast<const f32x4*>(slab + (it * RPI + rsub) * ES + col);
        if constexpr (SC) {
          if (rowscale) x[it] = x[it] * __uint_as_float((unsigned)tab[ridx[it]].x);      // 1 / a_scale of the row's utterance
        }
      }
      if constexpr (E_RES) {
        const float* rb = p.res1 + (long)mrow * p.ldr1 + n;
#pragma unroll
        for (int it = 0; it < NIT; ++it)
          r[it] = (nin && mrow + it * RPI < p.M) ? *reinterpret_cast<const f32x4*>(rb + (long)(it * RPI) * p.ldr1) : zero4;
      }
      float* ob = out + (long)mrow * p.ldo + n;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        f32x4 t = SC ? x[it] * cs + bb : x[it] + bb;
        if constexpr (E_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) t[e] = gelu_erf(t[e]);
        }
        if constexpr (E_RES) t += r[it];
        if constexpr (E_PL) {
          if (nin && mrow + it * RPI < p.M) {
            const Split2 s0 = split2h_pair(t[0] * p.out2_scale, t[1] * p.out2_scale);
            const Split2 s1 = split2h_pair(t[2] * p.out2_scale, t[3] * p.out2_scale);
            unsigned short* o2 = p.out2 + (long)(mrow + it * RPI) * p.ldo2 + n;
            typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
            *reinterpret_cast<u32x2_*>(o2) = u32x2_{s0.h, s1.h};
            *reinterpret_cast<u32x2_*>(o2 + p.out2_plane) = u32x2_{s0.l, s1.l};
          }
        } else
        if (nin && mrow + it * RPI < p.M && !JV_ABLATE(p, 64)) {
          *reinterpret_cast<f32x4*>(ob + (long)(it * RPI) * p.ldo) = t;
          rmax[it] = amax4(t);
        }
      }
      if (p.amax_out) amax_flush<NIT>(p.amax_out, tab, trow0, trow0 + 31, ridx, rmax, seen0);
      return;
    }
#pragma unroll 1
    for (int it0 = 0; it0 < NIT; it0 += UN) {
      f32x4 x[UN], r1[UN], r2[UN], pv[UN];
      bool ok[UN], keep[UN];
      int sample[UN];
      const int trow0 = wm * WM + mt * 32 + it0 * RPI;      // first tile row of this group
      unsigned seen0 = 0xffffffffu;
      if (p.amax_out) seen0 = *reinterpret_cast<const unsigned*>(p.amax_out + (tab[trow0].y & (ROW_UNTRACKED - 1)));
      int ridx[UN];
      unsigned rmax[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int row = (it0 + u) * RPI + rsub;
        const int m = m0 + wm * WM + mt * 32 + row;
        ok[u] = nin && m < p.M;
        ridx[u] = wm * WM + mt * 32 + row;
        rmax[u] = 0u;
        x[u] = *reinterpret_cast<const f32x4*>(slab + row * ES + col);
        if constexpr (SC) {
          if (rowscale) x[u] = x[u] * __uint_as_float((unsigned)tab[ridx[u]].x);
        }
        r1[u] = zero4; r2[u] = zero4; pv[u] = zero4;
        keep[u] = true;
        sample[u] = 0;
        if (ok[u]) {
          if (p.rowmask_out) keep[u] = p.rowmask_out[m] != 0;
          if (p.rowvec) sample[u] = p.row_sample[m];
          if (vec) {
            if (p.res1) r1[u] = *reinterpret_cast<const f32x4*>(p.res1 + (long)m * p.ldr1 + n);
            if (p.res2) r2[u] = *reinterpret_cast<const f32x4*>(p.res2 + (long)m * p.ldr2 + n);
            if (p.accumulate) pv[u] = *reinterpret_cast<const f32x4*>(out + (long)m * p.ldo + n);
          } else {
            for (int e = 0; e < 4; ++e) {
              if (n + e < p.N) {
                if (p.res1) r1[u][e] = p.res1[(long)m * p.ldr1 + n + e];
                if (p.res2) r2[u][e] = p.res2[(long)m * p.ldr2 + n + e];
                if (p.accumulate) pv[u][e] = out[(long)m * p.ldo + n + e];
              }
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        if (!ok[u]) continue;
        const int row = (it0 + u) * RPI + rsub;
        const int m = m0 + wm * WM + mt * 32 + row;
        f32x4 rvv = zero4;
        if (p.rowvec) {
          const float* rv = p.rowvec + (long)sample[u] * p.rowvec_ld + n;
          for (int e = 0; e < 4; ++e) rvv[e] = (n + e < p.N) ? rv[e] : 0.f;
        }
        f32x4 res;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = act_apply((SC ? x[u][e] * cs[e] : x[u][e]) + bb[e], p.act);
          if (!keep[u]) t = 0.f;
          t = ((t + rvv[e]) + r1[u][e]) + r2[u][e];
          res[e] = t * p.out_scale + pv[u][e];
        }
        float* o = out + (long)m * p.ldo + n;
        if (p.amax_out) {
          f32x4 tr = res;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e >= p.N) tr[e] = 0.f;
          rmax[u] = amax4(tr);
        }
        if (vec) {
          if (!JV_ABLATE(p, 64) || res[0] == 12345.678f) *reinterpret_cast<f32x4*>(o) = res;
        } else {
          for (int e = 0; e < 4; ++e)
            if (n + e < p.N) o[e] = res[e];
        }
      }
      // (18 K workgroups x 4 waves each sending an atomic to one address cost 0.5 ms per launch before amax_flush's check)
      if (p.amax_out) amax_flush<UN>(p.amax_out, tab, trow0, trow0 + UN * RPI - 1, ridx, rmax, seen0);
    }
  };
  epilogue_passes<0, MT>(pass);
  if (JV_STAMP(p)) t_p0 = __builtin_amdgcn_s_memtime();
  if (JV_STAMP(p) && tid == 0) {
    unsigned long long* d = p.stamps + (size_t)blockIdx.x * 4;
    d[0] = t_start; d[1] = t_loop; d[2] = t_epi; d[3] = __builtin_amdgcn_s_memtime();
  }
}

}  // namespace jv
