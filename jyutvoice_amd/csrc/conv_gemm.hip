// Implicit-GEMM 1-D convolution / linear layer on row buffers, fp32 in / fp32 accumulate on the
// CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact f32 fma chain, 64 FLOP/clk/SIMD).
//
//   out[m, n] = epilogue( sum_{j<ntaps} sum_{ci<Cin} pro(A[m + tap_row0 + j*tap_dil, ci]) * W[n, j*Cin + ci] )
//
// One kernel covers every contraction of the path except attention: Linear (ntaps=1), causal /
// "same" / dilated Conv1d (a tap is a row offset in the row buffer), strided Conv1d (lda = stride*C,
// Cin = k*C) and ConvTranspose1d (polyphase: N = stride*Cout, 3 taps, weights packed per phase).
//
// Tiling: 256 threads = 4 waves (2x2), each wave owns WM x WN outputs as (WM/32)x(WN/32) MFMA tiles.
// Per K-chunk of 32 channels the A *window* (BM + (ntaps-1)*dil rows) is staged into LDS once --
// with the prologue (mask select, Snake, LeakyReLU) applied on the way -- and reused by every tap;
// the weight tile [BN][32] is staged per (tap, chunk).  Rows are 36 floats apart in LDS so the
// ds_read_b128 operand fetches (16 rows of one k-quad per lane group) are bank-conflict free.
// K order inside a chunk is permuted (lane half h owns k = 16h..16h+15) so one b128 read feeds four
// MFMA steps; A and W use the same permutation, so the sum is unchanged.
//
// Pipeline (round-1 PMC: two barriers around the LDS refill left the MFMA pipe 46 % busy): LDS holds TWO A windows and
// TWO weight tiles.  Step s computes from buffers (c&1, s&1) while, between the two halves of its MFMA stream, the
// wave writes the registers prefetched during step s-1 into the *other* buffers and issues the global loads for step
// s+2 -- so the refill runs under queued MFMAs and one barrier per step suffices.
//
// Workgroups are numbered so that each XCD (private 4 MiB L2) gets a contiguous run of tiles with n fastest: the
// N/BN tiles that share an A row-panel run together on one XCD and the panel is fetched from HBM once
// (round 1: FETCH_SIZE was 2.6x the algorithmic bytes with the default round-robin order).
#include <math.h>
#include <stdlib.h>

#include <stdio.h>

#include <algorithm>
#include <string>
#include <type_traits>
#include <vector>

#include "conv_gemm_epilogue.h"
#include "jv_common.h"
#include "jv_device.h"
#include "jv_ops.h"

namespace jv {

constexpr int LDS_STRIDE = 36;

// EPI: bit 0 = exact GELU, bit 1 = + res1 (lean float4 epilogues); 4 = generic (any activation, mask, row vector,
// second residual, scaling, accumulation, ragged N)
template <int BM, int BN, int WM, int WN, int NAMAX, int PRO, int EPI>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const ConvGemmArgs p, const int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int MT = WM / 32, NT = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
  constexpr int NB = BN / 32;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int r32 = lane & 31, half = lane >> 5;
  const int r8 = tid >> 3, c4 = tid & 7;

  // XCD-aware tile id: blocks b and b+8 share an XCD; give each XCD a contiguous range, n fastest
  int m0, n0;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    n0 = (lid % tiles_n) * BN;
    m0 = (lid / tiles_n) * BM;
  }

  const float* A = p.A;
  const float* W = p.W;
  float* out = p.out;
  if (gridDim.z > 1) {
    const int z1 = blockIdx.z / p.nb2, z2 = blockIdx.z % p.nb2;
    A += z1 * p.sA1 + z2 * p.sA2;
    W += z1 * p.sW1 + z2 * p.sW2;
    out += z1 * p.sO1 + z2 * p.sO2;
  }

  unsigned long long t_start = 0, t_loop = 0;
  if (JV_STAMP(p)) t_start = __builtin_amdgcn_s_memtime();
  const int ntaps = p.ntaps, dil = p.tap_dil;
  const int win = BM + (ntaps - 1) * dil;
  const int na = (win + 31) >> 5;
  float* const ldsA0 = smem;                                   // [2][win][36]
  float* const ldsW0 = smem + 2 * win * LDS_STRIDE;            // [2][BN][36]

  // which of this thread's window rows exist and are unmasked (constant over the K loop)
  unsigned avalid = 0;
#pragma unroll
  for (int i = 0; i < NAMAX; ++i) {
    const int r = r8 + 32 * i;
    const long ar = (long)m0 + p.tap_row0 + r;
    bool ok = (i < na) && (r < win) && (ar >= 0) && (ar < p.a_rows);
    if (ok && p.rowmask_in) ok = p.rowmask_in[ar] != 0;
    avalid |= ok ? (1u << i) : 0u;
  }

  f32x4 pa[NAMAX];
  f32x4 pw[NB];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  int2* const rowtab = reinterpret_cast<int2*>(reinterpret_cast<unsigned char*>(smem) + p.rowtab_off);
  rowtab_fill<BM, false>(p, rowtab, m0);      // read by the epilogue (amax_out), behind the main loop's barriers

  auto load_A = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NAMAX; ++i) {
      const long ar = (long)m0 + p.tap_row0 + r8 + 32 * i;
      pa[i] = ((avalid >> i) & 1u) ? *reinterpret_cast<const f32x4*>(A + ar * p.lda + c0 + 4 * c4) : zero4;
    }
  };
  auto load_W = [&](int j, int c0) {
    const int kb = j * p.Cin + c0 + 4 * c4;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int n = n0 + r8 + 32 * i;
      pw[i] = (n < p.n_rows_w) ? *reinterpret_cast<const f32x4*>(W + (long)n * p.ldw + kb) : zero4;
    }
  };
  auto store_A = [&](int c0, int buf) {
    float* dst = ldsA0 + buf * win * LDS_STRIDE;
    f32x4 al = {1.f, 1.f, 1.f, 1.f};
    if (PRO == PRO_SNAKE) al = *reinterpret_cast<const f32x4*>(p.pro_alpha + c0 + 4 * c4);
#pragma unroll
    for (int i = 0; i < NAMAX; ++i) {
      const int r = r8 + 32 * i;
      if (i < na && r < win) {
        f32x4 v = pa[i];
        if (PRO == PRO_SNAKE) {
          const f32x4 arg = v * al;
          const bool big = fabsf(arg[0]) > 32768.f || fabsf(arg[1]) > 32768.f || fabsf(arg[2]) > 32768.f ||
                           fabsf(arg[3]) > 32768.f;      // false for NaN, which sin2_small propagates
          if (snake_args_small(big)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] + (1.0f / (al[e] + 1e-9f)) * sin2_small(arg[e]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float sn = sinf(arg[e]);
              v[e] = v[e] + (1.0f / (al[e] + 1e-9f)) * (sn * sn);
            }
          }
        } else if (PRO == PRO_LRELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.pro_slope;
        }
        *reinterpret_cast<f32x4*>(dst + r * LDS_STRIDE + 4 * c4) = v;
      }
    }
  };
  auto store_W = [&](int buf) {
    float* dst = ldsW0 + buf * BN * LDS_STRIDE;
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(dst + (r8 + 32 * i) * LDS_STRIDE + 4 * c4) = pw[i];
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;

  auto mfma_quads = [&](const float* la, const float* lw, int q_lo) {
#pragma unroll
    for (int q = q_lo; q < q_lo + 2; ++q) {
      f32x4 a[MT], b[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const f32x4*>(la + mt * 32 * LDS_STRIDE + 4 * q);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const f32x4*>(lw + nt * 32 * LDS_STRIDE + 4 * q);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].x, b[nt].x, acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].y, b[nt].y, acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].z, b[nt].z, acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].w, b[nt].w, acc[mt][nt], 0, 0, 0);
        }
    }
  };

  // ---- main loop: step s = (chunk c, tap j); registers hold exactly one pending A window and one pending W tile ----
  const int nchunks = p.Cin >> 5;
  const int nsteps = nchunks * ntaps;
  // De-phase the workgroups that share a CU: all blocks start together, run equally long main loops and would reach
  // their (HBM-write-bound) epilogues together, leaving the matrix pipe idle.  Blocks of the second dispatch wave
  // start half a tile late, so one workgroup's epilogue/prologue runs under the other's MFMAs from then on.
  if (JV_ABLATE(p, 32) && blockIdx.x >= 256 && blockIdx.x < 512) {
    const int naps = (nsteps * MT * NT * 1024) / 8128 / 2 + 1;     // ~half of this tile's MFMA time
    for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(127);
  }
  load_A(0);
  load_W(0, 0);
  store_A(0, 0);
  store_W(0);
  if (nsteps > 1) load_W(ntaps > 1 ? 1 : 0, ntaps > 1 ? 0 : 32);      // W of step 1
  if (ntaps == 1 && nchunks > 1) load_A(32);                          // A of chunk 1 is stored during step 0
  __syncthreads();

  if (JV_STAMP(p)) t_loop = __builtin_amdgcn_s_memtime();
  int c = 0, j = 0;
  for (int s = 0; s < nsteps; ++s) {
    const float* la = ldsA0 + (c & 1) * win * LDS_STRIDE + (wm * WM + r32 + j * dil) * LDS_STRIDE + 16 * half;
    const float* lw = ldsW0 + (s & 1) * BN * LDS_STRIDE + (wn * WN + r32) * LDS_STRIDE + 16 * half;
    mfma_quads(la, lw, 0);
    // refill the other buffers under the queued MFMAs
    int j1 = j + 1, c1 = c;                      // step s+1
    if (j1 == ntaps) { j1 = 0; c1 = c + 1; }
    if (s + 1 < nsteps) {
      if (!JV_ABLATE(p, 2)) {
        store_W((s + 1) & 1);
        if (j == ntaps - 1) store_A(c1 * 32, c1 & 1);
      }
      int j2 = j1 + 1, c2 = c1;                  // step s+2
      if (j2 == ntaps) { j2 = 0; c2 = c1 + 1; }
      if (!JV_ABLATE(p, 1)) {
        if (s + 2 < nsteps) load_W(j2, c2 * 32);
        if (j1 == ntaps - 1 && c1 + 1 < nchunks) load_A((c1 + 1) * 32);   // stored during step s+1
      }
    }
    mfma_quads(la, lw, 2);
    if (!JV_ABLATE(p, 4)) __syncthreads();
    j = j1;
    c = c1;
  }

  conv_epilogue<WM, WN, EPI>(p, out, acc, smem, m0, n0, wm, wn, t_start, t_loop, rowtab);
}

void conv_gemm_defaults(ConvGemmArgs& a) {
  a = ConvGemmArgs{};
  a.ntaps = 1;
  a.tap_dil = 1;
  a.out_scale = 1.f;
  a.a_scale = 1.f;
  a.out2_scale = 1.f;
  a.nb2 = 1;
  a.ln_eps = 1e-5f;
}

namespace {

constexpr size_t LDS_BUDGET = 80 * 1024;   // two workgroups per CU

template <int BM, int BN>
size_t lds_bytes(const ConvGemmArgs& a) {
  const int win = BM + (a.ntaps - 1) * a.tap_dil;
  return (size_t)2 * (win + BN) * LDS_STRIDE * sizeof(float);
}

template <int BM, int BN, int WM, int WN, int NAMAX, int PRO, int EPI>
int launch2(const ConvGemmArgs& a_in, int nbatch, hipStream_t st) {
  ConvGemmArgs a = a_in;
  const int win = BM + (a.ntaps - 1) * a.tap_dil;
  if (win > 32 * NAMAX) return fail(JV_ERR_ARG, "conv_gemm: window too tall for this tile variant");
  size_t lds = lds_bytes<BM, BN>(a);
  const size_t need = (size_t)4 * 32 * (WN + 4) * sizeof(float);
  if (lds < need) lds = need;
  a.rowtab_off = (int)lds;
  lds += (size_t)BM * sizeof(int2);
  const int tiles_m = cdiv(a.M, BM), tiles_n = cdiv(a.N, BN);
  dim3 grid(tiles_m * tiles_n, 1, nbatch);
  const bool prof = prof_on();
  if (prof) prof_begin(st);
  if (tuning_env("JV_STAMPS")) {   // diagnostic only: synchronous, prints phase shares of this launch
    static unsigned long long* dbuf = nullptr;
    const size_t nb = (size_t)grid.x * 4;
    if (!dbuf) (void)hipMalloc(reinterpret_cast<void**>(&dbuf), sizeof(unsigned long long) * 4 * 65536);
    ConvGemmArgs b = a;
    b.stamps = grid.x <= 65536 ? dbuf : nullptr;
    hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, NAMAX, PRO, EPI>), grid, dim3(256), lds, st, b, tiles_n);
    (void)hipStreamSynchronize(st);
    if (b.stamps) {
      std::vector<unsigned long long> h(nb);
      (void)hipMemcpy(h.data(), dbuf, nb * 8, hipMemcpyDeviceToHost);
      double pro = 0, loop = 0, epi = 0, first = 1e300, last = 0;
      for (unsigned i = 0; i < grid.x; ++i) {
        pro += (double)(h[4 * i + 1] - h[4 * i]); loop += (double)(h[4 * i + 2] - h[4 * i + 1]); epi += (double)(h[4 * i + 3] - h[4 * i + 2]);
        first = std::min(first, (double)h[4 * i]); last = std::max(last, (double)h[4 * i + 3]);
      }
      fprintf(stderr, "[stamps] %dx%d grid %u: prologue %.0f  loop %.0f  epilogue %.0f cycles avg per workgroup; kernel span %.0f cycles\n",
              BM, BN, grid.x, pro / grid.x, loop / grid.x, epi / grid.x, last - first);
    }
    return JV_OK;
  }
  hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, NAMAX, PRO, EPI>), grid, dim3(256), lds, st, a, tiles_n);
  if (prof) {
    static const std::string name = std::string("conv_gemm<") + std::to_string(BM) + "x" + std::to_string(BN) +
                                    (PRO == PRO_SNAKE ? ",snake" : PRO == PRO_LRELU ? ",lrelu" : "") +
                                    (EPI == 1 ? ",gelu" : EPI == 2 ? ",res" : EPI == 4 ? ",generic" : "") + ">";
    const double rows = (double)(a.alg_rows > 0 ? a.alg_rows : a.M) * nbatch;
    const double k = (double)(a.alg_k > 0 ? a.alg_k : a.ntaps * a.Cin);
    // algorithmic traffic: A rows once, W once, out once (+ residual reads), fp32
    const double bytes = 4.0 * (rows * a.Cin + (double)a.N * k * nbatch + rows * a.N * (1 + (a.res1 ? 1 : 0) + (a.res2 ? 1 : 0)));
    prof_end(st, name.c_str(), 2.0 * rows * a.N * k, bytes);
  }
  JV_HIP(hipGetLastError());
  return JV_OK;
}

template <int BM, int BN, int WM, int WN, int NAMAX, int PRO>
int launch1(const ConvGemmArgs& a, int nbatch, hipStream_t st) {
  const bool lean = !(a.N & 3) && !(a.ldo & 3) && (!a.res1 || !(a.ldr1 & 3)) && !a.res2 && !a.rowvec && !a.rowmask_out &&
                    !a.accumulate && a.out_scale == 1.f && !dyn_env("JV_GENERIC_EPI");
  if (lean && a.act == ACT_NONE)
    return a.res1 ? launch2<BM, BN, WM, WN, NAMAX, PRO, 2>(a, nbatch, st) : launch2<BM, BN, WM, WN, NAMAX, PRO, 0>(a, nbatch, st);
  if (lean && a.act == ACT_GELU && PRO == PRO_NONE && !a.res1) return launch2<BM, BN, WM, WN, NAMAX, PRO_NONE, 1>(a, nbatch, st);
  return launch2<BM, BN, WM, WN, NAMAX, PRO, 4>(a, nbatch, st);
}

template <int BM, int BN, int WM, int WN, int NAMAX>
int launch(const ConvGemmArgs& a, int nbatch, hipStream_t st) {
  switch (a.pro) {
    case PRO_NONE: return launch1<BM, BN, WM, WN, NAMAX, PRO_NONE>(a, nbatch, st);
    case PRO_SNAKE: return launch1<BM, BN, WM, WN, NAMAX, PRO_SNAKE>(a, nbatch, st);
    case PRO_LRELU: return launch1<BM, BN, WM, WN, NAMAX, PRO_LRELU>(a, nbatch, st);
    default: return fail(JV_ERR_ARG, "conv_gemm: unknown prologue");
  }
}

template <int BM, int BN, int WM, int WN, int NAMAX, int PRO, int EPI>
int raise_lds1() {
  JV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<BM, BN, WM, WN, NAMAX, PRO, EPI>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  return JV_OK;
}

template <int BM, int BN, int WM, int WN, int NAMAX, int PRO>
int raise_lds() {
  JV_TRY((raise_lds1<BM, BN, WM, WN, NAMAX, PRO, 0>()));
  if (PRO == PRO_NONE) JV_TRY((raise_lds1<BM, BN, WM, WN, NAMAX, PRO_NONE, 1>()));
  JV_TRY((raise_lds1<BM, BN, WM, WN, NAMAX, PRO, 2>()));
  JV_TRY((raise_lds1<BM, BN, WM, WN, NAMAX, PRO, 4>()));
  return JV_OK;
}

template <int BM, int BN, int WM, int WN, int NAMAX>
int raise_lds3() {
  JV_TRY((raise_lds<BM, BN, WM, WN, NAMAX, PRO_NONE>()));
  JV_TRY((raise_lds<BM, BN, WM, WN, NAMAX, PRO_SNAKE>()));
  JV_TRY((raise_lds<BM, BN, WM, WN, NAMAX, PRO_LRELU>()));
  return JV_OK;
}

}  // namespace

int conv_gemm_init() {
  JV_TRY((raise_lds3<128, 128, 64, 64, 6>()));
  JV_TRY((raise_lds3<64, 128, 32, 64, 4>()));
  JV_TRY((raise_lds3<64, 64, 32, 32, 4>()));
  return JV_OK;
}

int conv_gemm(const ConvGemmArgs& a, int nbatch, hipStream_t st) {
  if (a.M <= 0 || a.N <= 0) return JV_OK;
  if (a.Cin <= 0 || (a.Cin & 31)) return fail(JV_ERR_ARG, "conv_gemm: Cin must be a positive multiple of 32");
  if ((a.lda & 3) || (a.ldw & 3)) return fail(JV_ERR_ARG, "conv_gemm: lda/ldw must be multiples of 4 floats");
  if (a.ntaps < 1 || a.tap_dil < 1) return fail(JV_ERR_ARG, "conv_gemm: bad taps");
  if (a.n_rows_w < a.N) return fail(JV_ERR_ARG, "conv_gemm: weight has fewer rows than N");
  if (a.rowvec && !a.row_sample) return fail(JV_ERR_ARG, "conv_gemm: rowvec needs row_sample");
  if (a.ln) {
    // conv + bias into `out`, then one bandwidth-bound pass: LayerNorm over the N columns, activation, mask,
    // per-utterance vector, residual.  (A fused 64x256 LayerNorm tile ran at 53 TFLOP/s in round 1 -- slower than the
    // plain tile plus this 10 us pass -- and cannot double-buffer its 256-row weight tile within two workgroups per CU.)
    if (nbatch != 1 || a.res2 || a.accumulate || a.ldo != a.N || (a.N & 3) || a.N > 1024)
      return fail(JV_ERR_ARG, "conv_gemm: LayerNorm epilogue needs a contiguous [M,N] output, N % 4 == 0, N <= 1024");
    ConvGemmArgs g = a;
    g.ln = 0; g.act = ACT_NONE; g.rowmask_out = nullptr; g.rowvec = nullptr; g.res1 = nullptr; g.out_scale = 1.f;
    g.amax_out = nullptr;      // what stays in `out` is what the LayerNorm pass writes
    JV_TRY(conv_gemm(g, 1, st));
    return ln_epilogue_rows(a.out, a.ln_g, a.ln_b, a.ln_eps, a.M, a.N, a.act, a.rowmask_out, a.rowvec, a.row_sample,
                            a.rowvec_ld, a.res1, a.ldr1, a.out_scale, st, a.amax_out, a.amax_G, a.amax_S, a.amax_nb, a.amax_rows);
  }
  if (const char* ab = tuning_env("JV_ABLATE")) const_cast<ConvGemmArgs&>(a).ablate = atoi(ab);
  if ((a.W3 || a.W2) && nbatch == 1 && (a.ldw & 7) == 0 && !dyn_env("JV_NO_X6")) return conv_gemm_x6(a, st);
  // only the split-plane kernels honour ksplit / grid.y: here the whole sum would land in partial[0] and the reduction
  // would add stale partials to it
  if (a.ksplit > 1) return fail(JV_ERR_ARG, "conv_gemm: split-K exists on the split-plane (x6 / h3) kernels only");
  if (a.A2 || a.out2) return fail(JV_ERR_ARG, "conv_gemm: fp16 plane operands exist on the fp16x3 path only");
  // Tile choice: the kernel is MFMA-bound, so cost ~ (#workgroup waves over 256 CUs) x tile area, with a mild penalty
  // for the smaller tiles' lower operand reuse; a variant must fit two workgroups' double-buffered LDS on a CU.
  const int span = (a.ntaps - 1) * a.tap_dil;
  struct Cand { int bm, bn; double eff; size_t lds; };
  const Cand cands[3] = {{128, 128, 1.0, lds_bytes<128, 128>(a)}, {64, 128, 0.95, lds_bytes<64, 128>(a)},
                         {64, 64, 0.88, lds_bytes<64, 64>(a)}};
  int best = -1;
  double best_cost = 0;
  for (int i = 0; i < 3; ++i) {
    if (i == 0 && 128 + span > 192) continue;
    if (i > 0 && 64 + span > 128) continue;
    if (cands[i].lds > LDS_BUDGET && i < 2) continue;
    const long tiles = (long)cdiv(a.M, cands[i].bm) * cdiv(a.N, cands[i].bn) * nbatch;
    const double cost = (double)cdivl(tiles, 256) * cands[i].bm * cands[i].bn / cands[i].eff;
    if (best < 0 || cost < best_cost) { best = i; best_cost = cost; }
  }
  if (const char* force = dyn_env("JV_TILE")) {   // tuning aid: force a tile variant (0, 1, 2)
    const int f = atoi(force);
    if (f >= 0 && f <= 2 && !((f == 0 && 128 + span > 192) || (f > 0 && 64 + span > 128))) best = f;
  }
  switch (best) {
    case 0: return launch<128, 128, 64, 64, 6>(a, nbatch, st);
    case 1: return launch<64, 128, 32, 64, 4>(a, nbatch, st);
    case 2: return launch<64, 64, 32, 32, 4>(a, nbatch, st);
    default: return fail(JV_ERR_ARG, "conv_gemm: no tile variant fits this tap span");
  }
}

}  // namespace jv
