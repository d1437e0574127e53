// Tile choice of the split-plane conv_gemm (bf16x6 / fp16x3 main loops, conv_gemm_x6_kernel.h); each tile shape's kernels
// are instantiated in their own translation unit (conv_gemm_x6_t0..t4.hip).
#include <stdlib.h>

#include "conv_gemm_x6_kernel.h"

namespace jv {

int x6_tile0(const ConvGemmArgs& a, hipStream_t st);   // 128 x 128
int x6_tile1(const ConvGemmArgs& a, hipStream_t st);   //  64 x 128
int x6_tile2(const ConvGemmArgs& a, hipStream_t st);   //  64 x  64
int x6_tile3(const ConvGemmArgs& a, hipStream_t st);   // 160 x 128
int x6_tile4(const ConvGemmArgs& a, hipStream_t st);   // 128 x  64

// Same contract as conv_gemm() (argument checks done there); requires a.W3, or a.W2 + a.colscale (fp16x3, plain prologue).
int conv_gemm_x6(const ConvGemmArgs& a0, hipStream_t st) {
  ConvGemmArgs a = a0;
  if (a.W2 && (!a.colscale || !(a.amax_in || a.a_scale > 0.f))) {
    if (!a.W3) return fail(JV_ERR_ARG, "conv_gemm_x6: fp16x3 needs colscale and a_scale or amax_in");
    a.W2 = nullptr;
  }
  if (!a.W2 && !a.W3) return fail(JV_ERR_ARG, "conv_gemm_x6: no weight planes");
  if (a.A2 && (!a.W2 || a.ntaps != 1 || a.rowmask_in || (a.lda2 & 7)))
    return fail(JV_ERR_ARG, "conv_gemm_x6: pre-split A needs the fp16x3 path, one tap, no input mask, lda2 % 8 == 0");
  const int span = (a.ntaps - 1) * a.tap_dil;
  struct Cand { int bm, bn; double eff; size_t lds; int max_win; };
  // measured on the estimator shapes (tools/gemm_bench.py): the big tile amortises staging and barriers best.  The
  // 160-row tile (each wave 160 x 32) exists for the workgroup-round arithmetic: M = 19.5K rows x N = 1024 is 2.4 rounds
  // of 128 x 128 tiles over 512 slots (3 to wait for) but 1.9 rounds of 160 x 128 ones.
  const Cand cands[5] = {{128, 128, 1.0, x6_lds_bytes<128, 128>(a), 256}, {64, 128, 0.8, x6_lds_bytes<64, 128>(a), 128},
                         {64, 64, 0.8, x6_lds_bytes<64, 64>(a), 128}, {160, 128, 0.97, x6_lds_bytes<160, 128>(a), 256},
                         {128, 64, 0.9, x6_lds_bytes<128, 64>(a), 256}};
  int best = -1;
  double best_cost = 0;
  for (int i = 0; i < 3; ++i) {
    if (cands[i].bm + span > cands[i].max_win) continue;
    if (cands[i].lds > 80 * 1024 && i < 2) continue;
    const long tiles = (long)cdiv(a.M, cands[i].bm) * cdiv(a.N, cands[i].bn);
    const double cost = (double)cdivl(tiles, 256) * cands[i].bm * cands[i].bn / cands[i].eff;
    if (best < 0 || cost < best_cost) { best = i; best_cost = cost; }
  }
  // narrow outputs over very many rows (the vocoder's 64-channel stage, M ~ 1e6): a 128 x 64 tile (each wave 64 x 32) reads
  // LDS 0.75 times per MFMA instead of once and stages each weight row for twice the rows
  if (best == 2 && a.N <= 64 && a.M >= 128 * 512 && cands[4].bm + span <= cands[4].max_win && cands[4].lds <= 80 * 1024 &&
      !dyn_env("JV_NO_T128x64"))
    best = 4;
  if (best == 0 && cands[3].bm + span <= cands[3].max_win && !dyn_env("JV_NO_T160")) {
    // both run two workgroups per CU (three with two planes): rounds of resident tiles x rows per tile
    const long slots = a.W2 ? 768 : 512;
    const long r128 = cdivl((long)cdiv(a.M, 128) * cdiv(a.N, 128), slots) * 128;
    const long r160 = cdivl((long)cdiv(a.M, 160) * cdiv(a.N, 128), slots) * 160;
    if ((double)r160 / cands[3].eff < (double)r128) best = 3;
  }
  if (const char* force = dyn_env("JV_TILE")) {
    const int f = atoi(force);
    if (f >= 0 && f <= 4 && cands[f].bm + span <= cands[f].max_win) best = f;
  }
  switch (best) {
    case 0: return x6_tile0(a, st);
    case 1: return x6_tile1(a, st);
    case 2: return x6_tile2(a, st);
    case 3: return x6_tile3(a, st);
    case 4: return x6_tile4(a, st);
    default: return fail(JV_ERR_ARG, "conv_gemm_x6: no tile variant fits this tap span");
  }
}

}  // namespace jv
