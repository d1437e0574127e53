// Host side of the vocoder's row-owning ResBlock convolution (hiftconv_kernel.h): argument checks and launch.
#include "hiftconv_kernel.h"
#include "hiftpair_kernel.h"

namespace jv {

namespace {
template <int C, int NG>
int hc_launch(const HiftConvArgs& a, hipStream_t st) {
  static int raised[64] = {};      // per device: the LDS size the attribute was last raised to
  int dev = 0;
  JV_HIP(hipGetDevice(&dev));
  const int lds = hc_lds_bytes<C, NG>(a.ntaps, a.dil);
  if (lds > 160 * 1024) return fail(JV_ERR_ARG, "hiftconv: the window does not fit LDS");
  if (raised[dev & 63] < lds) {
    JV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hiftconv_kernel<C, NG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    raised[dev & 63] = lds;
  }
  const bool prof = prof_on();
  if (prof) prof_begin(st);
  hipLaunchKernelGGL((hiftconv_kernel<C, NG>), dim3(cdiv(a.M, hc_rows<NG>())), dim3(hc_threads<C, NG>()), lds, st, a);
  if (prof) {
    static const std::string name = std::string("hiftconv_h3<") + std::to_string(hc_rows<NG>()) + "x" + std::to_string(C) + ",snake>";
    const double rows = (double)(a.alg_rows > 0 ? a.alg_rows : a.M);
    const double k = (double)a.ntaps * C;
    prof_end(st, name.c_str(), 2.0 * rows * C * k, 4.0 * (rows * C * (2 + (a.res1 ? 1 : 0) + (a.res2 ? 1 : 0) + (a.accumulate ? 1 : 0)) + C * k));
  }
  JV_HIP(hipGetLastError());
  return JV_OK;
}
template <int C, int NG, int KT>
int hp_launch(const HiftPairArgs& a, hipStream_t st) {
  static int raised[64] = {};      // per device: the LDS size the attribute was last raised to
  int dev = 0;
  JV_HIP(hipGetDevice(&dev));
  const int lds = hp_lds_bytes<C, NG>(a.ntaps, a.dil);
  if (lds > 160 * 1024) return fail(JV_ERR_ARG, "hiftpair: the window does not fit LDS");
  if (raised[dev & 63] < lds) {
    JV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hiftpair_kernel<C, NG, KT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    raised[dev & 63] = lds;
  }
  const bool prof = prof_on();
  if (prof) prof_begin(st);
  const int ro = hp_rows<NG>() - (a.ntaps - 1);      // output rows per workgroup
  hipLaunchKernelGGL((hiftpair_kernel<C, NG, KT>), dim3(cdiv(a.M, ro)), dim3(hc_threads<C, NG>()), lds, st, a);
  if (prof) {
    static const std::string name = std::string("hiftpair_h3<") + std::to_string(hp_rows<NG>()) + "x" + std::to_string(C) + ",snake>";
    const double rows = (double)(a.alg_rows > 0 ? a.alg_rows : a.M);
    const double k = (double)a.ntaps * C;
    // algorithmic: both convolutions; bytes: x in (once as the operand, once as the residual), the result out (+ res2, + previous out)
    prof_end(st, name.c_str(), 2.0 * 2.0 * rows * C * k, 4.0 * (rows * C * (3 + (a.res2 ? 1 : 0) + (a.accumulate ? 1 : 0)) + 2 * C * k));
  }
  JV_HIP(hipGetLastError());
  return JV_OK;
}
}  // namespace

// A ResBlock's convolution pair in one launch (hiftpair_kernel.h): out = ((Conv1d_k(Snake2(Conv1d_k,d(Snake1(A)) + b1)) + b2) + A +
// res2) * out_scale (+ out), C = 64 / 128.  `out` must not alias A: a workgroup reads its neighbours' rows as halo.
int hiftpair(const HiftPairArgs& a, int C, hipStream_t st) {
  if (a.M <= 0) return JV_OK;
  if (!a.A || !a.alpha1 || !a.alpha2 || !a.Wf || !a.cs1 || !a.cs2 || !a.amax_in || !a.out)
    return fail(JV_ERR_ARG, "hiftpair: needs A, both Snake alphas, fragment-order weights, both colscales, the measured bound and an output");
  if (a.out == a.A) return fail(JV_ERR_ARG, "hiftpair: the output must not alias the input (halo rows)");
  if ((a.ntaps != 3 && a.ntaps != 7 && a.ntaps != 11) || a.dil < 1 || (a.ntaps - 1) * a.dil > 56) return fail(JV_ERR_ARG, "hiftpair: kernel sizes 3, 7, 11, window up to 56 rows");
  if (!(a.l1max > 0.f) || !(a.e1 > 0.f) || !(a.e2 > 0.f)) return fail(JV_ERR_ARG, "hiftpair: needs the intermediate's bound (L1 norm, Snake extras)");
  switch (C * 16 + a.ntaps) {      // (the kernel size is a template parameter: hiftpair_kernel.h)
    case 64 * 16 + 3: return hp_launch<64, 2, 3>(a, st);
    case 64 * 16 + 7: return hp_launch<64, 2, 7>(a, st);
    case 64 * 16 + 11: return hp_launch<64, 2, 11>(a, st);
    case 128 * 16 + 3: return hp_launch<128, 2, 3>(a, st);
    case 128 * 16 + 7: return hp_launch<128, 2, 7>(a, st);
    case 128 * 16 + 11: return hp_launch<128, 2, 11>(a, st);
    default: return fail(JV_ERR_ARG, "hiftpair: 64 or 128 channels, kernel size 3, 7 or 11");
  }
}

// out = ((Conv1d_k,d(Snake(A)) + bias) + res1 + res2) * out_scale (+ out): the vocoder's ResBlock convolutions, C = 64 / 128 / 256
int hiftconv(const HiftConvArgs& a, int C, hipStream_t st) {
  if (a.M <= 0) return JV_OK;
  if (!a.A || !a.alpha || !a.Wf || !a.colscale || !a.amax_in || !a.out)
    return fail(JV_ERR_ARG, "hiftconv: needs A, the Snake alphas, fragment-order weights, colscale, the measured bound and an output");
  if (a.ntaps < 1 || a.ntaps > 11 || a.dil < 1 || (a.ntaps - 1) * a.dil > 56) return fail(JV_ERR_ARG, "hiftconv: window too wide");
  // 64 and 128 channels: four-wave workgroups (160 / 80 rows, 54 - 66 KB of LDS), two per CU -- one's staging pass and
  // epilogue run under the other's main loop (vocoder stage 24.6 ms against 25.3 with one eight-wave workgroup per CU,
  // JV_HIFT_WG8, and 28.2 on the tile kernels: same box, back to back)
  const bool wg8 = dyn_env("JV_HIFT_WG8") != nullptr;
  switch (C) {
    case 64: return wg8 ? hc_launch<64, 4>(a, st) : hc_launch<64, 2>(a, st);
    case 128: return wg8 ? hc_launch<128, 2>(a, st) : hc_launch<128, 1>(a, st);
    case 256: return hc_launch<256, 1>(a, st);
    default: return fail(JV_ERR_ARG, "hiftconv: 64, 128 or 256 channels");
  }
}

}  // namespace jv
