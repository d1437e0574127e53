// conv_gemm_x6 kernels of one tile shape (BM, BN, WM, WN = 128, 128, 64, 64); see conv_gemm_x6_kernel.h
#include "conv_gemm_x6_kernel.h"

namespace jv {

int x6_tile0(const ConvGemmArgs& a, hipStream_t st) { return x6_launch<128, 128, 64, 64>(a, st); }

}  // namespace jv
