// conv_gemm_x6 kernels of one tile shape (BM, BN, WM, WN = 64, 64, 32, 32); see conv_gemm_x6_kernel.h
#include "conv_gemm_x6_kernel.h"

namespace jv {

int x6_tile2(const ConvGemmArgs& a, hipStream_t st) { return x6_launch<64, 64, 32, 32>(a, st); }

}  // namespace jv
