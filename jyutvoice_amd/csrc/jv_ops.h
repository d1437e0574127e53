// Declarations of the small row-buffer kernels (rowops.hip, hiftops.hip, encops.hip).
#pragma once
#include "jv_common.h"

namespace jv {

// uoff (optional, [nb * reps + 1] first rows, device): the COMPACT geometry of ragged batches -- utterance b at rows uoff[b] ..,
// no padding to the longest (rowops.hip); null: the uniform geometry G + b S + t
int row_meta(unsigned char* rowmask, int* row_sample, const int* lens, int nb, int reps, int G, int S, int L, long rows,
             int mul, int add, hipStream_t st, const int* uoff = nullptr);
// src[b*src_bstride + c*pitch + t] (channels-first) -> dst[(G + b*S + t)*ld + col0 + c]
int cf_to_rows(const float* src, long src_bstride, long pitch, int B, int C, int T, float* dst, int ld, int col0, int G,
               int S, float scale, const int* lens, hipStream_t st, const int* uoff = nullptr);
int rows_to_cf(const float* src, int ld, int col0, int G, int S, float* dst, long dst_bstride, int B, int C, int T,
               const int* lens, hipStream_t st, const int* uoff = nullptr);
int assemble_xin(const float* x, const float* mu, const float* spks, const float* cond, float* xin, int B, int G, int S,
                 int L, long rows2, hipStream_t st, const int* uoff = nullptr, const int* row_sample = nullptr,
                 const unsigned char* rowmask = nullptr);
int assemble_xin_plain(const float* x, const float* mu, const float* spks, const float* cond, float* xin, int B, int G,
                       int S, int L, long rows, hipStream_t st);
int time_sinusoid(const float* t, int t_stride, float* out, int B, hipStream_t st);
int euler_cfg(float* x, const float* d, int B, int G, int S, int L, const float* dt_table, int step, float rate,
              hipStream_t st, const int* uoff = nullptr, const int* lens = nullptr);
int ln_epilogue_rows(float* x, const float* g, const float* b, float eps, long rows, int C, int act,
                     const unsigned char* rowmask, const float* rowvec, const int* row_sample, int rowvec_ld, const float* res,
                     long ldr, float scale, hipStream_t st, float* amax_out = nullptr, int amax_G = 0, int amax_S = 0,
                     int amax_nb = 1, const int* amax_rows = nullptr);
int fill(float* p, float v, long n, hipStream_t st);
int fill_int(int* p, int v, long n, hipStream_t st);

}  // namespace jv
