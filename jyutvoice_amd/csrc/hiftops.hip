// HiFT vocoder glue kernels: NSF source (sine generator), 16-point STFT / iSTFT, F0 head.
// Reference: jyutvoice/hifigan/generator.py:141-176 (SineGen), 220-236 (SourceModuleHnNSF),
// 371-394 (_stft/_istft), 425-432 (exp/sin head + clamp); jyutvoice/hifigan/f0_predictor.py:52-55.
#include <math.h>

#include "jv_common.h"
#include "jv_ops.h"

namespace jv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__constant__ float c_cos16[16] = {1.f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
                                  0.f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f,
                                  -1.f, -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f,
                                  0.f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f};
__constant__ float c_sin16[16] = {0.f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f,
                                  1.f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
                                  0.f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f,
                                  -1.f, -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f};
// periodic Hann, scipy.signal.get_window("hann", 16, fftbins=True): 0.5 - 0.5 cos(2 pi n / 16)
__constant__ float c_hann16[16] = {0.f, 0.03806023374435663f, 0.14644660940672624f, 0.30865828381745514f,
                                   0.5f, 0.69134171618254486f, 0.85355339059327376f, 0.96193976625564337f,
                                   1.f, 0.96193976625564337f, 0.85355339059327376f, 0.69134171618254486f,
                                   0.5f, 0.30865828381745514f, 0.14644660940672624f, 0.03806023374435663f};

// ---- F0 head: f0[b,t] = | w . h[row(b,t)] + bias | over 512 channels, one wave per row --------------------
__global__ __launch_bounds__(256) void f0_head_kernel(const float* __restrict__ h, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ f0, int B, int T,
                                                      int G, int S) {
  const int lane = threadIdx.x & 63;
  const long idx = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (idx >= (long)B * T) return;
  const int b = (int)(idx / T), t = (int)(idx - (long)b * T);
  const float* row = h + ((long)G + (long)b * S + t) * 512;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(row + 4 * (lane + 64 * i));
    const f32x4 ww = *reinterpret_cast<const f32x4*>(w + 4 * (lane + 64 * i));
    s += (a[0] * ww[0] + a[1] * ww[1]) + (a[2] * ww[2] + a[3] * ww[3]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) f0[idx] = fabsf(s + bias[0]);
}

int f0_head(const float* h, const float* w, const float* bias, float* f0, int B, int T, int G, int S, hipStream_t st) {
  hipLaunchKernelGGL(f0_head_kernel, dim3((unsigned)cdivl((long)B * T, 4)), dim3(256), 0, st, h, w, bias, f0, B, T, G, S);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// ---- sine generator phase: frac[b,h,n] = (cumsum_n f0[b, n/480]*(h+1)/24000) mod 1 --------------------------
// The reference's torch.cumsum (CPU) accumulates sequentially in fp64 and rounds every prefix to fp32; the running sum
// reaches ~1e4 where one fp32 ulp is ~1e-3 cycles, so that rounding *is* part of the signal and the sequential order has
// to be reproduced, not approximated.  What makes that parallel: inside a mel frame the increment F is one fp32 value,
// and as long as the fp64 sum's ulp is no coarser than F's last fp32 bit, every one of the 480 additions is EXACT -- the
// prefix after k additions is start + k F with no rounding at all, whatever the order.  (F's last bit is 2^(e-23) for
// F in [2^e, 2^(e+1)); the sum's ulp is 2^(E-52): exact iff E - 52 <= e - 23, i.e. for every voiced frame of speech --
// f0 (h+1) > ~0.1 Hz after a minute of audio.)  So:
// (and the frame's start value must itself be a multiple of the end value's ulp).
//   1. frame_scan: one lane per (utterance, harmonic) walks the T frames (not the 480 T samples), adding 480 F in one exact
//      step, and falls back to the 480 sequential additions for a frame that fails the test (f0 ~ 0: rare);
//   2. source_mix evaluates frac = start + (k + 1) F per sample directly (exact frames), or re-runs the frame's sequential
//      additions up to its sample (inexact frames) -- the 9 x 480 T phase tensor is never written or read.
// Round 1 ran the 480 T dependent additions in one lane per (utterance, harmonic): 1.6 ms per pass.
__device__ __forceinline__ bool frame_adds_exact(const double start, const float F) {
  // exponent fields: start + 480 F stays below 2^(E+1) with E taken from the frame's end value
  const double end = start + 480.0 * (double)F;
  const int E = (int)((__double_as_longlong(end) >> 52) & 0x7ff) - 1023;
  const int e = (int)((__float_as_uint(F) >> 23) & 0xff) - 127;
  if (F == 0.f) return true;
  if (((__float_as_uint(F) >> 23) & 0xff) == 0 || E - 52 > e - 23) return false;      // (subnormal F: take the sequential path)
  // start itself must sit on the end value's grid too (it does unless the frame crosses into a higher binade with start's
  // last bits set: the sequential additions would round there)
  const double q = scalbn(start, 52 - E);
  return q == trunc(q);
}

__global__ void sine_frame_scan_kernel(const float* __restrict__ f0, double* __restrict__ start, int B, int T) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * 9) return;
  const int b = idx / 9, h = idx - b * 9;
  const float mult = (float)(h + 1);
  double cum = 0.0;
  for (int t = 0; t < T; ++t) {
    start[(long)idx * T + t] = cum;
    const float F = f0[b * T + t] * mult / 24000.0f;
    if (frame_adds_exact(cum, F)) {
      cum = cum + 480.0 * (double)F;      // exact product (24 x 9 bits), exact sum
    } else {
      for (int k = 0; k < 480; ++k) cum = cum + (double)F;
    }
  }
}

// Philox4x32-10 (Salmon et al., SC'11): the counter-based generator behind the seeded source noise -- a sample's nine draws
// are a pure function of (seed, call, utterance, sample), so no 9 x 480 T noise tensor is written and read back
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// two uniform words -> two N(0,1) draws (Box-Muller; u1 in (0, 1] so the logarithm is finite)
__device__ __forceinline__ void box_muller(unsigned a, unsigned b, float& z0, float& z1) {
  const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);
  const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
  const float r = sqrtf(-2.0f * __logf(u1));
  float sn, cs;
  __sincosf(6.283185307179586f * u2, &sn, &cs);
  z0 = r * cs;
  z1 = r * sn;
}

// s[b,n] = tanh( lin_b + sum_h lin_w[h] * ( 0.1 sin(2 pi frac + phi_h) * uv + namp * noise ) )
// noise == nullptr: the nine N(0,1) draws of a sample come from Philox keyed by (seed_lo, seed_hi), counter (utterance,
// sample, call, 3 blocks) -- generator.py:171 draws them with torch.randn_like, whose values no caller can depend on
__global__ __launch_bounds__(256) void source_mix_kernel(const float* __restrict__ f0, const double* __restrict__ start,
                                                         const float* __restrict__ phase, const float* __restrict__ noise,
                                                         const float* __restrict__ lin_w, const float* __restrict__ lin_b,
                                                         float* __restrict__ s, int B, int T, unsigned seed_lo, unsigned seed_hi,
                                                         unsigned call) {
  const long n_per = (long)T * 480;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n_per * B) return;
  const int b = (int)(idx / n_per);
  const long n = idx - (long)b * n_per;
  const int t = (int)(n / 480), k = (int)(n - (long)t * 480);
  const float f = f0[b * T + t];
  const float uv = f > 10.0f ? 1.f : 0.f;
  const float namp = uv * 0.003f + (1.f - uv) * 0.1f / 3.f;
  float z[12];
  if (!noise) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      unsigned r[4];
      philox4x32_10((unsigned)n, (unsigned)b, call, (unsigned)j, seed_lo, seed_hi, r);
      box_muller(r[0], r[1], z[4 * j], z[4 * j + 1]);
      box_muller(r[2], r[3], z[4 * j + 2], z[4 * j + 3]);
    }
  }
  float acc = 0.f;
#pragma unroll
  for (int h = 0; h < 9; ++h) {
    const float F = f * (float)(h + 1) / 24000.0f;
    const double st = start[((long)b * 9 + h) * T + t];
    double cum;
    if (frame_adds_exact(st, F)) {
      cum = st + (double)(k + 1) * (double)F;
    } else {
      cum = st;
      for (int j = 0; j <= k; ++j) cum = cum + (double)F;
    }
    const float c32 = (float)cum;
    const float frac = c32 - floorf(c32);
    const float theta = 2.0f * 3.14159265358979323846f * frac;
    const float ph = h == 0 ? 0.f : phase[b * 9 + h];
    const float sine = 0.1f * sinf(theta + ph);
    acc += (sine * uv + namp * (noise ? noise[((long)b * 9 + h) * n_per + n] : z[h])) * lin_w[h];
  }
  s[idx] = tanhf(acc + lin_b[0]);
}

int sine_source(const float* f0, const float* phase, const float* noise, const float* lin_w, const float* lin_b, float* frac,
                float* s, int B, int T, hipStream_t st, unsigned long long seed, unsigned call) {
  // `frac` (sized [B, 9, 480 T] floats by hift_ws_create) now only holds the [B, 9, T] frame-start sums, as doubles
  double* start = reinterpret_cast<double*>(frac);
  hipLaunchKernelGGL(sine_frame_scan_kernel, dim3(cdiv(B * 9, 64)), dim3(64), 0, st, f0, start, B, T);
  hipLaunchKernelGGL(source_mix_kernel, dim3((unsigned)cdivl((long)B * T * 480, 256)), dim3(256), 0, st, f0, start, phase, noise,
                     lin_w, lin_b, s, B, T, (unsigned)seed, (unsigned)(seed >> 32), call);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// ---- STFT(16, hop 4, periodic Hann, center/reflect) of s -> row buffer [479 + b*S3 + tau][32] = 9 re | 9 im | 0 ----
// Every row of the buffer is written (zeros outside the utterance): the strided source convs read it unmasked.
__global__ __launch_bounds__(256) void stft_rows_kernel(const float* __restrict__ s, float* __restrict__ out,
                                                        const int* __restrict__ lens, int B, int T, int G3, int S3, long rows,
                                                        const int* __restrict__ uoff3) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  f32x4 o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  long rel = r - G3;
  long bl = 0, tau = 0;
  if (uoff3) {      // compact geometry (hift.hip): utterance b starts at row uoff3[b]; the last entry is the first row past the batch
    int lo = 0, hi = B - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (uoff3[mid] <= r) lo = mid; else hi = mid - 1;
    }
    bl = r < uoff3[B] ? lo : B;
    tau = r - uoff3[lo];
    rel = tau;      // (< 0 ahead of the first utterance)
  } else if (rel >= 0) {
    bl = rel / S3;      // rows past the batch exist whenever the context's capacity exceeds this call's batch:
    tau = rel - bl * S3;
  }
  if (rel >= 0) {
    const int b = bl < B ? (int)bl : B - 1;      // never index lens[] / s with them
    const int len = lens ? min(lens[b], T) : T;
    const long nvalid = (long)len * 480;
    if (bl < B && tau <= (long)len * 120 && nvalid > 0) {
      const float* sb = s + (long)b * T * 480;
      float x[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        long i = 4 * tau - 8 + k;
        if (i < 0) i = -i;
        if (i >= nvalid) i = 2 * (nvalid - 1) - i;
        x[k] = sb[i] * c_hann16[k];
      }
      float re[9], im[9];
#pragma unroll
      for (int f = 0; f < 9; ++f) {
        float a = 0.f, c = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          a += x[k] * c_cos16[(f * k) & 15];
          c -= x[k] * c_sin16[(f * k) & 15];
        }
        re[f] = a;
        im[f] = c;
      }
      o[0] = f32x4{re[0], re[1], re[2], re[3]};
      o[1] = f32x4{re[4], re[5], re[6], re[7]};
      o[2] = f32x4{re[8], im[0], im[1], im[2]};
      o[3] = f32x4{im[3], im[4], im[5], im[6]};
      o[4] = f32x4{im[7], im[8], 0.f, 0.f};
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(out + r * 32 + 4 * i) = o[i];
}

int stft_rows(const float* s, float* out, const int* lens, int B, int T, int G3, int S3, long rows, hipStream_t st, const int* uoff3) {
  hipLaunchKernelGGL(stft_rows_kernel, dim3((unsigned)cdivl(rows, 256)), dim3(256), 0, st, s, out, lens, B, T, G3, S3, rows, uoff3);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// ---- ReflectionPad1d((1,0)) after the last up-conv: row(tau=0) := row(tau=2) --------------------------------------
__global__ void reflect_fix_kernel(float* __restrict__ x, int B, int G3, int S3, int C, const int* __restrict__ uoff3) {
  const int b = blockIdx.x;
  float* base = x + (uoff3 ? (long)uoff3[b] : (long)G3 + (long)b * S3) * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) base[c] = base[2 * C + c];
}
int reflect_fix(float* x, int B, int G3, int S3, int C, hipStream_t st, const int* uoff3) {
  hipLaunchKernelGGL(reflect_fix_kernel, dim3(B), dim3(64), 0, st, x, B, G3, S3, C, uoff3);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// ---- iSTFT head: conv_post rows [.,32] (9 log-mag | 9 phase) -> windowed 16-sample frames -> overlap-add -----------
__global__ __launch_bounds__(256) void istft_frames_kernel(const float* __restrict__ post, float* __restrict__ frames,
                                                           long rows) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const float* p = post + r * 32;
  float re[9], im[9];
#pragma unroll
  for (int f = 0; f < 9; ++f) {
    const float mag = fminf(expf(p[f]), 100.0f);
    const float ph = sinf(p[9 + f]);
    re[f] = mag * cosf(ph);
    im[f] = mag * sinf(ph);
  }
  // irfft(16): x[k] = (1/16) [ Re X0 + (-1)^k Re X8 + 2 sum_{f=1..7} (Re X_f cos(2 pi f k/16) - Im X_f sin(2 pi f k/16)) ]
  f32x4 o[4];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    float a = re[0] + ((k & 1) ? -re[8] : re[8]);
#pragma unroll
    for (int f = 1; f < 8; ++f) a += 2.0f * (re[f] * c_cos16[(f * k) & 15] - im[f] * c_sin16[(f * k) & 15]);
    o[k >> 2][k & 3] = a * (1.0f / 16.0f) * c_hann16[k];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(frames + r * 16 + 4 * i) = o[i];
}

__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ frames, float* __restrict__ wav,
                                                        const int* __restrict__ lens, int B, int T, int G3, int S3,
                                                        float limit, const int* __restrict__ uoff3) {
  const long n_per = (long)T * 480;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n_per * B) return;
  const int b = (int)(idx / n_per);
  const long n = idx - (long)b * n_per;
  const int len = lens ? min(lens[b], T) : T;
  float v = 0.f;
  if (n < (long)len * 480) {
    const long ntau = (long)len * 120;   // frames tau = 0 .. ntau
    float num = 0.f, den = 0.f;
    const long t_hi = (n + 8) >> 2;      // tau with 0 <= n + 8 - 4 tau < 16
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long tau = t_hi - i;
      const int k = (int)(n + 8 - 4 * tau);
      if (tau >= 0 && tau <= ntau && k >= 0 && k < 16) {
        num += frames[((uoff3 ? (long)uoff3[b] : (long)G3 + (long)b * S3) + tau) * 16 + k];
        den += c_hann16[k] * c_hann16[k];
      }
    }
    v = num / den;
    v = fminf(fmaxf(v, -limit), limit);
  }
  wav[idx] = v;
}

int istft_head(const float* post, float* frames, float* wav, const int* lens, int B, int T, int G3, int S3, long rows,
               hipStream_t st, const int* uoff3) {
  hipLaunchKernelGGL(istft_frames_kernel, dim3((unsigned)cdivl(rows, 256)), dim3(256), 0, st, post, frames, rows);
  hipLaunchKernelGGL(istft_ola_kernel, dim3((unsigned)cdivl((long)B * T * 480, 256)), dim3(256), 0, st, frames, wav, lens, B, T,
                     G3, S3, 0.99f, uoff3);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

}  // namespace jv
