// Prompt-mel front-end on the GPU: `mel_spectrogram` of jyutvoice/utils/audio.py:18-63 with the parameters
// `extract_speech_feat` fixes (infer.py:166-186): 24 kHz, n_fft = win = 1920 (periodic Hann), hop 480, center=False after
// a reflect pad of (n_fft - hop) / 2 = 720 samples per side, 80 mel bands, log of the 1e-5 clamp.
//
// A 1920-point STFT every 480 samples is a [frames, 1920] x [1920, 2 * 961] GEMM against a windowed cos / -sin basis
// (conv_gemm, bf16x6: fp32-level accumulation over the 1920 samples), then |.| with the reference's 1e-9 floor, then the
// [961 -> 80] mel projection with the log in its epilogue.  The mel filterbank is data supplied by the host
// (jv_load_mel_basis): in the reference it comes from librosa, which is not part of this build.
#include <math.h>

#include "../../include/jyutvoice_hip.h"
#include "jv_model.h"
#include "jv_ops.h"

namespace jv {

int split3_planes(const float* src, unsigned short* dst, long n, hipStream_t st);   // registry.hip

constexpr int A_NFFT = 1920, A_HOP = 480, A_BINS = 961, A_PAD = 720, A_NMEL = 80;
constexpr int A_NDFT = 1924;    // 961 cos rows + 961 -sin rows, padded to a multiple of 4
constexpr int A_KMEL = 992;     // 961 magnitude bins padded to a multiple of 32 (the mel GEMM's K)

struct AudioWs {
  GemmW dft, mel;                 // [1924][1920] windowed DFT basis; [80][992] mel filterbank (both with bf16x6 planes)
  bool mel_loaded = false;
  long rows = 0;
  std::vector<void*> allocs, ws;
  float *frames = nullptr, *spec = nullptr, *mag = nullptr, *out = nullptr;
};

void audio_ws_destroy(Context& c) {
  if (!c.aws) return;
  for (void* p : c.aws->allocs) (void)hipFree(p);
  for (void* p : c.aws->ws) (void)hipFree(p);
  delete c.aws;
  c.aws = nullptr;
}

namespace {

// basis[k][n] = hann[n] * cos(2 pi k n / 1920) (k < 961), basis[961 + k][n] = -hann[n] * sin(2 pi k n / 1920); the angle is
// reduced with an exact integer modulus and evaluated in fp64, then rounded once
__global__ __launch_bounds__(256) void dft_basis_kernel(float* __restrict__ w) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)A_NDFT * A_NFFT) return;
  const int r = (int)(idx / A_NFFT), n = (int)(idx - (long)r * A_NFFT);
  float v = 0.f;
  if (r < 2 * A_BINS) {
    const int k = r < A_BINS ? r : r - A_BINS;
    const double hann = 0.5 - 0.5 * cos(2.0 * M_PI * (double)n / (double)A_NFFT);
    const double ang = 2.0 * M_PI * (double)((k * n) % A_NFFT) / (double)A_NFFT;
    v = (float)(r < A_BINS ? hann * cos(ang) : -hann * sin(ang));
  }
  w[idx] = v;
}

// frames[b*T + t][i] = padded_b[t*480 + i], padded = reflect pad of 720 samples per side (F.pad mode="reflect")
__global__ __launch_bounds__(256) void frame_rows_kernel(const float* __restrict__ wav, float* __restrict__ frames, int B,
                                                         int n, int T) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * T * A_NFFT) return;
  const int i = (int)(idx % A_NFFT);
  const long bt = idx / A_NFFT;
  const int b = (int)(bt / T), t = (int)(bt - (long)b * T);
  int j = t * A_HOP + i - A_PAD;
  if (j < 0) j = -j;
  if (j >= n) j = 2 * (n - 1) - j;
  frames[idx] = wav[(long)b * n + j];
}

// mag[row][k] = sqrt(re^2 + im^2 + 1e-9) for k < 961, zero for the K padding 961 <= k < 992
__global__ __launch_bounds__(256) void mag_rows_kernel(const float* __restrict__ spec, float* __restrict__ mag, long rows) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * A_KMEL) return;
  const int k = (int)(idx % A_KMEL);
  const long r = idx / A_KMEL;
  float v = 0.f;
  if (k < A_BINS) {
    const float re = spec[r * A_NDFT + k], im = spec[r * A_NDFT + A_BINS + k];
    v = sqrtf((re * re + im * im) + 1e-9f);
  }
  mag[idx] = v;
}

// [80][961] host/device filterbank -> [80][992] zero-padded rows
__global__ void pad_mel_kernel(const float* __restrict__ src, float* __restrict__ dst) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= A_NMEL * A_KMEL) return;
  const int m = idx / A_KMEL, k = idx - m * A_KMEL;
  dst[idx] = k < A_BINS ? src[m * A_BINS + k] : 0.f;
}

int dev_alloc(AudioWs& w, std::vector<void*>& list, void** p, size_t bytes) {
  JV_HIP(hipMalloc(p, bytes));
  list.push_back(*p);
  return JV_OK;
}

int attach_planes(AudioWs& w, GemmW& g, hipStream_t st) {
  const long n = (long)g.n_rows * g.ldw;
  void* d = nullptr;
  JV_TRY(dev_alloc(w, w.allocs, &d, (size_t)3 * n * sizeof(unsigned short) + 64));
  JV_TRY(split3_planes(g.w, static_cast<unsigned short*>(d), n, st));
  g.w3 = static_cast<const unsigned short*>(d);
  return JV_OK;
}

AudioWs& get(Context& c) {
  if (!c.aws) c.aws = new AudioWs();
  return *c.aws;
}

}  // namespace

int load_mel_basis(Context& c, const float* data, bool on_device, hipStream_t st) {
  AudioWs& w = get(c);
  void *tmp = nullptr, *padded = nullptr;
  JV_TRY(dev_alloc(w, w.allocs, &tmp, sizeof(float) * A_NMEL * A_BINS));
  JV_HIP(hipMemcpyAsync(tmp, data, sizeof(float) * A_NMEL * A_BINS, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
  JV_TRY(dev_alloc(w, w.allocs, &padded, sizeof(float) * A_NMEL * A_KMEL));
  hipLaunchKernelGGL(pad_mel_kernel, dim3(cdiv(A_NMEL * A_KMEL, 256)), dim3(256), 0, st, static_cast<const float*>(tmp),
                     static_cast<float*>(padded));
  w.mel = GemmW{};
  w.mel.w = static_cast<const float*>(padded); w.mel.ldw = A_KMEL; w.mel.n_rows = A_NMEL; w.mel.N = A_NMEL;
  w.mel.Cin = A_KMEL; w.mel.ntaps = 1;
  JV_TRY(attach_planes(w, w.mel, st));
  JV_HIP(hipStreamSynchronize(st));
  w.mel_loaded = true;
  return JV_OK;
}

int mel_spectrogram(Context& c, const float* wav, int B, int n, float* mel, hipStream_t st) {
  AudioWs& w = get(c);
  if (!w.mel_loaded) return fail(JV_ERR_STATE, "mel filterbank not loaded (jv_load_mel_basis)");
  if (B < 1 || n <= A_PAD || n < A_NFFT - 2 * A_PAD)
    return fail(JV_ERR_ARG, "jv_mel_spectrogram: need more than 720 samples per utterance (reflect padding)");
  const int T = 1 + (n + 2 * A_PAD - A_NFFT) / A_HOP;
  const long rows = (long)B * T;
  if (rows > (1L << 22)) return fail(JV_ERR_SHAPE, "jv_mel_spectrogram: too many frames");
  if (!w.dft.w) {      // first call: build the windowed DFT basis and its bf16x6 planes
    void* d = nullptr;
    JV_TRY(dev_alloc(w, w.allocs, &d, sizeof(float) * A_NDFT * A_NFFT));
    hipLaunchKernelGGL(dft_basis_kernel, dim3((unsigned)cdivl((long)A_NDFT * A_NFFT, 256)), dim3(256), 0, st,
                       static_cast<float*>(d));
    w.dft.w = static_cast<const float*>(d); w.dft.ldw = A_NFFT; w.dft.n_rows = A_NDFT; w.dft.N = A_NDFT; w.dft.Cin = A_NFFT;
    w.dft.ntaps = 1;
    JV_TRY(attach_planes(w, w.dft, st));
  }
  if (rows > w.rows) {
    JV_HIP(hipDeviceSynchronize());
    for (void* p : w.ws) (void)hipFree(p);
    w.ws.clear();
    const size_t R = (size_t)round_up((int)rows, 128) + 128;
    JV_TRY(dev_alloc(w, w.ws, reinterpret_cast<void**>(&w.frames), R * A_NFFT * sizeof(float)));
    JV_TRY(dev_alloc(w, w.ws, reinterpret_cast<void**>(&w.spec), R * A_NDFT * sizeof(float)));
    JV_TRY(dev_alloc(w, w.ws, reinterpret_cast<void**>(&w.mag), R * A_KMEL * sizeof(float)));
    JV_TRY(dev_alloc(w, w.ws, reinterpret_cast<void**>(&w.out), R * A_NMEL * sizeof(float)));
    w.rows = (long)R;
  }
  hipLaunchKernelGGL(frame_rows_kernel, dim3((unsigned)cdivl(rows * A_NFFT, 256)), dim3(256), 0, st, wav, w.frames, B, n, T);
  ConvGemmArgs a;
  conv_gemm_defaults(a);
  a.A = w.frames; a.lda = A_NFFT; a.a_rows = rows; a.M = (int)rows; a.Cin = A_NFFT; a.ntaps = 1;
  a.W = w.dft.w; a.W3 = w.dft.w3; a.w3_plane = (long)A_NDFT * A_NFFT; a.ldw = A_NFFT; a.n_rows_w = A_NDFT; a.N = A_NDFT;
  a.out = w.spec; a.ldo = A_NDFT;
  JV_TRY(conv_gemm(a, 1, st));
  hipLaunchKernelGGL(mag_rows_kernel, dim3((unsigned)cdivl(rows * A_KMEL, 256)), dim3(256), 0, st, w.spec, w.mag, rows);
  conv_gemm_defaults(a);
  a.A = w.mag; a.lda = A_KMEL; a.a_rows = rows; a.M = (int)rows; a.Cin = A_KMEL; a.ntaps = 1;
  a.W = w.mel.w; a.W3 = w.mel.w3; a.w3_plane = (long)A_NMEL * A_KMEL; a.ldw = A_KMEL; a.n_rows_w = A_NMEL; a.N = A_NMEL;
  a.out = w.out; a.ldo = A_NMEL;
  a.act = ACT_LOGCLIP;
  JV_TRY(conv_gemm(a, 1, st));
  JV_HIP(hipGetLastError());
  return rows_to_cf(w.out, A_NMEL, 0, 0, T, mel, (long)A_NMEL * T, B, A_NMEL, T, nullptr, st);
}

}  // namespace jv

extern "C" {

int jv_load_mel_basis(jv_context* ctx, const float* data, int64_t numel, int on_device, void* stream) {
  if (!ctx || !data) return jv::fail(JV_ERR_ARG, "jv_load_mel_basis: null argument");
  if (numel != (int64_t)jv::A_NMEL * jv::A_BINS) return jv::fail(JV_ERR_SHAPE, "jv_load_mel_basis: expected 80*961 floats");
  JV_HIP(hipSetDevice(ctx->c.device));
  return jv::load_mel_basis(ctx->c, data, on_device != 0, static_cast<hipStream_t>(stream));
}

int jv_mel_spectrogram(jv_context* ctx, const float* wav, int B, int n_samples, float* mel, void* stream) {
  if (!ctx || !wav || !mel) return jv::fail(JV_ERR_ARG, "jv_mel_spectrogram: null argument");
  JV_HIP(hipSetDevice(ctx->c.device));
  return jv::mel_spectrogram(ctx->c, wav, B, n_samples, mel, static_cast<hipStream_t>(stream));
}

}  // extern "C"
