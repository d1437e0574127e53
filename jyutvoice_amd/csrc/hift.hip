// HiFT vocoder stage (jyutvoice/hifigan/generator.py:396-466, f0_predictor.py:52-55) on row buffers.
//
// Four time resolutions share one geometry derived from the mel level (guard G0 = gap = 4 frames):
//   level 0 (mel, T)      rows r0            = 4   + b*(T+4) + t
//   level 1 (x8)          rows 8  r0 + phase = 32  + b*8(T+4)   + t
//   level 2 (x40)         rows 40 r0 + phase = 160 + b*40(T+4)  + t
//   level 3 (x120, +1)    rows               = 479 + b*120(T+4) + tau      (tau = t+1 after ReflectionPad1d((1,0)))
// so a polyphase ConvTranspose1d maps input row r to output rows s*r .. s*r+s-1 with no index arithmetic, the
// strided source convs are GEMMs over contiguous spans of the 32-column STFT row buffer (row stride 15*32 / 3*32), and
// the gaps (>= 25 rows at every level) are the zero padding of every dilated ResBlock convolution.
// Snake / LeakyReLU run in the conv kernel's prologue, bias / residual / x+si fusion / MRF mean in its epilogue.
#include <algorithm>

#include "../../include/jyutvoice_hip.h"
#include "hiftconv_kernel.h"
#include "hiftpair_kernel.h"
#include "jv_model.h"
#include "jv_ops.h"

namespace jv {

int hiftconv(const HiftConvArgs& a, int C, hipStream_t st);      // hiftconv.hip
int hiftpair(const HiftPairArgs& a, int C, hipStream_t st);

int f0_head(const float* h, const float* w, const float* bias, float* f0, int B, int T, int G, int S, hipStream_t st);
int sine_source(const float* f0, const float* phase, const float* noise, const float* lin_w, const float* lin_b, float* frac,
                float* s, int B, int T, hipStream_t st, unsigned long long seed, unsigned call);
// (uoff3: optional first rows of the utterances at level 3 -- the compact geometry of ragged batches, below)
int stft_rows(const float* s, float* out, const int* lens, int B, int T, int G3, int S3, long rows, hipStream_t st, const int* uoff3 = nullptr);
int reflect_fix(float* x, int B, int G3, int S3, int C, hipStream_t st, const int* uoff3 = nullptr);
int istft_head(const float* post, float* frames, float* wav, const int* lens, int B, int T, int G3, int S3, long rows,
               hipStream_t st, const int* uoff3 = nullptr);

constexpr int H_G0 = 4, H_GAP0 = 4;
constexpr int UPS[3] = {8, 5, 3};
constexpr int LVL_MUL[4] = {1, 8, 40, 120};

struct HiftWs {
  long rows0_alloc = 0;
  float* mel = nullptr;                      // [rows0, 96]
  float *f0a = nullptr, *f0b = nullptr;      // [rows0, 512]
  float* x0 = nullptr;                       // [rows0, 512] conv_pre output
  float *x[3] = {}, *r[3] = {}, *tmp[3] = {}, *xs[3] = {}, *si[3] = {};   // per level 1..3: [rows_l, C_l]
  float* stft_alloc = nullptr;               // [16 + rows3 + 16, 32]
  float* post = nullptr;                     // [rows3, 32]
  float* frames = nullptr;                   // [rows3, 16]
  float* frac = nullptr;                     // [B, 9, 480 T]
  unsigned char* mask[4] = {};
  int* lens = nullptr;
  // COMPACT geometry of ragged batches (hift_decode): utterance b starts at mel-level row u0[b] = G0 + sum_{b' < b} (len + gap) and,
  // every level being the mel level's rows times its factor, at row mul_l u0[b] (- 1 at level 3) of level l.  uoff[l]: those first
  // rows (+ the first row past the batch), [max_batch + 1] ints per level; rsample[l]: the utterance of every row (row_meta), what
  // the fp16x3 kernels index their per-utterance bounds with; h_lens / h_uoff: pinned host staging
  int* uoff[4] = {};
  int* rsample[4] = {};
  int *h_lens = nullptr, *h_uoff = nullptr;
  // max |value| ever written to a buffer during the current decode, per buffer and per UTTERANCE ([A_SLOTS][max_batch]
  // device floats, zeroed at its start; conv_gemm's amax_out): the measured bound from which the consuming fp16x3
  // convolution derives its scale (amax_in) -- an utterance's waveform does not depend on its batch neighbours
  float* amax = nullptr;
  int amax_stride = 0;      // = max_batch
  enum Slot { A_X0 = 0, A_SI = 1, A_R = 4, A_TMP = 7, A_XS = 10, A_X = 13, A_SLOTS = 16 };
};

int hift_ws_create(Context& c) {
  HiftWs* w = new HiftWs();
  c.hws = w;
  const long rows0 = H_G0 + (long)c.max_batch * (c.max_frames + H_GAP0);
  w->rows0_alloc = rows0 + 8;
  const size_t R0 = (size_t)w->rows0_alloc;
  auto F = [&](float** p, size_t floats) { return ws_alloc(c, floats * sizeof(float), reinterpret_cast<void**>(p)); };
  JV_TRY(F(&w->mel, R0 * 96));
  JV_TRY(F(&w->f0a, R0 * 512));
  JV_TRY(F(&w->f0b, R0 * 512));
  JV_TRY(F(&w->x0, R0 * 512));
  for (int l = 0; l < 3; ++l) {
    const size_t R = R0 * LVL_MUL[l + 1];
    const int C = HIFT_CH >> (l + 1);
    JV_TRY(F(&w->x[l], R * C));
    JV_TRY(F(&w->r[l], R * C));
    JV_TRY(F(&w->tmp[l], R * C));
    JV_TRY(F(&w->xs[l], R * C));
    JV_TRY(F(&w->si[l], R * C));
  }
  const size_t R3 = R0 * 120;
  JV_TRY(F(&w->stft_alloc, (R3 + 32) * 32));
  JV_TRY(F(&w->post, R3 * 32));
  JV_TRY(F(&w->frames, R3 * 16));
  JV_TRY(F(&w->frac, (size_t)c.max_batch * 9 * 480 * c.max_frames));
  for (int l = 0; l < 4; ++l) JV_TRY(ws_alloc(c, R0 * LVL_MUL[l], reinterpret_cast<void**>(&w->mask[l])));
  JV_TRY(ws_alloc(c, sizeof(int) * c.max_batch, reinterpret_cast<void**>(&w->lens)));
  for (int l = 0; l < 4; ++l) {
    JV_TRY(ws_alloc(c, sizeof(int) * (c.max_batch + 1), reinterpret_cast<void**>(&w->uoff[l])));
    JV_TRY(ws_alloc(c, sizeof(int) * R0 * LVL_MUL[l], reinterpret_cast<void**>(&w->rsample[l])));
  }
  JV_HIP(hipHostMalloc(reinterpret_cast<void**>(&w->h_lens), sizeof(int) * c.max_batch, hipHostMallocDefault));
  JV_HIP(hipHostMalloc(reinterpret_cast<void**>(&w->h_uoff), sizeof(int) * 4 * (c.max_batch + 1), hipHostMallocDefault));
  w->amax_stride = c.max_batch;
  JV_TRY(ws_alloc(c, sizeof(float) * HiftWs::A_SLOTS * c.max_batch, reinterpret_cast<void**>(&w->amax)));
  return JV_OK;
}

void hift_ws_destroy(Context& c) {
  if (c.hws) {
    if (c.hws->h_lens) (void)hipHostFree(c.hws->h_lens);
    if (c.hws->h_uoff) (void)hipHostFree(c.hws->h_uoff);
  }
  delete c.hws;
  c.hws = nullptr;
}

namespace {

struct HGeo {
  int B, T, S0;
  long rows[4];      // rows computed at each level
  long alloc[4];     // rows readable at each level
  int G[4], S[4], L[4];
  const int* uoff[4] = {nullptr, nullptr, nullptr, nullptr};      // compact geometry: first rows per level (device); null: uniform
  const int* rsample[4] = {nullptr, nullptr, nullptr, nullptr};   // ... and the row -> utterance tables
  long frames = 0;                                                // profiler: real mel frames of the call (0: B * T)
};

HGeo make_geo(const Context& c, int B, int T) {
  HGeo g;
  g.B = B; g.T = T; g.S0 = T + H_GAP0;
  const long rows0 = H_G0 + (long)B * g.S0;
  for (int l = 0; l < 4; ++l) {
    g.rows[l] = rows0 * LVL_MUL[l];
    g.alloc[l] = c.hws->rows0_alloc * LVL_MUL[l];
    g.S[l] = g.S0 * LVL_MUL[l];
    g.G[l] = H_G0 * LVL_MUL[l];
    g.L[l] = T * LVL_MUL[l];
  }
  g.G[3] = 479;          // tau = 0 sits one row before the natural x120 position
  g.L[3] = 120 * T + 1;
  return g;
}

int check(Context& c, int B, int T) {
  if (!c.ready[MODEL_HIFT]) return fail(JV_ERR_STATE, "hift weights not finalized");
  if (B < 1 || T < 1) return fail(JV_ERR_ARG, "batch and frame count must be positive");
  if (B > c.max_batch || T > c.max_frames) return fail(JV_ERR_SHAPE, "batch/frames exceed the capacity given to jv_create");
  return JV_OK;
}

ConvGemmArgs conv_args(const float* A, int lda, long a_rows, long M, const GemmW& w, float* out, int ldo, int tap_row0,
                       int dil, const unsigned char* mask_in) {
  ConvGemmArgs a;
  conv_gemm_defaults(a);
  a.A = A; a.lda = lda; a.a_rows = a_rows; a.M = (int)M;
  a.Cin = w.Cin; a.ntaps = w.ntaps; a.tap_row0 = tap_row0; a.tap_dil = dil;
  a.W = w.w; a.ldw = w.ldw; a.n_rows_w = w.n_rows; a.N = w.N; a.bias = w.bias;
  a.W3 = w.w3; a.w3_plane = (long)w.n_rows * w.ldw;
  a.out = out; a.ldo = ldo;
  a.rowmask_in = mask_in;
  return a;
}

// algorithmic size of a launch for the profiler: B*frames real rows, real taps*channels
void set_alg(ConvGemmArgs& a, long rows, int k) {
  a.alg_rows = rows;
  a.alg_k = k;
}

// lens (device int32 [B] or null) -> ws.lens + the four row masks
int prepare_masks(Context& c, const HGeo& g, const int* lens, hipStream_t st) {
  HiftWs& w = *c.hws;
  if (lens) JV_HIP(hipMemcpyAsync(w.lens, lens, sizeof(int) * g.B, hipMemcpyDeviceToDevice, st));
  else JV_TRY(fill_int(w.lens, g.T, g.B, st));
  for (int l = 0; l < 4; ++l)
    JV_TRY(row_meta(w.mask[l], g.uoff[l] ? w.rsample[l] : nullptr, w.lens, g.B, 1, g.G[l], g.S[l], g.L[l], g.alloc[l], LVL_MUL[l],
                    l == 3 ? 1 : 0, st, g.uoff[l]));
  return JV_OK;
}

int mel_to_rows(Context& c, const HGeo& g, const float* mel, hipStream_t st) {
  return cf_to_rows(mel, 80L * g.T, g.T, g.B, 80, g.T, c.hws->mel, 96, 0, H_G0, g.S0, 1.f, c.hws->lens, st, g.uoff[0]);
}

// the per-utterance slot geometry of the rows at `lvl` (ConvGemmArgs::amax_G/S/nb); rows the level's mask marks as
// padding are not tracked
void amax_geo(ConvGemmArgs& a, const HGeo& g, int lvl, const unsigned char* mask) {
  a.amax_G = g.G[lvl]; a.amax_S = g.S[lvl]; a.amax_nb = g.B; a.amax_mask = mask;
  a.amax_rows = g.rsample[lvl];      // compact geometry: a row's utterance by table
}

// fp16x3 on a convolution whose input bound is measured (am_in) and whose prologue adds at most `extra` to |x|
void h3_measured(ConvGemmArgs& a, const GemmW& w, const float* am_in, float extra, bool on) {
  if (!on || !w.w2 || !am_in) return;
  a.W2 = w.w2; a.w2_plane = (long)w.n_rows * w.ldw; a.colscale = w.colscale;
  a.amax_in = am_in; a.a_extra = extra;
}

// ResBlock (generator.py:90-97) of kernel k on [rows, C]: cur -> (result scaled/accumulated into dst, + extra residual)
// am_*: the amax slots of cur, r, tmp and dst (HiftWs::amax); h3: fp16x3 allowed (not exact-range mode)
int resblock(const ResBlockW& rb, const HGeo& g, int lvl, int C, long rows, long alloc, const unsigned char* mask, const float* cur,
             float* r, float* tmp, float* dst, const float* extra_res, float scale, int accumulate, float* am_cur, float* am_r,
             float* am_tmp, float* am_dst, bool h3, bool rowconv, bool pair, hipStream_t st) {
  const int dils[3] = {1, 3, 5};
  const float* in = cur;
  // The pair of convolutions of each dilation in ONE launch (hiftpair_kernel.h) at 64 / 128 channels.  A workgroup reads its
  // neighbours' rows as halo, so a pair never writes the buffer it reads: cur -> r -> tmp -> dst (the unfused form's
  // intermediate buffer is free), each buffer with its own bound slots.
  bool all_pair = pair && rowconv && h3 && (C == 64 || C == 128);
  for (int j = 0; j < 3; ++j) all_pair = all_pair && rb.wfp[j] && rb.e1[j] > 0.f && rb.e2[j] > 0.f && rb.l1max[j] > 0.f;
  if (all_pair) {
    float* const bufs[3] = {r, tmp, dst};
    float* const ams[3] = {am_r, am_tmp, am_dst};
    float* am_in = am_cur;
    for (int j = 0; j < 3; ++j) {
      const bool last = j == 2;
      HiftPairArgs a{};
      a.A = in; a.a_rows = alloc; a.M = (int)rows; a.ntaps = rb.k; a.dil = dils[j]; a.rowmask = mask;
      a.alpha1 = rb.a1[j]; a.alpha2 = rb.a2[j];
      a.Wf = rb.wfp[j]; a.wf_plane = (long)C * 2 * rb.k * C;
      a.cs1 = rb.c1[j].colscale; a.b1 = rb.c1[j].bias; a.cs2 = rb.c2[j].colscale; a.b2 = rb.c2[j].bias;
      a.amax_in = am_in; a.e1 = rb.e1[j]; a.e2 = rb.e2[j]; a.l1max = rb.l1max[j]; a.b1max = rb.b1max[j];
      a.slot_G = g.G[lvl]; a.slot_S = g.S[lvl]; a.slot_nb = g.B; a.slot_map = g.rsample[lvl];
      a.out = bufs[j]; a.res2 = last ? extra_res : nullptr; a.out_scale = last ? scale : 1.f; a.accumulate = last ? accumulate : 0;
      a.amax_out = ams[j];
      a.alg_rows = g.frames ? g.frames * LVL_MUL[lvl] : (long)g.B * g.L[lvl];
      JV_TRY(hiftpair(a, C, st));
      in = bufs[j];
      am_in = ams[j];
    }
    return JV_OK;
  }
  // the row-owning form (hiftconv_kernel.h): the whole Snake'd window in LDS, weights in fragment order
  auto rc = [&](const GemmW& w, const float* A, const float* alpha, float extra, float* am_in, int tap_row0, int dil, float* out,
                const float* res1, const float* res2, float out_scale, int acc, float* am_out) -> int {
    HiftConvArgs a{};
    a.A = A; a.a_rows = alloc; a.M = (int)rows; a.ntaps = w.ntaps; a.dil = dil; a.tap_row0 = tap_row0; a.rowmask_in = mask;
    a.alpha = alpha; a.Wf = w.wf; a.wf_plane = (long)w.N * w.ntaps * w.Cin; a.colscale = w.colscale; a.bias = w.bias;
    a.amax_in = am_in; a.a_extra = extra; a.slot_G = g.G[lvl]; a.slot_S = g.S[lvl]; a.slot_nb = g.B; a.slot_map = g.rsample[lvl];
    a.out = out; a.res1 = res1; a.res2 = res2; a.out_scale = out_scale; a.accumulate = acc;
    a.amax_out = am_out; a.amax_mask = mask;
    a.alg_rows = g.frames ? g.frames * LVL_MUL[lvl] : (long)g.B * g.L[lvl];
    return hiftconv(a, C, st);
  };
  for (int j = 0; j < 3; ++j) {
    const int k = rb.k, d = dils[j];
    if (rowconv && h3 && rb.c1[j].wf && rb.c2[j].wf && rb.e1[j] > 0.f && rb.e2[j] > 0.f) {
      const bool last = j == 2;
      JV_TRY(rc(rb.c1[j], in, rb.a1[j], rb.e1[j], j == 0 ? am_cur : am_r, -(d * (k - 1) / 2), d, tmp, nullptr, nullptr, 1.f, 0, am_tmp));
      JV_TRY(rc(rb.c2[j], tmp, rb.a2[j], rb.e2[j], am_tmp, -((k - 1) / 2), 1, last ? dst : r, in, last ? extra_res : nullptr,
                last ? scale : 1.f, last ? accumulate : 0, last ? am_dst : am_r));
      in = r;
      continue;
    }
    ConvGemmArgs a = conv_args(in, C, alloc, rows, rb.c1[j], tmp, C, -(d * (k - 1) / 2), d, mask);
    a.pro = PRO_SNAKE; a.pro_alpha = rb.a1[j];
    h3_measured(a, rb.c1[j], j == 0 ? am_cur : am_r, rb.e1[j], h3 && rb.e1[j] > 0.f);
    a.amax_out = h3 ? am_tmp : nullptr;
    amax_geo(a, g, lvl, mask);
    JV_TRY(conv_gemm(a, 1, st));
    const bool last = j == 2;
    a = conv_args(tmp, C, alloc, rows, rb.c2[j], last ? dst : r, C, -((k - 1) / 2), 1, mask);
    a.pro = PRO_SNAKE; a.pro_alpha = rb.a2[j];
    a.res1 = in; a.ldr1 = C;
    if (last) {
      a.res2 = extra_res; a.ldr2 = C;
      a.out_scale = scale;
      a.accumulate = accumulate;
    }
    h3_measured(a, rb.c2[j], am_tmp, rb.e2[j], h3 && rb.e2[j] > 0.f);
    a.amax_out = h3 ? (last ? am_dst : am_r) : nullptr;
    amax_geo(a, g, lvl, mask);
    JV_TRY(conv_gemm(a, 1, st));
    in = r;
  }
  return JV_OK;
}

}  // namespace

int hift_f0(Context& c, const float* mel, const int* lens, int B, int T, float* f0, hipStream_t st) {
  JV_TRY(check(c, B, T));
  HiftWs& w = *c.hws;
  const HGeo g = make_geo(c, B, T);
  JV_TRY(prepare_masks(c, g, lens, st));
  JV_TRY(mel_to_rows(c, g, mel, st));
  const float* in = w.mel;
  int lda = 96;
  float* bufs[2] = {w.f0a, w.f0b};
  for (int i = 0; i < 5; ++i) {
    ConvGemmArgs a = conv_args(in, lda, g.alloc[0], g.rows[0], c.hift.f0_conv[i], bufs[i & 1], 512, -1, 1, w.mask[0]);
    a.act = ACT_ELU;
    JV_TRY(conv_gemm(a, 1, st));
    in = bufs[i & 1];
    lda = 512;
  }
  return f0_head(in, c.hift.f0_cls_w, c.hift.f0_cls_b, f0, B, T, H_G0, g.S0, st);
}

// noise == nullptr: the N(0,1) draws are generated inside source_mix from (seed, call) -- hiftops.hip
int hift_source(Context& c, const float* f0, const float* phase, const float* noise, int B, int T, float* s, hipStream_t st,
                unsigned long long seed = 0, unsigned call = 0) {
  JV_TRY(check(c, B, T));
  return sine_source(f0, phase, noise, c.hift.src_lin_w, c.hift.src_lin_b, c.hws->frac, s, B, T, st, seed, call);
}

int hift_decode(Context& c, const float* mel, const float* s, const int* lens, int B, int T, float* wav, hipStream_t st) {
  JV_TRY(check(c, B, T));
  HiftWs& w = *c.hws;
  const HiftW& h = c.hift;
  HGeo g = make_geo(c, B, T);
  // Ragged batch: the COMPACT geometry (HiftWs::uoff) when it saves at least 8 % of the rows -- every level's rows are the mel
  // level's times a factor, so laying the utterances end to end at the mel level lays them end to end everywhere.  The lengths
  // come down once (B ints, one synchronisation per decode); every per-row sum is the uniform geometry's: the same bits.
  if (lens && B > 1 && !c.no_compact) {
    JV_HIP(hipMemcpyAsync(w.h_lens, lens, sizeof(int) * B, hipMemcpyDeviceToHost, st));
    JV_HIP(hipStreamSynchronize(st));
    long r = H_G0, frames = 0;
    int* const hu = w.h_uoff;
    for (int b = 0; b <= B; ++b) {
      for (int l = 0; l < 4; ++l) hu[l * (B + 1) + b] = (int)(r * LVL_MUL[l]) - (l == 3 ? 1 : 0);
      if (b < B) {
        const int len = std::min(std::max(w.h_lens[b], 0), T);
        r += len + H_GAP0;
        frames += len;
      }
    }
    if (r * 100 <= (H_G0 + (long)B * g.S0) * 92) {
      for (int l = 0; l < 4; ++l) {
        JV_HIP(hipMemcpyAsync(w.uoff[l], hu + l * (B + 1), sizeof(int) * (B + 1), hipMemcpyHostToDevice, st));      // (pinned: stays valid)
        g.uoff[l] = w.uoff[l];
        g.rsample[l] = w.rsample[l];
        g.rows[l] = r * LVL_MUL[l];
      }
      g.frames = frames;
    }
  }
  JV_TRY(prepare_masks(c, g, lens, st));
  JV_TRY(mel_to_rows(c, g, mel, st));
  float* stft = w.stft_alloc + 16 * 32;
  JV_TRY(stft_rows(s, stft, w.lens, B, T, g.G[3], g.S[3], g.alloc[3], st, g.uoff[3]));
  JV_HIP(hipMemsetAsync(w.amax, 0, sizeof(float) * HiftWs::A_SLOTS * w.amax_stride, st));
  const bool h3 = !c.exact_range;
  auto am = [&](int slot) { return w.amax + (long)slot * w.amax_stride; };

  // conv_pre (k7, pad 3)
  {
    ConvGemmArgs a = conv_args(w.mel, 96, g.alloc[0], g.rows[0], h.conv_pre, w.x0, 512, -3, 1, w.mask[0]);
    a.amax_out = h3 ? am(HiftWs::A_X0) : nullptr;
    amax_geo(a, g, 0, w.mask[0]);
    JV_TRY(conv_gemm(a, 1, st));
  }
  const float* prev = w.x0;
  float* am_prev = am(HiftWs::A_X0);
  int prevC = 512;
  const int sd_stride[3] = {15, 3, 1}, sd_off[3] = {-8, -2, 0};
  for (int i = 0; i < 3; ++i) {
    const int l = i + 1, C = HIFT_CH >> l;
    // leaky_relu(0.1) -> ConvTranspose1d as a 3-tap polyphase GEMM: input row r -> output rows UPS*r .. UPS*r+UPS-1
    {
      ConvGemmArgs a = conv_args(prev, prevC, g.alloc[i], g.rows[i], h.ups[i], w.x[i], UPS[i] * C, -1, 1, w.mask[i]);
      a.pro = PRO_LRELU; a.pro_slope = 0.1f;
      h3_measured(a, h.ups[i], am_prev, 0.f, h3);      // |leaky_relu(x)| <= |x|
      a.amax_out = h3 ? am(HiftWs::A_X + i) : nullptr;
      amax_geo(a, g, i, w.mask[i]);      // rows of the INPUT level: output row r carries the UPS[i] frames it expands to
      JV_TRY(conv_gemm(a, 1, st));
      if (i == 2) JV_TRY(reflect_fix(w.x[i], B, g.G[3], g.S[3], C, st, g.uoff[3]));
    }
    // source branch: strided conv of the STFT rows, then a ResBlock whose last layer also adds the up-sampled trunk
    {
      ConvGemmArgs a = conv_args(stft + sd_off[i] * 32, sd_stride[i] * 32, g.rows[l], g.rows[l], h.src_down[i], w.si[i], C,
                                 0, 1, nullptr);
      a.amax_out = h3 ? am(HiftWs::A_SI + i) : nullptr;
      amax_geo(a, g, l, w.mask[l]);
      JV_TRY(conv_gemm(a, 1, st));
      JV_TRY(resblock(h.src_rb[i], g, l, C, g.rows[l], g.alloc[l], w.mask[l], w.si[i], w.r[i], w.tmp[i], w.xs[i], w.x[i], 1.f, 0,
                      am(HiftWs::A_SI + i), am(HiftWs::A_R + i), am(HiftWs::A_TMP + i), am(HiftWs::A_XS + i), h3, !c.no_hiftconv, !c.no_hift_pair, st));
    }
    // x = xs now holds x_up + si ; MRF: mean of the three ResBlocks, accumulated into w.x[i]
    for (int j = 0; j < 3; ++j)
      JV_TRY(resblock(h.rb[3 * i + j], g, l, C, g.rows[l], g.alloc[l], w.mask[l], w.xs[i], w.r[i], w.tmp[i], w.x[i], nullptr,
                      1.f / 3.f, j > 0 ? 1 : 0, am(HiftWs::A_XS + i), am(HiftWs::A_R + i), am(HiftWs::A_TMP + i),
                      am(HiftWs::A_X + i), h3, !c.no_hiftconv, !c.no_hift_pair, st));
    prev = w.x[i];
    am_prev = am(HiftWs::A_X + i);
    prevC = C;
  }
  // leaky_relu (default slope 0.01) -> conv_post (k7) -> exp / sin -> iSTFT -> clamp
  {
    ConvGemmArgs a = conv_args(prev, 64, g.alloc[3], g.rows[3], h.conv_post, w.post, 32, -3, 1, w.mask[3]);
    a.pro = PRO_LRELU; a.pro_slope = 0.01f;
    JV_TRY(conv_gemm(a, 1, st));
  }
  return istft_head(w.post, w.frames, wav, w.lens, B, T, g.G[3], g.S[3], g.rows[3], st, g.uoff[3]);
}

}  // namespace jv


extern "C" {

int jv_hift_f0(jv_context* ctx, const float* mel, const int32_t* lens, int B, int T, float* f0, void* stream) {
  if (!ctx || !mel || !f0) return jv::fail(JV_ERR_ARG, "jv_hift_f0: null argument");
  JV_HIP(hipSetDevice(ctx->c.device));
  return jv::hift_f0(ctx->c, mel, lens, B, T, f0, static_cast<hipStream_t>(stream));
}

int jv_hift_source(jv_context* ctx, const float* f0, const float* phase, const float* noise, int B, int T, float* s,
                   void* stream) {
  if (!ctx || !f0 || !phase || !noise || !s) return jv::fail(JV_ERR_ARG, "jv_hift_source: null argument");
  JV_HIP(hipSetDevice(ctx->c.device));
  return jv::hift_source(ctx->c, f0, phase, noise, B, T, s, static_cast<hipStream_t>(stream));
}

int jv_hift_source_seeded(jv_context* ctx, const float* f0, const float* phase, uint64_t seed, uint32_t call, int B, int T, float* s,
                          void* stream) {
  if (!ctx || !f0 || !phase || !s) return jv::fail(JV_ERR_ARG, "jv_hift_source_seeded: null argument");
  JV_HIP(hipSetDevice(ctx->c.device));
  return jv::hift_source(ctx->c, f0, phase, nullptr, B, T, s, static_cast<hipStream_t>(stream), seed, call);
}

int jv_hift_decode(jv_context* ctx, const float* mel, const float* s, const int32_t* lens, int B, int T, float* wav,
                   void* stream) {
  if (!ctx || !mel || !s || !wav) return jv::fail(JV_ERR_ARG, "jv_hift_decode: null argument");
  JV_HIP(hipSetDevice(ctx->c.device));
  return jv::hift_decode(ctx->c, mel, s, lens, B, T, wav, static_cast<hipStream_t>(stream));
}

}  // extern "C"
