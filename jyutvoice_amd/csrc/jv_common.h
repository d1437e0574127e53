// Shared declarations for libjyutvoice_hip.so (gfx950 / CDNA4 only).
//
// Data layout used by every kernel in this library ("row buffers"):
//   an activation is a row-major [rows, C] fp32 matrix, one row per time step (mel frame, token or
//   audio-rate sample), channels contiguous.  Utterance b's step t lives at row G + b*S + t, where
//   G is a leading guard band and S = L + gap the per-utterance stride; the gap rows between
//   utterances double as the zero padding every convolution of the path needs, so a conv tap is just
//   a row offset and an utterance batch is one flat GEMM M dimension.  A byte-per-row `rowmask` says
//   which rows are real frames; kernels *select* zero for masked rows (never multiply), so guard
//   rows may hold garbage.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <string>

#define JV_OK 0
#define JV_ERR_ARG 1
#define JV_ERR_STATE 2
#define JV_ERR_HIP 3
#define JV_ERR_SHAPE 4
#define JV_ERR_NAME 5

// Tuning aids (ablation switches, in-kernel phase stamps) compile to nothing unless the library is built with
// JV_TUNING=1 python -m jyutvoice_amd.build --force (tools/gemm_bench.py documents the switches).
#ifdef JV_TUNING
#define JV_ABLATE(p, bit) (((p).ablate & (bit)) != 0)
#ifdef JV_NO_STAMPS      // (a tuning build without the in-kernel stamps: ablation switches only)
#define JV_STAMP(p) (false)
#else
#define JV_STAMP(p) ((p).stamps != nullptr)
#endif
#else
#define JV_ABLATE(p, bit) (false)
#define JV_STAMP(p) (false)
#endif

namespace jv {

void set_error(const std::string& msg);          // thread-local last error (api.hip)
int fail(int code, const std::string& msg);      // set_error + return code

#define JV_HIP(expr)                                                                         \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return ::jv::fail(JV_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));      \
  } while (0)

#define JV_TRY(expr)              \
  do {                            \
    int _rc = (expr);             \
    if (_rc != JV_OK) return _rc; \
  } while (0)

// Test / tuning switches that may change between calls (JV_TILE, JV_OP_X6, JV_NO_X6, ...) are looked up per launch only
// when JV_DYNAMIC_ENV was set before the library's first launch (tests/conftest.py, tools/gemm_bench.py); a product
// process never walks the environment on its launch path.
inline const char* dyn_env(const char* name) {
  static const bool on = getenv("JV_DYNAMIC_ENV") != nullptr;
  return on ? getenv(name) : nullptr;
}

#ifdef JV_TUNING
inline const char* tuning_env(const char* name) { return getenv(name); }
#else
inline const char* tuning_env(const char*) { return nullptr; }
#endif

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline long cdivl(long a, long b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return cdiv(a, b) * b; }

// ---- implicit-GEMM convolution (conv_gemm.hip) -------------------------------------------------
enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_MISH = 3, ACT_ELU = 4, ACT_SILU = 5, ACT_LOGCLIP = 6 };
enum Pro : int { PRO_NONE = 0, PRO_SNAKE = 1, PRO_LRELU = 2 };

struct ConvGemmArgs {
  // A: row buffer.  Output row m, tap j reads A row (m + tap_row0 + j*tap_dil), channels [0,Cin).
  const float* A;
  long lda;        // floats between consecutive A rows
  long a_rows;     // rows of A that may be read; rows outside [0,a_rows) read as zero
  int M;           // output rows
  int Cin;         // channels per tap, multiple of 32
  int ntaps, tap_row0, tap_dil;
  // W: packed [n_rows_w][ldw], K-contiguous: column j*Cin + ci multiplies tap j, channel ci
  const float* W;
  const unsigned short* W3;   // optional: the same matrix as three bf16 planes [3][n_rows_w][ldw] (bf16x6 path)
  long w3_plane;              // elements between planes
  // optional fp16x3 path (conv_gemm_x6_kernel.h, NP = 2): W2 = two fp16 planes of W[n][:] * 2^e_n, colscale[n] = 2^-e_n, and a
  // power of two a_scale with |A| * a_scale < 65504 PROVEN by the caller (a load-time bound, registry.hip); the kernel
  // returns acc * colscale[n] / a_scale.  Ignored unless W2 is set; plain prologue only.
  const unsigned short* W2;
  long w2_plane;
  const float* colscale;
  float a_scale;
  // ... or MEASURED: amax_in points at a device float >= max |A| over every row this launch may read, maintained by the
  // kernels that wrote A (amax_out below); the kernel then derives its own power of two from *amax_in + a_extra, where
  // a_extra bounds what the prologue can add (Snake: x + sin^2(alpha x) / alpha <= |x| + 1 / min alpha).  Overflow is
  // impossible by construction either way; a_scale is ignored when amax_in is set.
  const float* amax_in;
  float a_extra;
  // optional: atomic max of |value| over everything this launch writes (bit pattern of a non-negative float), for the
  // consumer's amax_in.  The slot must be zeroed by the caller before the first producer of the buffer runs, and must not
  // be the slot this launch reads (amax_in): every buffer has its own.
  float* amax_out;
  // Both bounds are kept PER UTTERANCE so that an utterance's scales -- and with them its results, bit for bit -- do not
  // depend on what else is in the batch: amax_in / amax_out point at amax_nb floats and row r (input and output rows share
  // the row geometry) uses slot clamp((r - amax_G) / amax_S, 0, amax_nb - 1); amax_S = 0: one slot.  amax_mask (optional,
  // per output row): rows with 0 are padding whose values never reach a real frame and are left out of amax_out.
  int amax_G, amax_S, amax_nb;
  const unsigned char* amax_mask;
  // the COMPACT geometry of ragged batches (flow.hip): the slot of a row by table (row_meta's row_sample) instead of by
  // arithmetic -- utterances then start wherever the one before them ended
  const int* amax_rows;
  // split-K for short M (flow.hip: a single utterance is 20 tiles of 64x64 x 32 dependent K steps otherwise): grid.y = ksplit
  // workgroups per tile each contract a contiguous share of the 32-channel chunks and write a plain fp32 partial to
  // out + y * split_stride (the caller points `out` at a partial-sum workspace and leaves bias / epilogue to
  // splitk_reduce_rows); 0 / 1 = off
  int ksplit;
  long split_stride;
  int rowtab_off;      // set by the launcher: byte offset of the tile's per-row table in dynamic LDS (conv_gemm_epilogue.h)
  // optional with W2 (linears, ntaps = 1): A already split by its producer -- two fp16 planes [2][a_rows][lda2] of
  // A * a_scale (LayerNorm, attention and the GELU epilogue write them, same bytes as the fp32 rows); both operands then
  // reach LDS by LDS-DMA and the main loop has no VALU work and one barrier per step
  const unsigned short* A2;
  long a2_plane, lda2;
  // optional output as planes instead of fp32 rows: out2[0/1][m][ldo2] = fp16 split of value * out2_scale (lean epilogues)
  unsigned short* out2;
  long out2_plane, ldo2;
  float out2_scale;
  int ldw;
  int n_rows_w;    // rows of W that may be read (>= N, zero padded)
  int N;           // valid output columns
  const float* bias;               // [N] or null
  float* out;
  long ldo;
  // prologue (applied to A while staging it into LDS)
  int pro;
  const float* pro_alpha;          // PRO_SNAKE: per input channel
  float pro_slope;                 // PRO_LRELU
  const unsigned char* rowmask_in; // per A row, or null: 0 -> row reads as zero
  // epilogue, in this order
  int ln;                          // LayerNorm over the N columns (LN tile variant only)
  const float* ln_g;
  const float* ln_b;
  float ln_eps;
  int act;
  const unsigned char* rowmask_out;  // per output row, or null: 0 -> value := 0
  const float* rowvec;             // + rowvec[row_sample[m]*rowvec_ld + n]
  const int* row_sample;
  int rowvec_ld;
  const float* res1;               // + res1[m*ldr1 + n]
  long ldr1;
  const float* res2;
  long ldr2;
  float out_scale;                 // * out_scale
  int accumulate;                  // += previous out value
  // optional batch (grid.z = nb1*nb2): pointers advance by z1*s?1 + z2*s?2 floats
  int nb2;
  long sA1, sA2, sW1, sW2, sO1, sO2;
  // algorithmic size for the profiler (0 = use M / ntaps*Cin): real frames and real taps*channels, without guard
  // rows, channel padding or the zero taps of a polyphase transposed conv
  long alg_rows;
  int alg_k;
  unsigned long long* stamps;   // tuning aid (JV_STAMPS): per-workgroup s_memtime at start / loop / epilogue / end
  int ablate;   // tuning aid (JV_ABLATE): 1 no global loads in loop, 2 no LDS stores, 4 no barrier, 16 no epilogue
};

void conv_gemm_defaults(ConvGemmArgs& a);
int conv_gemm(const ConvGemmArgs& a, int nbatch, hipStream_t st);
int conv_gemm_init();   // raises the dynamic-LDS limit of every instantiation
int conv_gemm_x6(const ConvGemmArgs& a, hipStream_t st);   // split-plane main loops (conv_gemm_x6_kernel.h); needs a.W3

// ---- in-library kernel profiler (profile.hip): HIP events around the hot kernels, on the launch stream ----------
bool prof_on();
void prof_begin(hipStream_t st);                                   // records the start event
void prof_end(hipStream_t st, const char* name, double flops, double bytes);   // records the stop event
void prof_group(const char* tag);   // launches until the next call also count under "_group:<tag>" (null: none)

// ---- attention (attention.hip) -----------------------------------------------------------------
struct AttnArgs {
  const float* qkv;   // [rows, ld] with q at col 0, k at col k_off, v at col v_off (+ head*64)
  long ld;
  int k_off, v_off;
  float* out;         // [rows, ldo], head h writes cols [h*64, h*64+64)
  long ldo;
  int B, H;
  int G, S, L;        // row geometry: utterance b, frame t at row G + b*S + t, t < L
  const int* uoff;    // optional (attention64_planes / attention64_single only): the COMPACT geometry of ragged batches -- utterance b
                      // starts at row uoff[b] (device, [B]) and owns lens[b] rows: queries past lens[b] are the next utterance's rows
  const int* lens;    // [B] valid keys per utterance (device), or null = L
  // fp16x3 contractions (attention.hip, NP = 2) when q_scale > 0: exact powers of two with |q| log2(e)/8 * q_scale,
  // |k| * k_scale, |v| * v_scale < 65504 PROVEN by the caller; 0 = bf16x6, any fp32 operand
  float q_scale, k_scale, v_scale;
  // optional: the result as two fp16 planes of value * out2_scale (out2[0/1][row][ldo]) instead of fp32 rows
  unsigned short* out2;
  long out2_plane;
  float out2_scale;
  // attention64_planes (attention_pl.hip): K and V pre-split by the qkv GEMM's epilogue -- fp16 planes [2][rows][kv_ld] of
  // k * k_scale (columns 0..511) and v * v_scale (columns 512..1023); q stays fp32 in qkv (row stride ld)
  const unsigned short* kv2;
  long kv2_plane;
  int kv_ld;
  int chunk;          // > 0: chunk-causal (streaming) mask -- query i sees keys j < (i / chunk + 1) * chunk; 0: all keys
  int ablate;         // tuning aid (JV_ABLATE): 1 no K/V loads, 2 no split + LDS stores, 4 no barriers, 8 no PV, 16 no QK^T, 32 no softmax
};
int attention64(const AttnArgs& a, hipStream_t st);
int attention64_planes(const AttnArgs& a, hipStream_t st);
int attention64_rows(const AttnArgs& a, hipStream_t st);      // attention_r.hip: attention64_planes' contract, chunk == 0 only
int attention64_single(const AttnArgs& a, hipStream_t st);    // attention_s.hip: the same, one wave per SIMD, software-pipelined
bool attention64_single_fits(const AttnArgs& a);              // ... and whether its workgroups fill their rounds of the chip

// ---- row-wise / elementwise kernels (rowops.hip) -------------------------------------------------
// out = LayerNorm_C(x (+ add)) * g + b, optional ReLU, rows with rowmask_out == 0 written as zero
int layernorm_rows(const float* x, const float* add, float* out, const float* g, const float* b, float eps, long rows,
                   int C, const unsigned char* rowmask_out, hipStream_t st, int relu = 0);
// out = tail(sum_z partial[z] + bias), tail = [LayerNorm -> act -> mask] -> + rowvec[sample] -> + res, tracked like
// ln_epilogue_rows; optionally a second LayerNorm of the stored row -> out2 (the next GEMM's input).  256 columns.
struct SplitKReduceArgs {
  const float* partial; int ksplit; long split_stride; long rows;
  const float* bias;
  int ln; const float *ln_g, *ln_b; float ln_eps; int act;
  const unsigned char* rowmask;
  const float* rowvec; const int* row_sample; int rowvec_ld;
  const float* res; long ldr;
  float* out; long ldo;
  float* amax_out; const unsigned char* amax_mask;      // slot = row_sample[row]
  const float *ln2_g, *ln2_b; float* out2;              // optional: LayerNorm_256(out row) -> out2 [rows, 256]
};
int splitk_reduce_rows(const SplitKReduceArgs& a, hipStream_t st);
// fp16 planes [2][N][ldw] (row n = output column n, K-contiguous) -> fragment order [2][K/32][N/16][64 lanes][8 halves] (k-step major: the
// column blocks a workgroup loads in one step are contiguous): lane l of a v_mfma_f32_16x16x32_f16 B operand holds row
// 16 nb + (l & 15), k = 32 ks + 8 (l >> 4) ... + 8 (rowgemm.hip)
int pack_wfrag(const unsigned short* w2, long w2_plane, int ldw, int N, int K, unsigned short* wf, long wf_plane, hipStream_t st);
int layernorm256_planes(const float* x, unsigned short* out2, long plane, float scale, const float* g, const float* b, float eps,
                        long rows, hipStream_t st);

}  // namespace jv
