// Context, weight registry and per-stage weight views of libjyutvoice_hip.so.
#pragma once
#include <string>
#include <unordered_map>
#include <vector>

#include "jv_common.h"

namespace jv {

// ---- architecture constants (configs/base.yaml:1-110 of the reference; mirrored by spec.py) ------
constexpr int N_FEATS = 80;
constexpr int SPK_DIM = 192;
constexpr int ENC_CH = 192, ENC_HID = 576, ENC_FILTER = 768, ENC_HEADS = 2, ENC_HEAD_DIM = 288, ENC_ROPE = 144,
              ENC_LAYERS = 6;
constexpr int DP_FILTER = 256;
constexpr int EST_IN = 320, EST_CH = 256, EST_TIME = 1024, EST_HEADS = 8, EST_INNER = 512, EST_FF = 1024, EST_NBLK = 4,
              EST_NMID = 12, EST_NRES = 14;
constexpr int NOISE_FRAMES = 15000;
constexpr int HIFT_CH = 512, HIFT_F0_CH = 512, HIFT_NFFT = 16, HIFT_HOP = 4, HIFT_HARM = 9;

constexpr int PR_DIM = 512, PR_HEADS = 8, PR_FFN = 2048, PR_BLOCKS = 6, PR_UP_BLOCKS = 4, PR_VOCAB = 6561;

enum Model : int { MODEL_TTS = 0, MODEL_HIFT = 1, MODEL_PROMPT = 2 };

struct RawTensor {
  std::string name;
  std::vector<int64_t> shape;
  long numel = 0;
  float* dev = nullptr;
  bool loaded = false;
  int model = 0;
};

// simple bump allocator over hipMalloc'd chunks (device memory only; never freed before destroy)
struct Arena {
  std::vector<void*> chunks;
  size_t chunk_floats = (size_t)64 << 20;   // 256 MiB
  size_t used = 0, cap = 0;
  float* cur = nullptr;
  int alloc(size_t floats, float** out);     // 256-byte aligned
  void release();
};

// a packed weight matrix for conv_gemm: rows = output channels, K-contiguous, column j*Cin + ci
struct GemmW {
  const float* w = nullptr;
  const unsigned short* w3 = nullptr;   // three bf16 planes of w (split at load time), plane stride n_rows*ldw
  int ldw = 0, n_rows = 0, N = 0, Cin = 0, ntaps = 1;
  const float* bias = nullptr;
  // fp16x3 path, attached only where a load-time bound on the layer's input exists (registry.hip `half3`):
  const unsigned short* w2 = nullptr;   // two fp16 planes of w[n][:] * 2^e_n
  const unsigned short* wf = nullptr;   // the same two planes in MFMA-fragment order (rowgemm_kernel.h, `Wf`), plane stride N * Cin
  const float* colscale = nullptr;      // 2^-e_n
  float a_scale = 0.f;                  // power of two with bound(|input|) * a_scale < 65504; 0 = no bound, use bf16x6
};

struct LnW { const float* g = nullptr; const float* b = nullptr; };

struct ResnetW {
  GemmW block1, block2, res;
  LnW ln1, ln2;
  // block1's three taps AND res_conv as a fourth fragment step per 32-channel chunk (k-step major fragments: steps 0 .. 3 NCH - 1
  // are block1's, 3 NCH .. 4 NCH - 1 res_conv's), plane stride 256 * 4 * Cin halves: rowconv_wd_kernel<RT, true> (registry.hip)
  const unsigned short* wf4 = nullptr;
  // sqrt(255) max |ln1.g| + max |ln1.b|: the range of block1's LayerNorm, hence (with Mish(v) <= max(v, 0.31) and the time
  // embedding's own maximum) of h2 -- the whole-resnet launch scales its LDS-resident h2 with it (rowres_kernel.h); 0 = n/a
  float h2_bound = 0.f;
};
struct BtbW {
  LnW n1, n3;
  GemmW qkv, out, ff1, ff2;
  float q_scale = 0.f, k_scale = 0.f, v_scale = 0.f;   // fp16x3 attention (AttnArgs), from the load-time bounds; 0 = bf16x6
};
struct EstimatorW {
  GemmW time1, time2, temb_all;
  ResnetW res[EST_NRES];                 // 0 = down, 1..12 = mid, 13 = up
  BtbW blk[EST_NRES][EST_NBLK];
  GemmW down_conv, up_conv, final_conv, final_proj;
  LnW final_ln;
};

struct EncLayerW { GemmW qkv, o, ffn1, ffn2; LnW n1, n2; };
struct EncoderW {
  const float *emb = nullptr, *lang_emb = nullptr, *tone_emb = nullptr, *wpos_emb = nullptr, *spos_emb = nullptr;
  GemmW pre_conv[3], pre_proj;
  LnW pre_ln[3];
  EncLayerW layer[ENC_LAYERS];
  GemmW proj;
  GemmW dp_cond, dp_conv1, dp_conv2, dp_proj;
  LnW dp_ln1, dp_ln2;
  GemmW spk_affine;
};

struct ResBlockW {
  GemmW c1[3], c2[3];
  const float* a1[3];
  const float* a2[3];
  float e1[3] = {}, e2[3] = {};   // max_c 1 / (alpha_c + 1e-9): what Snake can add to |x| (fp16x3 with a measured bound); 0 = n/a
  // the pair in one launch (hiftpair_kernel.h): both convolutions' fragments as one stream (c1's steps, then c2's; plane stride
  // C * 2 k C halves), and the load-time half of the intermediate's bound: c1's largest row L1 norm and largest |bias|
  const unsigned short* wfp[3] = {};
  float l1max[3] = {}, b1max[3] = {};
  int k = 0;
};
struct HiftW {
  GemmW f0_conv[5];
  const float *f0_cls_w = nullptr, *f0_cls_b = nullptr;
  const float *src_lin_w = nullptr, *src_lin_b = nullptr;
  GemmW conv_pre, ups[3], src_down[3], conv_post;
  ResBlockW src_rb[3], rb[9];
};

// prompt branch: FlowEncoder of infer.py:35-83 (Embedding -> UpsampleConformerEncoder -> Linear)
struct ConfBlockW { LnW n_mha, n_ff; GemmW qkv, pos, out, w1, w2; const float *u = nullptr, *v = nullptr; };
struct PromptW {
  const float* emb = nullptr;          // [6561][512]
  const float* div = nullptr;          // [256] positional frequencies (host-computed, embedding.py:239-242)
  GemmW emb_lin, up_emb_lin;           // LinearNoSubsampling.out.0
  LnW emb_ln, up_emb_ln;               // .out.1 with gain and offset pre-multiplied by sqrt(512) (the xscale of the pos-enc)
  GemmW look1, look2, up_conv, proj;
  LnW after;
  ConfBlockW blk[PR_BLOCKS], up[PR_UP_BLOCKS];
};

struct Buf {             // a device allocation owned by the context
  float* p = nullptr;
  size_t floats = 0;
};

struct Context {
  int device = 0;
  int max_batch = 0, max_frames = 0, max_tokens = 0;
  std::vector<RawTensor> raw;
  std::unordered_map<std::string, int> index;
  Arena raw_arena, packed;
  bool ready[3] = {false, false, false};
  bool broken = false;           // jv_reserve failed and could not restore the old workspace: every entry point returns JV_ERR_STATE
  EstimatorW est;
  EncoderW enc;
  HiftW hift;
  PromptW prompt;
  float* noise = nullptr;        // [80][15000] fixed CFM noise (device), supplied by the host
  bool noise_loaded = false;
  bool dma_a = false;            // JV_DMA_A=1: fp16x3 linears take their A operand pre-split from the producer (measured slower
                                 // in the pipeline than the in-kernel split; kept as a tested alternative)
  bool no_ffn_fuse = false;      // JV_NO_FFN_FUSE=1: ff.net.0 and ff.net.2 as two launches (the path rowffn_kernel is checked against)
  bool no_temb_pre = false;      // JV_NO_TEMB_PRE=1: the timestep embedding inside every Euler step instead of once per solve (flow.hip cfm_solve)
  bool no_ln_fold = false;       // JV_NO_LN_FOLD=1: a stage's first norm1 as its own launch (layernorm256_planes) instead of in the resnet's last convolution
  bool no_res_qkv = false;       // JV_NO_RES_QKV=1: a stage's first q | k | v as its own launch (rowgemm_wa) instead of inside the resnet's (rowres_kernel.h)
  bool no_res_pair = false;      // JV_NO_RES_PAIR=1: a resnet as two row-owning launches (block1 + res_conv, block2) instead of one (rowres_kernel.h)
  bool no_res_fold = false;      // JV_NO_RES_FOLD=1: a resnet's 1 x 1 res_conv as a tile-kernel launch of its own instead of inside block1's row-owning launch
  bool no_compact = false;       // JV_NO_COMPACT=1: ragged batches keep the uniform row geometry (every utterance padded to the longest; flow.hip cfm_solve)
  bool no_qkv_split = false;     // JV_NO_QKV_SPLIT=1: q|k|v stays inside the fused block launch at every batch size (flow.hip `qkv_split`)
  bool no_block_fuse = false;    // JV_NO_BLOCK_FUSE=1: to_out / feed-forward / next q|k|v as three launches (the path rowblock_kernel is checked against)
  bool rg_ff1 = true;            // ff.net.0 on the row-owning GEMM too; JV_TILE_FF1=1: on the tile kernel (the round-2 first build, for A/B runs)
  bool attn_single = true;       // attention_s.hip for whole-utterance attention (one wave per SIMD, software-pipelined); JV_NO_ATTN_SINGLE=1: attention_pl.hip
  bool attn_rows = false;        // JV_ATTN_ROWS=1: the estimator's attention on attention_r.hip (one workgroup per head, 80 queries per wave; measured
                                 // 66.6 us against attention_pl.hip's 60.0 at 32 x 300 frames: kept as a tested alternative)
  bool no_hift_pair = false;     // JV_NO_HIFT_PAIR=1: a ResBlock's two convolutions as two hiftconv launches (the path hiftpair_kernel is checked against)
  bool no_hiftconv = false;      // JV_NO_HIFTCONV=1: the vocoder's ResBlock convolutions on the tile kernels (A/B aid; the path hiftconv_kernel is checked against)
  bool no_attn_planes = false;   // JV_NO_ATTN_PLANES=1: attention splits K / V itself (attention.hip) instead of taking planes
  bool no_splitk = false;        // JV_NO_SPLITK=1: no split-K at short M (A/B aid)
  bool no_rowgemm = false;       // JV_NO_ROWGEMM=1: keep the transformer linears on the tile kernels at every batch size (A/B aid)
  bool exact_range = false;      // true: bf16x6 everywhere (jv_flow_set_contraction); false: fp16x3 where the range is proven
  bool step_graphs = false;      // replay the Euler step as a captured hipGraph (jv_flow_set_graph; never under the profiler)
  int attn_chunk = 0;            // > 0: streaming (chunk-causal) estimator attention, in frames (jv_flow_set_streaming)
  // workspace
  std::vector<void*> ws_allocs;
  struct FlowWs* flow = nullptr;
  struct HiftWs* hws = nullptr;
  struct EncWs* ews = nullptr;
  struct PromptWs* pws = nullptr;
  struct AudioWs* aws = nullptr;
  std::string last_error;
};

// registry.hip
void build_registry(Context& c);
int finalize_model(Context& c, int model, hipStream_t st);

// workspace helpers (api.hip)
int ws_alloc(Context& c, size_t bytes, void** out);

// flow.hip
int flow_ws_create(Context& c);
void flow_ws_forget_attention(Context& c, hipStream_t st);   // zero the attention buffer (new weights / contraction mode)
bool flow_has_graphs(const Context& c);
void flow_graphs_drop(Context& c);   // forget captured Euler-step graphs (weights or workspace pointers changed)
int flow_estimator(Context& c, const float* x, const int* lens_dev, const float* mu, const float* t_dev, const float* spks,
                   const float* cond, int B2, int T, float* out, hipStream_t st, const float* mask_f32 = nullptr);
int cfm_solve(Context& c, const float* mu, const int* lens_dev, const float* spks, const float* cond, int B, int T,
              int n_timesteps, float temperature, const float* t_span_host, float* mel, hipStream_t st);

// prompt.hip
int prompt_encoder_fwd(Context& c, const long* tok, const long* len, int B, int Tk, float* h_out, hipStream_t st);
void prompt_ws_destroy(Context& c);

// audio.hip
void audio_ws_destroy(Context& c);

}  // namespace jv

// the opaque handle of include/jyutvoice_hip.h
struct jv_context {
  jv::Context c;
};
