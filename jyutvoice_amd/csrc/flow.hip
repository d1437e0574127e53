// Flow-matching decoder: one estimator evaluation (jyutvoice/flow/decoder.py:917-1018) and the
// Euler / classifier-free-guidance solver around it (jyutvoice/flow/flow_matching.py:215-265, 356-401).
//
// Everything runs on row buffers [G + b*S + t][C] (jv_common.h): the estimator's `b c t <-> b t c`
// transposes (decoder.py:950,966) disappear, causal convolutions are 3-tap row-offset GEMMs, and the
// 2B CFG rows are just more rows.  Per estimator call: 1 input assembly, 3 tiny time-MLP GEMMs,
// 14 x (3 conv GEMMs) + 56 x (2 LayerNorm + 4 GEMM + 1 attention) + 4 tail launches, all enqueued on the
// caller's stream with no host synchronisation (graph-capturable).
#include <math.h>
#include <stdio.h>

#include <algorithm>
#include <vector>

#include "jv_model.h"
#include "jv_ops.h"
#include "rowblock_kernel.h"
#include "rowconv_kernel.h"
#include "rowres_kernel.h"

namespace jv {

int rowgemm(const RowGemmArgs& a, int epi, hipStream_t st);   // rowgemm.hip
int rowconv(const RowConvArgs& a, hipStream_t st);
int rowres(const RowResArgs& a, hipStream_t st);      // rowgemm.hip: a whole resnet in one launch
bool rowres_fits(int M);
bool rowconv_w_direct(const RowConvArgs& a);      // rowgemm.hip
int rowffn(const RowFfnArgs& a, hipStream_t st);
int rowblock(const RowBlockArgs& a, bool qkv, hipStream_t st);   // rowblock.hip
int rowgemm_tile(int M);

constexpr int FLOW_G = 4;      // leading guard rows (>= causal left context 2)
constexpr int FLOW_GAP = 4;    // rows between utterances
constexpr int PARTIAL_ROWS = 2048, PARTIAL_SPLITS = 8;      // split-K is for short M only (estimator_body)

struct FlowWs {
  long rows_alloc = 0;      // rows every [*,C] buffer below can hold
  float *x = nullptr, *mu = nullptr, *cond = nullptr, *spks = nullptr;   // [rows,80] x3, [maxB,80]
  float *xin = nullptr;                                               // [rows,320]
  float *h = nullptr, *h2 = nullptr, *res = nullptr, *cat = nullptr;  // [rows,256] x3, [rows,512]
  float *ln = nullptr, *qkv = nullptr, *att = nullptr, *ff = nullptr; // 256, 1536, 512, 1024
  // max |value| written to the trunk buffers during the current solve, one slot per BUFFER (h, h2, cat: a launch never
  // reads the slot it writes) and per UTTERANCE (CFG twins count as utterances of their own: [3][2 * max_batch] floats):
  // every kernel that writes one of them tracks it (ConvGemmArgs::amax_out, ln_epilogue_rows), and the convolutions that
  // read them -- whose input, the residual stream, has no load-time bound -- derive their fp16x3 scale from it (amax_in).
  // An utterance's scales therefore depend on that utterance alone: its result is the same bit for bit whatever else is
  // in the batch and however a batch is sharded over GPUs.  Zeroed once per solve.
  float* amax = nullptr;
  int amax_stride = 0;      // floats per buffer = 2 * max_batch
  float* partial = nullptr;      // [8][PARTIAL_ROWS][256] split-K partial sums (short M only)
  unsigned long long* rb_stamps = nullptr;      // tuning builds, JV_RB_STAMPS: rowblock_kernel's phase stamps of the last launch
  unsigned long long* rc_stamps = nullptr;      // ... and rowconv_wd_kernel's
  float *d = nullptr;                                                 // [rows,80]
  float *tsin = nullptr, *t1 = nullptr, *tmish = nullptr, *temb = nullptr;
  float *t_dev = nullptr, *t_table = nullptr, *dt_table = nullptr;
  // cfm_solve: the timestep embedding of EVERY step of a solve, computed in three launches before the loop (the steps'
  // t are known up front and the same for all rows): [TS_MAX] rows of sinusoid / hidden / Mish / the 14 projections
  float *ts_sin = nullptr, *ts_1 = nullptr, *ts_mish = nullptr, *ts_emb = nullptr;
  unsigned char* rowmask = nullptr;
  int* row_sample = nullptr;
  int* lens2 = nullptr;     // [2*maxB]
  // COMPACT geometry of ragged batches (cfm_solve): first row of every utterance (+ the first row past the batch), [2*maxB + 1];
  // h_lens / h_uoff: pinned host staging (the lengths come down with the solve's one synchronisation, the offsets go up async)
  int* uoff = nullptr;
  int *h_lens = nullptr, *h_uoff = nullptr;
  int max_steps = 1024;
  // One Euler step (step scalars -> input assembly -> estimator -> CFG update) captured as a hipGraph per (B, T,
  // attention mode): the step reads its (t, dt) through a device-side counter, so one executable graph replays for
  // every step of every solve of that geometry.  Replayed on a private stream (the caller's may be the legacy
  // default stream, which cannot be captured), fenced against the caller's stream with events.
  int* step_ctr = nullptr;
  float *t_cur = nullptr, *dt_cur = nullptr;
  struct StepGraph { int B, T, chunk, pre; hipGraph_t graph; hipGraphExec_t exec; };      // pre: captured with the embeddings precomputed
  std::vector<StepGraph> graphs;
  hipStream_t gstream = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
};

constexpr int TS_MAX = 64;      // steps whose timestep embeddings cfm_solve computes ahead of the loop (more: per step, as the seam does)

static long flow_rows(int B2, int T) { return (long)FLOW_G + (long)B2 * (T + FLOW_GAP); }

int flow_ws_create(Context& c) {
  FlowWs* w = new FlowWs();
  c.flow = w;
  const int B2 = 2 * c.max_batch;
  // row indices are ints inside the kernels
  if (c.max_batch > (1 << 20) || flow_rows(B2, c.max_frames) > (1L << 30)) return fail(JV_ERR_SHAPE, "flow workspace: batch x frames beyond 2^30 rows");
  w->rows_alloc = round_up((int)flow_rows(B2, c.max_frames), 128) + 256;
  const size_t R = (size_t)w->rows_alloc;
  auto F = [&](float** p, size_t floats) { return ws_alloc(c, floats * sizeof(float), reinterpret_cast<void**>(p)); };
  JV_TRY(F(&w->x, R * 80));
  JV_TRY(F(&w->mu, R * 80));
  JV_TRY(F(&w->cond, R * 80));
  JV_TRY(F(&w->spks, (size_t)B2 * 80));
  JV_TRY(F(&w->xin, R * 320));
  JV_TRY(F(&w->h, R * 256));
  JV_TRY(F(&w->h2, R * 256));
  JV_TRY(F(&w->res, R * 256));
  JV_TRY(F(&w->cat, R * 512));
  JV_TRY(F(&w->ln, R * 256));
  JV_TRY(F(&w->qkv, R * 1536));
  JV_TRY(F(&w->att, R * 512));
  JV_TRY(F(&w->ff, R * 1024));
  JV_TRY(F(&w->d, R * 80));
  JV_TRY(F(&w->partial, (size_t)PARTIAL_SPLITS * PARTIAL_ROWS * 256));
  JV_TRY(F(&w->tsin, (size_t)B2 * 320));
  JV_TRY(F(&w->t1, (size_t)B2 * 1024));
  JV_TRY(F(&w->tmish, (size_t)B2 * 1024));
  JV_TRY(F(&w->temb, (size_t)B2 * EST_NRES * 256));
  w->amax_stride = B2;
  JV_TRY(F(&w->amax, (size_t)3 * B2));
  JV_TRY(F(&w->t_dev, (size_t)B2));
  JV_TRY(F(&w->ts_sin, (size_t)TS_MAX * 320));
  JV_TRY(F(&w->ts_1, (size_t)TS_MAX * 1024));
  JV_TRY(F(&w->ts_mish, (size_t)TS_MAX * 1024));
  JV_TRY(F(&w->ts_emb, (size_t)TS_MAX * EST_NRES * 256));
  JV_TRY(F(&w->t_table, (size_t)w->max_steps));
  JV_TRY(F(&w->dt_table, (size_t)w->max_steps));
  JV_TRY(ws_alloc(c, R, reinterpret_cast<void**>(&w->rowmask)));
  JV_TRY(ws_alloc(c, R * sizeof(int), reinterpret_cast<void**>(&w->row_sample)));
  JV_TRY(ws_alloc(c, (size_t)B2 * sizeof(int), reinterpret_cast<void**>(&w->lens2)));
  JV_TRY(ws_alloc(c, (size_t)(B2 + 1) * sizeof(int), reinterpret_cast<void**>(&w->uoff)));
  JV_HIP(hipHostMalloc(reinterpret_cast<void**>(&w->h_lens), (size_t)B2 * sizeof(int), hipHostMallocDefault));
  JV_HIP(hipHostMalloc(reinterpret_cast<void**>(&w->h_uoff), (size_t)(B2 + 1) * sizeof(int), hipHostMallocDefault));
  JV_TRY(ws_alloc(c, sizeof(int), reinterpret_cast<void**>(&w->step_ctr)));
  JV_TRY(F(&w->t_cur, 1));
  JV_TRY(F(&w->dt_cur, 1));
  JV_HIP(hipStreamCreateWithFlags(&w->gstream, hipStreamNonBlocking));
  JV_HIP(hipEventCreateWithFlags(&w->ev_in, hipEventDisableTiming));
  JV_HIP(hipEventCreateWithFlags(&w->ev_out, hipEventDisableTiming));
  return JV_OK;
}

bool flow_has_graphs(const Context& c) { return c.flow && !c.flow->graphs.empty(); }

void flow_graphs_drop(Context& c) {
  if (!c.flow) return;
  for (auto& e : c.flow->graphs) {
    (void)hipGraphExecDestroy(e.exec);
    (void)hipGraphDestroy(e.graph);
  }
  c.flow->graphs.clear();
}

// new weights: rows of the attention buffer written under the old ones may exceed the new V bound (registry.hip)
void flow_ws_forget_attention(Context& c, hipStream_t st) {
  if (c.flow && c.flow->att) (void)hipMemsetAsync(c.flow->att, 0, (size_t)c.flow->rows_alloc * 512 * sizeof(float), st);
}

namespace {

// (t, dt) of the step the device-side counter points at, then advance it: the only step-dependent state of a solve
// (256 threads; emb_steps != null: the step's precomputed timestep embedding, [EST_NRES * 256] floats, moves to the fixed
// place the estimator's launches read it from -- their arguments are frozen inside a captured graph)
__global__ void step_advance_kernel(const float* __restrict__ t_table, const float* __restrict__ dt_table, int* ctr,
                                    float* t_cur, float* dt_cur, const float* __restrict__ emb_steps, float* __restrict__ emb_cur) {
  const int i = *ctr;
  if (emb_steps)
    for (int k = threadIdx.x; k < EST_NRES * 256; k += 256) emb_cur[k] = emb_steps[(long)i * EST_NRES * 256 + k];
  __syncthreads();      // every thread has read the counter
  if (threadIdx.x == 0) {
    *t_cur = t_table[i];
    *dt_cur = dt_table[i];
    *ctr = i + 1;
  }
}

struct Geo {
  int B2, T, S;
  long M;        // rows computed by every GEMM: [0, M)
  long a_rows;   // rows that may be read
  const float* t_ptr = nullptr;   // timestep per utterance: t_ptr[b * t_stride]
  int t_stride = 1;
  bool temb_pre = false;          // w.temb row 0 already holds this step's embedding, the same for every utterance (cfm_solve)
  // COMPACT geometry (ragged batches, cfm_solve): utterance b starts at row uoff[b] and owns lens2[b] rows + the gap; M is then
  // G + sum (len + gap), not G + B2 (T + gap), and every launch of the call is that much shorter.  null: uniform, G + b S + t
  const int* uoff = nullptr;
  long alg_rows = 0;              // profiler: real frames of the call (0: B2 * T)
};

ConvGemmArgs base_args(const Geo& g, const float* A, int lda, const GemmW& w, float* out, int ldo) {
  ConvGemmArgs a;
  conv_gemm_defaults(a);
  a.A = A; a.lda = lda; a.a_rows = g.a_rows; a.M = (int)g.M;
  a.Cin = w.Cin; a.ntaps = w.ntaps; a.tap_row0 = 0; a.tap_dil = 1;
  a.W = w.w; a.ldw = w.ldw; a.n_rows_w = w.n_rows; a.N = w.N; a.bias = w.bias;
  a.W3 = w.w3; a.w3_plane = (long)w.n_rows * w.ldw;
  a.out = out; a.ldo = ldo;
  a.alg_rows = g.alg_rows ? g.alg_rows : (long)g.B2 * g.T;
  return a;
}

// timestep embedding of n timesteps t[i * t_stride]: sinusoid -> Linear + SiLU -> Linear (+ Mish, the only consumer) -> the 14
// resnets' projections, emb [n, 14 * 256] (decoder.py:917-935, 98-108).  Rows are independent: a row's bits do not depend on n.
int time_embedding(Context& c, const float* t, int t_stride, int n, float* sin_buf, float* h1, float* hm, float* emb, hipStream_t st) {
  const EstimatorW& e = c.est;
  JV_TRY(time_sinusoid(t, t_stride, sin_buf, n, st));
  Geo tg{n, 1, 1, n, n, nullptr, 1};
  ConvGemmArgs a = base_args(tg, sin_buf, 320, e.time1, h1, 1024);
  a.act = ACT_SILU;
  JV_TRY(conv_gemm(a, 1, st));
  a = base_args(tg, h1, 1024, e.time2, hm, 1024);
  a.act = ACT_MISH;
  JV_TRY(conv_gemm(a, 1, st));
  a = base_args(tg, hm, 1024, e.temb_all, emb, EST_NRES * 256);
  return conv_gemm(a, 1, st);
}

// the estimator body on prepared inputs: ws.xin [rows,320], ws.rowmask/row_sample/lens2, ws.t_dev [B2] -> ws.d [rows,80]
int estimator_body(Context& c, const Geo& g, hipStream_t st) {
  FlowWs& w = *c.flow;
  const EstimatorW& e = c.est;
  const int B2 = g.B2;

  // ---- timestep embedding: sinusoid -> Linear+SiLU -> Linear (+Mish, the only consumer) -> 14 projections
  if (!g.temb_pre) JV_TRY(time_embedding(c, g.t_ptr, g.t_stride, B2, w.tsin, w.t1, w.tmish, w.temb, st));

  auto causal3 = [&](ConvGemmArgs& a) {   // CausalConv1d k=3: rows t-2, t-1, t of the masked input
    a.tap_row0 = -2;
    a.rowmask_in = w.rowmask;
  };
  float* skip = w.cat + 256;   // columns [256,512) of the concat buffer
  // trunk convolutions: fp16x3 from the measured bound of the trunk buffers (not for A = xin, which assemble_xin writes)
  // (zeroed by the caller once per solve, not per call: in the first launch after a reset every wave sends its atomic --
  // the per-CU L1 keeps serving the value the slot had at kernel start -- which cost 2.4 ms per step when done ten times)
  auto slots_of = [&](const float* buf) -> float* {
    if (buf == w.h) return w.amax;
    if (buf == w.h2) return w.amax + w.amax_stride;
    if (buf == w.cat || buf == skip) return w.amax + 2 * w.amax_stride;
    return nullptr;
  };
  auto amax_geo = [&](ConvGemmArgs& a) {
    a.amax_G = FLOW_G; a.amax_S = g.S; a.amax_nb = B2; a.amax_mask = w.rowmask;
    a.amax_rows = g.uoff ? w.row_sample : nullptr;      // compact geometry: a row's slot by table
  };
  // every launch that writes a trunk buffer tracks max |value| into that buffer's slots (nothing consumes them in
  // exact-range mode)
  auto track = [&](ConvGemmArgs& a) {
    if (c.exact_range) return;
    a.amax_out = slots_of(a.out);
    amax_geo(a);
  };
  auto h3m = [&](ConvGemmArgs& a, const GemmW& m) {
    if (c.exact_range || !m.w2 || a.A == w.xin || !slots_of(a.A)) return;
    a.W2 = m.w2; a.w2_plane = (long)m.n_rows * m.ldw; a.colscale = m.colscale; a.amax_in = slots_of(a.A); a.a_extra = 0.f;
    amax_geo(a);
  };
  // ---- short M (a single utterance: 268 rows = 20 tiles of 64x64): the K = 512 ... 1536 contractions to 256 channels are
  // split over ksplit workgroups per tile (ConvGemmArgs::ksplit) and one row-wise kernel sums the partials and runs the
  // whole tail -- bias, LayerNorm / Mish / mask, time embedding, residual, tracking, and the LayerNorm that feeds the next
  // GEMM (splitk_reduce_rows).  Traced at B = 1: ff.net.2 + its LayerNorm 20.8 + 4.8 us -> 2 x ~5.
  // The split is a function of K alone (PARTIAL_SPLITS shares of the chunks whenever M <= PARTIAL_ROWS), never of M: the
  // grouping of a row's partial sums must not depend on how many rows the batch has, or a shard would no longer reproduce
  // the whole batch bit for bit (tests/test_gpu_dist.py).
  // ... and never together with the row-owning kernels (the debugging override JV_ROWGEMM_RT can force those at short M):
  // the split-K tails hand the next LayerNorm on as fp32 rows, the row-owning blocks read fp16 planes from the same buffer
  const int ksplit = (!c.no_splitk && g.M <= PARTIAL_ROWS && rowgemm_tile((int)g.M) == 0) ? PARTIAL_SPLITS : 1;
  // split-K lives in the split-plane kernels (conv_gemm_x6); a launch that would take the fp32-MFMA route (no weight
  // planes, unaligned ldw, JV_NO_X6) runs unsplit instead
  auto splittable = [&](const ConvGemmArgs& a) { return ksplit > 1 && (a.W3 || a.W2) && (a.ldw & 7) == 0 && !dyn_env("JV_NO_X6"); };
  // `a`: the full-semantics launch (N = 256); ln2 / out2: optional LayerNorm of the stored row for the next GEMM
  auto splitk = [&](const ConvGemmArgs& a, const LnW* ln2, float* out2) -> int {
    ConvGemmArgs p = a;
    p.ln = 0; p.act = ACT_NONE; p.bias = nullptr; p.rowmask_out = nullptr; p.rowvec = nullptr; p.row_sample = nullptr;
    p.res1 = nullptr; p.amax_out = nullptr;
    p.out = w.partial; p.ldo = 256;
    p.ksplit = std::min(ksplit, a.Cin >> 5); p.split_stride = (long)a.M * 256;
    JV_TRY(conv_gemm(p, 1, st));
    SplitKReduceArgs r{};
    r.partial = w.partial; r.ksplit = p.ksplit; r.split_stride = p.split_stride; r.rows = a.M;
    r.bias = a.bias; r.ln = a.ln; r.ln_g = a.ln_g; r.ln_b = a.ln_b; r.ln_eps = a.ln_eps; r.act = a.act;
    r.rowmask = a.rowmask_out; r.rowvec = a.rowvec; r.row_sample = w.row_sample; r.rowvec_ld = a.rowvec_ld;
    r.res = a.res1; r.ldr = a.ldr1; r.out = a.out; r.ldo = a.ldo;
    r.amax_out = a.amax_out; r.amax_mask = a.amax_mask;
    if (ln2) { r.ln2_g = ln2->g; r.ln2_b = ln2->b; r.out2 = out2; }
    return splitk_reduce_rows(r, st);
  };
  // A causal k = 3 convolution of a trunk buffer: on the row-owning kernel (rowconv_kernel.h: LayerNorm / Mish / mask / time
  // embedding / residual in its epilogue, no ln_epilogue_rows pass) when the batch fills the chip, else on the tile kernels
  const bool use_rc = !c.exact_range && !c.no_rowgemm && rowgemm_tile((int)g.M) > 0;
  // one decision for every block of the call (a block's split-K tail writes the NEXT block's LayerNorm): all of them take
  // the split-plane route, or none is split
  bool sk_blocks = ksplit > 1 && !c.dma_a && !dyn_env("JV_NO_X6");
  for (int i = 0; i < EST_NRES && sk_blocks; ++i)
    for (int j = 0; j < EST_NBLK; ++j) {
      const GemmW &o = e.blk[i][j].out, &f = e.blk[i][j].ff2;
      sk_blocks = sk_blocks && (o.w3 || o.w2) && (f.w3 || f.w2) && !(o.ldw & 7) && !(f.ldw & 7);
    }
  // `follow` / `followed`: the transformer block whose norm1 reads this convolution's output (a resnet's second
  // convolution, in place in the trunk): on the W-direct row-owning kernel its LayerNorm planes are written by the
  // convolution's own epilogue (RowConvArgs::ln2_out) and *followed is set; every other route leaves it to the caller
  // `rn` / `res_done`: the resnet whose block1 this is -- on the W-direct row-owning kernel its 1 x 1 res_conv (which reads the
  // same rows) rides along as a fourth fragment step per chunk and lands in w.res (RowConvArgs::res_out); *res_done says so
  auto conv3 = [&](ConvGemmArgs& a, const GemmW& m, const BtbW* follow = nullptr, bool* followed = nullptr, const ResnetW* rn = nullptr,
                   bool* res_done = nullptr) -> int {
    if (followed) *followed = false;
    if (res_done) *res_done = false;
    if (splittable(a) && a.N == 256 && !a.res2) {
      // split-K tiles: the reduce kernel's tail writes the following norm1 too (fp32 rows into w.ln, what the split-K
      // blocks read), as it does between the blocks of a stage
      const bool fl = follow && followed && sk_blocks && !c.no_ln_fold && a.out == w.h && a.ldo == 256;
      if (fl) *followed = true;
      return splitk(a, fl ? &follow->n1 : nullptr, fl ? w.ln : nullptr);
    }
    if (!use_rc || !a.amax_in || !m.w2 || a.ntaps != 3 || a.tap_row0 != -2 || a.N != 256 || a.ldo != 256 && a.ldo != 512)
      return conv_gemm(a, 1, st);
    RowConvArgs r{};
    r.A = a.A; r.lda = a.lda; r.a_rows = a.a_rows; r.M = a.M; r.Cin = a.Cin; r.rowmask_in = a.rowmask_in;
    r.W2 = m.w2; r.w2_plane = (long)m.n_rows * m.ldw; r.ldw = m.ldw; r.colscale = m.colscale;
    r.Wf = m.wf; r.wf_plane = (long)m.N * m.ntaps * m.Cin;
    r.amax_in = a.amax_in; r.row_slot = w.row_sample; r.bias = a.bias;
    r.slot_G = FLOW_G; r.slot_S = g.uoff ? -1 : g.S; r.slot_nb = B2;      // = row_sample, by arithmetic (row_meta lays utterance b at G + b S; compact: by table)
    r.out = a.out; r.ldo = a.ldo;
    r.ln = a.ln; r.ln_g = a.ln_g; r.ln_b = a.ln_b; r.ln_eps = a.ln_eps; r.act = a.act; r.rowmask_out = a.rowmask_out;
    r.rowvec = a.rowvec; r.rowvec_ld = a.rowvec_ld; r.res = a.res1; r.ldr = a.ldr1;
    r.amax_out = a.amax_out; r.row_mask = w.rowmask;
    r.alg_rows = a.alg_rows;
    if (rn && res_done && !c.no_res_fold && rn->wf4 && rn->res.colscale && rn->res.Cin == a.Cin && rowconv_w_direct(r)) {
      r.Wf = rn->wf4; r.wf_plane = 256L * 4 * a.Cin;
      r.res_out = w.res; r.res_cs = rn->res.colscale; r.res_bias = rn->res.bias;
      *res_done = true;
    }
    if (follow && followed && !c.no_ln_fold && a.out == w.h && a.ldo == 256 && follow->qkv.w2 && follow->qkv.a_scale > 0.f &&
        rowconv_w_direct(r)) {
      r.ln2_out = reinterpret_cast<unsigned short*>(w.ln); r.ln2_plane = (long)w.rows_alloc * 256;
      r.ln2_g = follow->n1.g; r.ln2_b = follow->n1.b; r.ln2_scale = follow->qkv.a_scale;
      *followed = true;
    }
    if (tuning_env("JV_RB_STAMPS")) {
      if (!w.rc_stamps) JV_TRY(ws_alloc(c, 1024 * 8 * sizeof(unsigned long long), reinterpret_cast<void**>(&w.rc_stamps)));
      r.stamps = w.rc_stamps;
      JV_HIP(hipMemsetAsync(r.stamps, 0, 1024 * 8 * sizeof(unsigned long long), st));
    }
    return rowconv(r, st);
  };
  // CausalResnetBlock1D (decoder.py:110-115, 784-795)
  // (profiler: the estimator's Conv1d stack -- resnets, down / up / final convolutions, final projection -- is summed as one
  // group; BASELINE.json's north star quotes an HBM fraction for it)
  struct ConvStackScope {
    bool on;
    ConvStackScope() : on(prof_on()) { if (on) prof_group("flow_conv_stack"); }
    ~ConvStackScope() { if (on) prof_group(nullptr); }
  };
  auto res_pair_ok = [&](int i, const float* in, int ldin) {
    const ResnetW& r = e.res[i];
    return use_rc && !c.no_res_pair && !c.no_res_fold && g.temb_pre && slots_of(in) && in != w.xin && r.wf4 && r.block2.wf &&
           r.h2_bound > 0.f && r.block1.colscale && r.block2.colscale && r.res.colscale && r.block1.Cin == ldin && !(ldin & 63) &&
           rowres_fits((int)g.M);
  };
  // `qkv_done` (non-null: the caller's stage can take it): the launch also produced the following block's q | k | v
  auto resnet = [&](int i, const float* in, int ldin, float* out, int ldo, const BtbW* follow = nullptr, bool* followed = nullptr,
                    bool* qkv_done = nullptr) -> int {
    if (qkv_done) *qkv_done = false;
    ConvStackScope scope;
    const ResnetW& r = e.res[i];
    // The whole resnet in ONE launch (rowres_kernel.h) where every piece has its row-owning form: a trunk input with a
    // measured bound, fragment-order weights for block1 | res_conv and for block2, the step's time embedding shared by all
    // rows (cfm_solve), a tile height that keeps the launch in as many rounds as the two it replaces.  JV_NO_RES_PAIR=1: two.
    // (a workgroup reads its neighbours' rows as halo, so the launch never writes the buffer it reads: the caller alternates the
    // trunk between w.h and w.h2 -- res_pair_ok / the mid-stage loop below)
    if (res_pair_ok(i, in, ldin) && out != in) {
      RowResArgs a{};
      a.A = in; a.lda = ldin; a.a_rows = g.a_rows; a.M = (int)g.M; a.Cin = ldin;
      a.rowmask = w.rowmask;
      a.amax_in = slots_of(in); a.slot_G = FLOW_G; a.slot_S = g.uoff ? -1 : g.S; a.slot_nb = B2; a.row_slot = w.row_sample;
      a.Wf1 = r.wf4; a.wf1_plane = 256L * 4 * ldin;
      a.cs1 = r.block1.colscale; a.b1 = r.block1.bias; a.ln1_g = r.ln1.g; a.ln1_b = r.ln1.b;
      a.csr = r.res.colscale; a.br = r.res.bias;
      a.temb = w.temb + i * 256; a.h2_bound = r.h2_bound; a.ln_eps = 1e-5f;
      a.Wf2 = r.block2.wf; a.wf2_plane = 256L * 3 * 256;
      a.cs2 = r.block2.colscale; a.b2 = r.block2.bias; a.ln2_g = r.ln2.g; a.ln2_b = r.ln2.b;
      a.out = out; a.ldo = ldo;
      a.amax_out = slots_of(out);
      a.alg_rows = g.alg_rows ? g.alg_rows : (long)g.B2 * g.T;
      if (followed) *followed = false;
      if (follow && followed && !c.no_ln_fold && ldo == 256 && follow->qkv.w2 && follow->qkv.a_scale > 0.f) {
        a.lnf_out = reinterpret_cast<unsigned short*>(w.ln); a.lnf_plane = (long)w.rows_alloc * 256;
        a.lnf_g = follow->n1.g; a.lnf_b = follow->n1.b; a.lnf_scale = follow->qkv.a_scale;
        *followed = true;
        if (qkv_done) {      // ... and its to_q | to_k | to_v over those planes, which then never leave LDS (btb_rg's buffers and scales)
          a.lnf_out = nullptr;
          a.Wqf = follow->qkv.wf; a.wqf_plane = (long)follow->qkv.N * follow->qkv.Cin; a.csq = follow->qkv.colscale;
          a.q = w.qkv; a.kv2 = reinterpret_cast<unsigned short*>(w.qkv + (long)w.rows_alloc * 512); a.kv2_plane = (long)w.rows_alloc * 1024;
          a.k_scale = follow->k_scale; a.v_scale = follow->v_scale;
          *qkv_done = true;
        }
      }
      return rowres(a, st);
    }
    ConvGemmArgs a = base_args(g, in, ldin, r.block1, w.h2, 256);
    causal3(a);
    a.ln = 1; a.ln_g = r.ln1.g; a.ln_b = r.ln1.b; a.ln_eps = 1e-5f; a.act = ACT_MISH;
    a.rowmask_out = w.rowmask;
    a.rowvec = w.temb + i * 256; a.row_sample = w.row_sample; a.rowvec_ld = g.temb_pre ? 0 : EST_NRES * 256;      // (0: one embedding for all rows)
    h3m(a, r.block1);
    track(a);      // -> h2
    bool res_done = false;
    JV_TRY(conv3(a, r.block1, nullptr, nullptr, &r, &res_done));
    if (!res_done) {      // (the tile kernels' route, the first resnet -- its input has no measured bound --, JV_NO_RES_FOLD=1)
      a = base_args(g, in, ldin, r.res, w.res, 256);
      a.rowmask_in = w.rowmask;
      h3m(a, r.res);
      JV_TRY(conv_gemm(a, 1, st));
    }
    a = base_args(g, w.h2, 256, r.block2, out, ldo);
    causal3(a);
    a.ln = 1; a.ln_g = r.ln2.g; a.ln_b = r.ln2.b; a.ln_eps = 1e-5f; a.act = ACT_MISH;
    a.rowmask_out = w.rowmask;
    a.res1 = w.res; a.ldr1 = 256;
    h3m(a, r.block2);
    track(a);      // -> h
    return conv3(a, r.block2, follow, followed);
  };
  // BasicTransformerBlock (transformer.py:355-443): h -> h, last GEMM may retarget its output
  // the four linears of a block run fp16x3 when registry.hip proved their input range (GemmW::a_scale)
  auto h3 = [&](ConvGemmArgs& a, const GemmW& m) {
    if (c.exact_range || !m.w2 || !(m.a_scale > 0.f)) return;
    a.W2 = m.w2; a.w2_plane = (long)m.n_rows * m.ldw; a.colscale = m.colscale; a.a_scale = m.a_scale;
  };
  // ... and then take their A operand as the fp16 planes their producer wrote into the same buffer (same bytes as fp32)
  const long R = (long)w.rows_alloc;
  auto pre = [&](const GemmW& m) { return c.dma_a && !c.exact_range && m.w2 && m.a_scale > 0.f; };
  auto planes_in = [&](ConvGemmArgs& a, float* buf, int C) {
    a.A2 = reinterpret_cast<const unsigned short*>(buf); a.a2_plane = R * C; a.lda2 = C;
  };
  auto ln_to = [&](const LnW& n, const GemmW& m, const float* h) -> int {
    if (pre(m)) return layernorm256_planes(h, reinterpret_cast<unsigned short*>(w.ln), R * 256, m.a_scale, n.g, n.b, 1e-5f, g.M, st);
    return layernorm_rows(h, nullptr, w.ln, n.g, n.b, 1e-5f, g.M, 256, nullptr, st);
  };
  auto btb = [&](const BtbW& b, const BtbW* next, bool ln_ready, float* h, float* out, int ldo) -> int {
    const bool sk = sk_blocks;      // split-K tails also write the next LayerNorm (fp32 rows) into w.ln
    if (!(sk && ln_ready)) JV_TRY(ln_to(b.n1, b.qkv, h));
    ConvGemmArgs a = base_args(g, w.ln, 256, b.qkv, w.qkv, 1536);
    h3(a, b.qkv);
    if (pre(b.qkv)) planes_in(a, w.ln, 256);
    JV_TRY(conv_gemm(a, 1, st));
    AttnArgs at{};
    at.qkv = w.qkv; at.ld = 1536; at.k_off = 512; at.v_off = 1024; at.out = w.att; at.ldo = 512;
    at.B = B2; at.H = EST_HEADS; at.G = FLOW_G; at.S = g.S; at.L = g.T; at.lens = w.lens2;
    at.chunk = c.attn_chunk;
    if (!c.exact_range && b.q_scale > 0.f) { at.q_scale = b.q_scale; at.k_scale = b.k_scale; at.v_scale = b.v_scale; }
    if (pre(b.out)) { at.out2 = reinterpret_cast<unsigned short*>(w.att); at.out2_plane = R * 512; at.out2_scale = b.out.a_scale; }
    JV_TRY(attention64(at, st));
    a = base_args(g, w.att, 512, b.out, h, 256);
    a.res1 = h; a.ldr1 = 256;
    track(a);      // -> h
    h3(a, b.out);
    if (pre(b.out)) planes_in(a, w.att, 512);
    if (sk) {
      JV_TRY(splitk(a, &b.n3, w.ln));      // h += to_out(att); ln = LayerNorm3(h)
    } else {
      JV_TRY(conv_gemm(a, 1, st));
      JV_TRY(ln_to(b.n3, b.ff1, h));
    }
    a = base_args(g, w.ln, 256, b.ff1, w.ff, 1024);
    a.act = ACT_GELU;
    h3(a, b.ff1);
    if (pre(b.ff1)) planes_in(a, w.ln, 256);
    if (pre(b.ff2) && a.W2) {      // the GELU epilogue writes ff2's operand (plane output exists on the fp16x3 lean path)
      a.out2 = reinterpret_cast<unsigned short*>(w.ff); a.out2_plane = R * 1024; a.ldo2 = 1024; a.out2_scale = b.ff2.a_scale;
    }
    const bool ff_planes = a.out2 != nullptr;
    JV_TRY(conv_gemm(a, 1, st));
    a = base_args(g, w.ff, 1024, b.ff2, out, ldo);
    a.res1 = h; a.ldr1 = 256;
    track(a);      // -> h / cat
    h3(a, b.ff2);
    if (ff_planes) planes_in(a, w.ff, 1024);
    if (sk) return splitk(a, (next && out == h) ? &next->n1 : nullptr, w.ln);      // + the next block's norm1
    return conv_gemm(a, 1, st);
  };
  // ---- the same block on the row-owning GEMM (rowgemm_kernel.h) when the batch fills the chip: every linear takes its A
  // operand as the fp16 planes its producer wrote (LayerNorm, attention, the previous linear's epilogue), to_out and
  // ff.net.2 add the residual AND run the LayerNorm that follows in their epilogue, ff.net.0 applies GELU and writes
  // ff.net.2's operand: four GEMM launches + attention per block, no stand-alone row-wise kernel except the first
  // LayerNorm of a stage.  `next`: the block that follows in the same stage (its norm1 runs in this block's last epilogue).
  const bool use_rg = !c.exact_range && !c.no_rowgemm && rowgemm_tile((int)g.M) > 0;
  auto rg_ok = [&](const BtbW& b) {
    return b.qkv.w2 && b.out.w2 && b.ff1.w2 && b.ff2.w2 && b.qkv.a_scale > 0.f && b.out.a_scale > 0.f && b.ff1.a_scale > 0.f &&
           b.ff2.a_scale > 0.f && b.q_scale > 0.f;
  };
  auto rg_args = [&](const float* planes, int K, const GemmW& m) {
    RowGemmArgs a{};
    a.A2 = reinterpret_cast<const unsigned short*>(planes); a.a2_plane = R * K; a.a_rows = g.a_rows; a.lda2 = K;
    a.M = (int)g.M; a.K = K; a.N = m.N;
    a.W2 = m.w2; a.w2_plane = (long)m.n_rows * m.ldw; a.ldw = m.ldw; a.colscale = m.colscale; a.a_scale = m.a_scale;
    a.Wf = m.wf; a.wf_plane = (long)m.N * m.Cin;
    a.bias = m.bias; a.ln_eps = 1e-5f; a.out2_scale = 1.f;
    a.alg_rows = g.alg_rows ? g.alg_rows : (long)g.B2 * g.T;
    return a;
  };
  auto rg_track = [&](RowGemmArgs& a) {
    a.amax_out = slots_of(a.out); a.row_slot = w.row_sample; a.row_mask = w.rowmask;
  };
  // Few row tiles (3 - 10 utterances of 300 frames: 64 - 192 workgroups of 32 rows on 256 CUs): a workgroup's length is set
  // by the weights it streams through its CU's L2 port, not by its MFMAs, and a third of the fused block's steps are the
  // next block's q | k | v, whose six 256-column chunks need nothing from each other.  There the q | k | v phase leaves the
  // fused launch: phase B's epilogue writes the LayerNorm1 planes to HBM and rowgemm_wa runs with its chunks dealt over
  // qkv_split workgroups per row tile.  Same K order, same epilogue expressions: the same bits as the fused launch
  // (tests/test_gpu_pipeline.py::test_split_qkv_equals_fused_block).  JV_NO_QKV_SPLIT=1: fused at every batch size.
  // The stand-alone launch takes the TALLEST tile: what a launch requests from L2 is (row tiles) x (weight bytes) -- at 152
  // tiles of 32 rows 228 MB, and dealing the chunks of those tiles out moved it only from 27 to 21 us -- so 80-row tiles (61
  // of them at 8 utterances, 92 MB) with as many column groups as fit one round of the chip: 19 us.
  int qkv_split = 1, qkv_rt = 0;
  if (use_rg && !c.no_qkv_split) {
    const int rt = rowgemm_tile((int)g.M);
    const long wgs = rt > 0 ? cdivl(g.M, 16 * rt) : 0;
    if (wgs > 0 && wgs <= 192) {
      qkv_rt = 5;
      const long tiles = cdivl(g.M, 80);
      qkv_split = tiles * 6 <= 256 ? 6 : tiles * 3 <= 256 ? 3 : 2;
    }
  }
  // `qkv_ready`: the previous block's fused launch has already produced this block's q | k | v (rowblock_kernel.h);
  // `qkv_next` (out): this block's launch produced the next block's
  auto ffn_fusable = [&](const BtbW& b) {
    return c.rg_ff1 && !c.no_ffn_fuse && b.ff1.wf && b.ff2.wf && b.ff1.N == 1024 && b.ff1.Cin == 256 && b.ff2.N == 256 && b.ff2.Cin == 1024;
  };
  auto btb_rg = [&](const BtbW& b, const BtbW* next, bool ln_ready, bool qkv_ready, bool* qkv_next, float* h, float* out, int ldo) -> int {
    *qkv_next = false;
    if (!ln_ready && !qkv_ready)
      JV_TRY(layernorm256_planes(h, reinterpret_cast<unsigned short*>(w.ln), R * 256, b.qkv.a_scale, b.n1.g, b.n1.b, 1e-5f, g.M, st));
    // q | k | v = to_q/k/v(ln): q as fp32 rows [R,512] at the head of the qkv buffer, k and v as fp16 planes [2][R][1024]
    // behind it (same bytes as [R,1536] fp32), scaled for the attention kernel, which then splits nothing
    RowGemmArgs a = rg_args(w.ln, 256, b.qkv);
    unsigned short* const kv2 = reinterpret_cast<unsigned short*>(w.qkv + R * 512);
    AttnArgs at{};
    at.qkv = w.qkv; at.out = w.att; at.ldo = 512;
    at.B = B2; at.H = EST_HEADS; at.G = FLOW_G; at.S = g.S; at.L = g.T; at.lens = w.lens2; at.uoff = g.uoff;
    at.chunk = c.attn_chunk;
    at.q_scale = b.q_scale; at.k_scale = b.k_scale; at.v_scale = b.v_scale;
    at.out2 = reinterpret_cast<unsigned short*>(w.att); at.out2_plane = R * 512; at.out2_scale = b.out.a_scale;
    if (!c.no_attn_planes) {
      if (!qkv_ready) {
        a.nsplit = qkv_split; a.rt = qkv_rt;
        a.out = w.qkv; a.ldo = 512;
        a.out2 = kv2; a.out2_plane = R * 1024; a.ldo2 = 1024; a.out2_scale = b.k_scale; a.out2_scale2 = b.v_scale;
        JV_TRY(rowgemm(a, RG_QKV, st));
      }
      at.ld = 512; at.kv2 = kv2; at.kv2_plane = R * 1024; at.kv_ld = 1024;
      if (c.attn_single && attention64_single_fits(at)) JV_TRY(attention64_single(at, st));      // one wave per SIMD, 160 queries per wave (whole-utterance attention; JV_NO_ATTN_SINGLE: attn64_pl)
      else if (at.chunk == 0 && c.attn_rows) JV_TRY(attention64_rows(at, st));      // one workgroup per head, 80 queries per wave (opt-in)
      else JV_TRY(attention64_planes(at, st));
    } else {
      a.out = w.qkv; a.ldo = 1536;
      JV_TRY(rowgemm(a, RG_PLAIN, st));
      at.ld = 1536; at.k_off = 512; at.v_off = 1024;
      JV_TRY(attention64(at, st));
    }
    if (!c.no_block_fuse && !c.no_attn_planes && ffn_fusable(b) && b.out.wf && b.out.N == 256 && b.out.Cin == 512 &&
        (!next || (next->qkv.wf && next->qkv.N == 1536 && next->qkv.Cin == 256 && !next->qkv.bias))) {
      // to_out -> LayerNorm3 -> feed-forward (-> the next block's LayerNorm1 -> q | k | v) in ONE launch on the same rows: the
      // LayerNorm planes never leave LDS (rowblock_kernel.h)
      RowBlockArgs f{};
      f.A2 = reinterpret_cast<const unsigned short*>(w.att); f.a2_plane = R * 512; f.a_rows = g.a_rows; f.M = (int)g.M;
      f.Wof = b.out.wf; f.wof_plane = (long)b.out.N * b.out.Cin; f.cso = b.out.colscale; f.bo = b.out.bias; f.a_scale_o = b.out.a_scale;
      f.h = h; f.ln3_g = b.n3.g; f.ln3_b = b.n3.b;
      f.W1f = b.ff1.wf; f.w1f_plane = (long)b.ff1.N * b.ff1.Cin; f.cs1 = b.ff1.colscale; f.b1 = b.ff1.bias; f.a_scale1 = b.ff1.a_scale;
      f.h_scale = b.ff2.a_scale;
      f.W2f = b.ff2.wf; f.w2f_plane = (long)b.ff2.N * b.ff2.Cin; f.cs2 = b.ff2.colscale; f.b2 = b.ff2.bias;
      f.out = out; f.ldo = ldo;
      f.amax_h = slots_of(h); f.amax_out = slots_of(out); f.row_slot = w.row_sample; f.row_mask = w.rowmask;
      f.alg_rows = g.alg_rows ? g.alg_rows : (long)g.B2 * g.T;
      const bool follows = next && out == h;
      const bool qkv = follows && qkv_split <= 1;
      if (follows) {
        f.ln1_g = next->n1.g; f.ln1_b = next->n1.b; f.a_scale_q = next->qkv.a_scale;
      }
      if (qkv) {
        f.Wqf = next->qkv.wf; f.wqf_plane = (long)next->qkv.N * next->qkv.Cin; f.csq = next->qkv.colscale;
        f.q = w.qkv; f.kv2 = kv2; f.kv2_plane = R * 1024; f.k_scale = next->k_scale; f.v_scale = next->v_scale;
      } else if (follows) {
        // few row tiles: the next block's LayerNorm1 planes leave through HBM and its q | k | v runs as its own launch with
        // the column chunks dealt over the idle CUs (qkv_split, above)
        f.ln_out = reinterpret_cast<unsigned short*>(w.ln); f.ln_out_plane = R * 256;
      }
      *qkv_next = qkv;
      if (tuning_env("JV_RB_STAMPS")) {
        if (!w.rb_stamps) JV_TRY(ws_alloc(c, 2 * 1024 * 48 * sizeof(unsigned long long), reinterpret_cast<void**>(&w.rb_stamps)));
        f.stamps = w.rb_stamps + (qkv ? 0 : 1024 * 48);
        JV_HIP(hipMemsetAsync(f.stamps, 0, 1024 * 48 * sizeof(unsigned long long), st));      // the stamps are atomic maxima
      }
      return rowblock(f, qkv, st);
    }
    a = rg_args(w.att, 512, b.out);      // h += to_out(att); ln = LayerNorm3(h)
    a.out = h; a.ldo = 256; a.res = h; a.ldr = 256;
    a.out2 = reinterpret_cast<unsigned short*>(w.ln); a.out2_plane = R * 256; a.ldo2 = 256; a.out2_scale = b.ff1.a_scale;
    a.ln_g = b.n3.g; a.ln_b = b.n3.b;
    rg_track(a);
    JV_TRY(rowgemm(a, RG_RES_LN, st));
    if (ffn_fusable(b)) {
      // the feed-forward pair in one launch (rowffn_kernel): the 1024-wide hidden tile never leaves LDS
      RowFfnArgs f{};
      f.A2 = reinterpret_cast<const unsigned short*>(w.ln); f.a2_plane = R * 256; f.a_rows = g.a_rows; f.lda2 = 256; f.M = (int)g.M;
      f.W1f = b.ff1.wf; f.w1f_plane = (long)b.ff1.N * b.ff1.Cin; f.cs1 = b.ff1.colscale; f.b1 = b.ff1.bias; f.a_scale1 = b.ff1.a_scale;
      f.h_scale = b.ff2.a_scale;
      f.W2f = b.ff2.wf; f.w2f_plane = (long)b.ff2.N * b.ff2.Cin; f.cs2 = b.ff2.colscale; f.b2 = b.ff2.bias;
      f.out = out; f.ldo = ldo; f.res = h; f.ldr = 256;
      f.ln_eps = 1e-5f; f.out2_scale = 1.f;
      if (!c.exact_range) { f.amax_out = slots_of(out); f.row_slot = w.row_sample; f.row_mask = w.rowmask; }
      f.alg_rows = g.alg_rows ? g.alg_rows : (long)g.B2 * g.T;
      if (next && out == h) {
        f.ln = 1; f.out2 = reinterpret_cast<unsigned short*>(w.ln); f.out2_plane = R * 256; f.ldo2 = 256; f.out2_scale = next->qkv.a_scale;
        f.ln_g = next->n1.g; f.ln_b = next->n1.b;
      }
      return rowffn(f, st);
    }
    if (c.rg_ff1) {
      a = rg_args(w.ln, 256, b.ff1);       // ff = gelu(ff.net.0(ln))
      a.out2 = reinterpret_cast<unsigned short*>(w.ff); a.out2_plane = R * 1024; a.ldo2 = 1024; a.out2_scale = b.ff2.a_scale;
      JV_TRY(rowgemm(a, RG_GELU_PL, st));
    } else {
      // JV_TILE_FF1: the tile kernel, which beat the first row-owning kernel here (both operands through LDS: 62 us against
      // 56); with the weights loaded straight into registers the row-owning kernel takes 56 alone and the whole pass 188 ms
      // against 199 (same box, back to back).  Same planes in and out.
      ConvGemmArgs t = base_args(g, w.ln, 256, b.ff1, w.ff, 1024);
      t.act = ACT_GELU;
      h3(t, b.ff1);
      planes_in(t, w.ln, 256);
      t.out2 = reinterpret_cast<unsigned short*>(w.ff); t.out2_plane = R * 1024; t.ldo2 = 1024; t.out2_scale = b.ff2.a_scale;
      JV_TRY(conv_gemm(t, 1, st));
    }
    a = rg_args(w.ff, 1024, b.ff2);      // out = h + ff.net.2(ff); the next block's norm1 of it
    a.out = out; a.ldo = ldo; a.res = h; a.ldr = 256;
    rg_track(a);
    if (next && out == h) {
      a.out2 = reinterpret_cast<unsigned short*>(w.ln); a.out2_plane = R * 256; a.ldo2 = 256; a.out2_scale = next->qkv.a_scale;
      a.ln_g = next->n1.g; a.ln_b = next->n1.b;
      return rowgemm(a, RG_RES_LN, st);
    }
    return rowgemm(a, RG_RES, st);
  };
  // the four blocks of a stage; the last one may retarget its output (skip / concat buffer)
  auto stage_all_rg = [&](const BtbW* blk) {
    bool all = use_rg;
    for (int j = 0; j < EST_NBLK; ++j) all = all && rg_ok(blk[j]);
    return all;
  };
  // `ln_first`: the first block's norm1 planes are already in w.ln (written by the resnet's last convolution)
  // `qkv_first`: ... and its q | k | v too (the whole-resnet launch's product 3)
  auto stage_blocks = [&](const BtbW* blk, float* h, float* last_out, int last_ldo, bool ln_first = false, bool qkv_first = false) -> int {
    const bool all = stage_all_rg(blk);
    if (g.uoff && (!all || c.no_attn_planes)) return fail(JV_ERR_STATE, "flow: the compact geometry exists on the row-owning kernels only");
    bool qkv_ready = qkv_first && all && !c.no_attn_planes;
    for (int j = 0; j < EST_NBLK; ++j) {
      const bool last = j == EST_NBLK - 1;
      if (all) JV_TRY(btb_rg(blk[j], last ? nullptr : &blk[j + 1], j > 0 || ln_first, qkv_ready, &qkv_ready, h, last ? last_out : h, last ? last_ldo : 256));
      else JV_TRY(btb(blk[j], last ? nullptr : &blk[j + 1], j > 0 || ln_first, h, last ? last_out : h, last ? last_ldo : 256));
    }
    return JV_OK;
  };

  // down: resnet -> 4 blocks (result doubles as the skip) -> causal conv
  // (a stage on the row-owning kernels takes its first norm1 from the resnet's last convolution)
  bool lnf = false, qkv0 = false;
  // may the resnet ahead of a stage run the stage's first q | k | v?  The full-chip regime of the row-owning blocks (no column split)
  auto qkv_of = [&](const BtbW* blk) -> bool* {
    const GemmW& m = blk[0].qkv;
    return (!c.no_res_qkv && !c.no_attn_planes && stage_all_rg(blk) && qkv_split <= 1 && m.wf && m.N == 1536 && m.Cin == 256 && !m.bias) ? &qkv0 : nullptr;
  };
  auto follow_of = [&](const BtbW* blk) -> const BtbW* { return (stage_all_rg(blk) || sk_blocks) ? blk : nullptr; };
  JV_TRY(resnet(0, w.xin, 320, w.h, 256, follow_of(e.blk[0]), &lnf, qkv_of(e.blk[0])));
  JV_TRY(stage_blocks(e.blk[0], w.h, skip, 512, lnf, qkv0));
  {
    ConvStackScope scope;
    ConvGemmArgs a = base_args(g, skip, 512, e.down_conv, w.h, 256);
    causal3(a);
    h3m(a, e.down_conv);
    track(a);
    JV_TRY(conv3(a, e.down_conv));
  }
  // mid x12; the last block writes straight into columns [0,256) of the concat buffer
  // (the whole-resnet launch must not write the buffer it reads: the trunk then alternates between w.h and w.h2 -- h2 is free,
  // the launch keeps block1's output in LDS -- and an even number of mid stages brings it back to w.h; both have bound slots)
  float* trunk = w.h;
  const bool mid_pair = (EST_NMID % 2 == 0);
  for (int i = 1; i <= EST_NMID; ++i) {
    float* const dst = (mid_pair && res_pair_ok(i, trunk, 256)) ? (trunk == w.h ? w.h2 : w.h) : trunk;
    JV_TRY(resnet(i, trunk, 256, dst, 256, follow_of(e.blk[i]), &lnf, qkv_of(e.blk[i])));
    trunk = dst;
    JV_TRY(stage_blocks(e.blk[i], trunk, i == EST_NMID ? w.cat : trunk, i == EST_NMID ? 512 : 256, lnf, qkv0));
  }
  if (trunk != w.h) return fail(JV_ERR_STATE, "flow: the mid stages left the trunk in the scratch buffer");
  // up: resnet(cat[x, skip]) -> 4 blocks -> causal conv -> final block -> 1x1 projection
  JV_TRY(resnet(EST_NRES - 1, w.cat, 512, w.h, 256, follow_of(e.blk[EST_NRES - 1]), &lnf, qkv_of(e.blk[EST_NRES - 1])));
  JV_TRY(stage_blocks(e.blk[EST_NRES - 1], w.h, w.h, 256, lnf, qkv0));
  {
    ConvStackScope scope;
    ConvGemmArgs a = base_args(g, w.h, 256, e.up_conv, w.h2, 256);
    causal3(a);
    h3m(a, e.up_conv);
    track(a);
    JV_TRY(conv3(a, e.up_conv));
    a = base_args(g, w.h2, 256, e.final_conv, w.h, 256);
    causal3(a);
    a.ln = 1; a.ln_g = e.final_ln.g; a.ln_b = e.final_ln.b; a.ln_eps = 1e-5f; a.act = ACT_MISH;
    a.rowmask_out = w.rowmask;
    h3m(a, e.final_conv);
    track(a);
    JV_TRY(conv3(a, e.final_conv));
    a = base_args(g, w.h, 256, e.final_proj, w.d, 80);
    a.rowmask_in = w.rowmask;
    a.rowmask_out = w.rowmask;
    h3m(a, e.final_proj);
    JV_TRY(conv_gemm(a, 1, st));
  }
  return JV_OK;
}

// may a ragged batch of M rows take the compact geometry?  Only the route whose every kernel knows it: the row-owning kernels with
// the plane attention (flow_compact_ok mirrors estimator_body's own predicates)
bool flow_compact_ok(const Context& c, long M) {
  if (c.no_compact || c.exact_range || c.no_rowgemm || c.no_attn_planes || c.attn_chunk != 0 || c.attn_rows || rowgemm_tile((int)M) <= 0) return false;
  const EstimatorW& e = c.est;
  for (int i = 0; i < EST_NRES; ++i)
    for (int j = 0; j < EST_NBLK; ++j) {
      const BtbW& b = e.blk[i][j];
      if (!(b.qkv.w2 && b.out.w2 && b.ff1.w2 && b.ff2.w2 && b.qkv.a_scale > 0.f && b.out.a_scale > 0.f && b.ff1.a_scale > 0.f &&
            b.ff2.a_scale > 0.f && b.q_scale > 0.f))
        return false;
    }
  return true;
}

int check_shape(Context& c, int B2, int T) {
  if (!c.ready[MODEL_TTS]) return fail(JV_ERR_STATE, "tts weights not finalized");
  if (B2 < 1 || T < 1) return fail(JV_ERR_ARG, "batch and frame count must be positive");
  if (B2 > 2 * c.max_batch || T > c.max_frames || flow_rows(B2, T) + 128 > c.flow->rows_alloc)
    return fail(JV_ERR_SHAPE, "batch/frames exceed the capacity given to jv_create");
  return JV_OK;
}

}  // namespace

// In a resnet whose input and output are the same buffer (mid blocks), block2 writes `out` only after
// block1 and res_conv have consumed `in`; stream order makes that safe.

// valid frames per row from the reference's float mask [B2,1,T] (1 = frame, 0 = padding; make_pad_mask gives prefixes)
__global__ void mask_to_lens_kernel(const float* __restrict__ mask, int T, int* __restrict__ lens) {
  const int b = blockIdx.x;
  int n = 0;
  for (int t = threadIdx.x; t < T; t += 64) n += mask[(long)b * T + t] != 0.f ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
  if (threadIdx.x == 0) lens[b] = n;
}

int flow_estimator(Context& c, const float* x, const int* lens_dev, const float* mu, const float* t_dev, const float* spks,
                   const float* cond, int B2, int T, float* out, hipStream_t st, const float* mask_f32) {
  JV_TRY(check_shape(c, B2, T));
  FlowWs& w = *c.flow;
  Geo g{B2, T, T + FLOW_GAP, flow_rows(B2, T), w.rows_alloc, w.t_dev, 1};
  // channels-first [B2,80,T] inputs -> row buffers (reusing x/mu/cond, which hold 2B utterances here)
  JV_TRY(cf_to_rows(x, 80L * T, T, B2, 80, T, w.x, 80, 0, FLOW_G, g.S, 1.f, nullptr, st));
  JV_TRY(cf_to_rows(mu, 80L * T, T, B2, 80, T, w.mu, 80, 0, FLOW_G, g.S, 1.f, nullptr, st));
  JV_TRY(cf_to_rows(cond, 80L * T, T, B2, 80, T, w.cond, 80, 0, FLOW_G, g.S, 1.f, nullptr, st));
  if (mask_f32) hipLaunchKernelGGL(mask_to_lens_kernel, dim3(B2), dim3(64), 0, st, mask_f32, T, w.lens2);
  else if (lens_dev) JV_HIP(hipMemcpyAsync(w.lens2, lens_dev, sizeof(int) * B2, hipMemcpyDeviceToDevice, st));
  else JV_TRY(fill_int(w.lens2, T, B2, st));
  JV_TRY(row_meta(w.rowmask, w.row_sample, w.lens2, B2, 1, FLOW_G, g.S, T, w.rows_alloc, 1, 0, st));
  JV_HIP(hipMemcpyAsync(w.t_dev, t_dev, sizeof(float) * B2, hipMemcpyDeviceToDevice, st));
  JV_TRY(assemble_xin_plain(w.x, w.mu, spks, w.cond, w.xin, B2, FLOW_G, g.S, T, g.M, st));
  JV_HIP(hipMemsetAsync(w.amax, 0, sizeof(float) * 3 * w.amax_stride, st));
  JV_TRY(estimator_body(c, g, st));
  return rows_to_cf(w.d, 80, 0, FLOW_G, g.S, out, 80L * T, B2, 80, T, nullptr, st);
}

int cfm_solve(Context& c, const float* mu, const int* lens_dev, const float* spks, const float* cond, int B, int T,
              int n_timesteps, float temperature, const float* t_span_host, float* mel, hipStream_t st) {
  JV_TRY(check_shape(c, 2 * B, T));
  if (!c.noise_loaded) return fail(JV_ERR_STATE, "CFM noise tensor not loaded (jv_load_noise)");
  if (T > NOISE_FRAMES) return fail(JV_ERR_SHAPE, "more frames than the fixed noise tensor holds (15000)");
  FlowWs& w = *c.flow;
  if (n_timesteps < 1 || n_timesteps > w.max_steps) return fail(JV_ERR_ARG, "n_timesteps out of range");
  const int B2 = 2 * B;
  Geo g{B2, T, T + FLOW_GAP, flow_rows(B2, T), w.rows_alloc, nullptr, 0};

  // cosine schedule and the reference's running (t, dt) recurrence, in fp32 (flow_matching.py:230,260-263,387-389)
  std::vector<float> ts(n_timesteps + 1), tt(n_timesteps), dts(n_timesteps);
  if (t_span_host) {
    for (int i = 0; i <= n_timesteps; ++i) ts[i] = t_span_host[i];
  } else {
    const int steps = n_timesteps + 1;
    const float step = 1.0f / (float)(steps - 1);
    for (int i = 0; i < steps; ++i) {
      const float lin = i < steps / 2 ? step * (float)i : 1.0f - step * (float)(steps - 1 - i);
      ts[i] = 1.0f - cosf(lin * 0.5f * 3.14159265358979323846f);
    }
  }
  {
    float t = ts[0], dt = ts[1] - ts[0];
    for (int s = 1; s <= n_timesteps; ++s) {
      tt[s - 1] = t;
      dts[s - 1] = dt;
      t = t + dt;
      if (s < n_timesteps) dt = ts[s + 1] - t;
    }
  }
  JV_HIP(hipMemcpyAsync(w.t_table, tt.data(), sizeof(float) * n_timesteps, hipMemcpyHostToDevice, st));
  JV_HIP(hipMemcpyAsync(w.dt_table, dts.data(), sizeof(float) * n_timesteps, hipMemcpyHostToDevice, st));
  // Ragged batch?  The lengths come down with the synchronisation below (which the staging vectors need anyway): B ints.
  const bool ragged_candidate = lens_dev && B > 1 && flow_compact_ok(c, g.M);
  if (ragged_candidate) JV_HIP(hipMemcpyAsync(w.h_lens, lens_dev, sizeof(int) * B, hipMemcpyDeviceToHost, st));
  // pageable-host staging vectors die at scope exit: make sure the copies have been consumed
  JV_HIP(hipStreamSynchronize(st));
  // COMPACT geometry: every utterance (and its CFG twin) gets its own frames + the gap, nothing is padded to the longest;
  // taken when it saves at least 8 % of the rows and the shorter batch still fills the row-owning kernels.  All per-row
  // arithmetic is the uniform geometry's (a row's sums do not depend on where the row sits): the same bits per utterance.
  if (ragged_candidate) {
    long r = FLOW_G, frames = 0;
    for (int b2 = 0; b2 < B2; ++b2) {
      const int len = std::min(std::max(w.h_lens[b2 % B], 0), T);
      w.h_uoff[b2] = (int)r;
      r += len + FLOW_GAP;
      frames += len;
    }
    w.h_uoff[B2] = (int)r;
    if (r * 100 <= g.M * 92 && flow_compact_ok(c, r)) {
      JV_HIP(hipMemcpyAsync(w.uoff, w.h_uoff, sizeof(int) * (B2 + 1), hipMemcpyHostToDevice, st));      // (pinned: stays valid)
      g.M = r; g.uoff = w.uoff; g.alg_rows = frames;
    }
  }

  // per-solve preparation: lengths (duplicated for the CFG twin rows), masks, row-layout mu / cond / z
  if (lens_dev) {
    JV_HIP(hipMemcpyAsync(w.lens2, lens_dev, sizeof(int) * B, hipMemcpyDeviceToDevice, st));
    JV_HIP(hipMemcpyAsync(w.lens2 + B, lens_dev, sizeof(int) * B, hipMemcpyDeviceToDevice, st));
  } else {
    JV_TRY(fill_int(w.lens2, T, B2, st));
  }
  const int* const clens = g.uoff ? w.lens2 : nullptr;      // (compact: only an utterance's own frames are written)
  JV_TRY(row_meta(w.rowmask, w.row_sample, w.lens2, B2, 1, FLOW_G, g.S, T, w.rows_alloc, 1, 0, st, g.uoff));
  JV_TRY(cf_to_rows(mu, 80L * T, T, B, 80, T, w.mu, 80, 0, FLOW_G, g.S, 1.f, clens, st, g.uoff));
  JV_TRY(cf_to_rows(cond, 80L * T, T, B, 80, T, w.cond, 80, 0, FLOW_G, g.S, 1.f, clens, st, g.uoff));
  // z = rand_noise[:, :, :T] * temperature, the same prefix for every utterance (flow_matching.py:385)
  JV_TRY(cf_to_rows(c.noise, 0, NOISE_FRAMES, B, 80, T, w.x, 80, 0, FLOW_G, g.S, temperature, clens, st, g.uoff));

  JV_HIP(hipMemsetAsync(w.step_ctr, 0, sizeof(int), st));
  JV_HIP(hipMemsetAsync(w.amax, 0, sizeof(float) * 3 * w.amax_stride, st));      // trunk bounds: maxima over the whole solve
  JV_HIP(hipMemcpyAsync(w.spks, spks, sizeof(float) * 80 * B, hipMemcpyDeviceToDevice, st));
  g.t_ptr = w.t_cur;   // the same t for all 2B rows (stride 0)
  // Every step's t is on the device already and is the same for all rows: the n embeddings are three GEMMs of n rows HERE
  // instead of three GEMMs of 2B identical rows inside every step -- K = 1024 contractions of one row tile, 20 - 48 us each
  // at any batch size, 1.2 ms of serial launches per solve (JV_NO_TEMB_PRE=1: per step, as jv_flow_estimator_masked does)
  g.temb_pre = !c.no_temb_pre && n_timesteps <= TS_MAX;
  if (g.temb_pre) JV_TRY(time_embedding(c, w.t_table, 1, n_timesteps, w.ts_sin, w.ts_1, w.ts_mish, w.ts_emb, st));
  auto euler_step = [&](hipStream_t s) -> int {
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(256), 0, s, w.t_table, w.dt_table, w.step_ctr, w.t_cur, w.dt_cur,
                       g.temb_pre ? w.ts_emb : nullptr, w.temb);
    JV_TRY(assemble_xin(w.x, w.mu, w.spks, w.cond, w.xin, B, FLOW_G, g.S, T, g.M, s, g.uoff, w.row_sample, w.rowmask));
    JV_TRY(estimator_body(c, g, s));
    return euler_cfg(w.x, w.d, B, FLOW_G, g.S, T, w.dt_cur, 0, 0.7f, s, g.uoff, w.lens2);
  };
  // The in-library profiler brackets every launch with events, which a capture would turn into graph nodes: eager then.
  const bool use_graph = c.step_graphs && !prof_on() && n_timesteps > 1 && !g.uoff;      // (a captured step is keyed by (B, T): uniform geometry only)
  if (!use_graph) {
    for (int s = 0; s < n_timesteps; ++s) JV_TRY(euler_step(st));
  } else {
    FlowWs::StepGraph* sg = nullptr;
    for (auto& e : w.graphs)
      if (e.B == B && e.T == T && e.chunk == c.attn_chunk && e.pre == (int)g.temb_pre) sg = &e;
    int first = 0;
    if (!sg) {
      // first solve of this geometry: step 0 runs eagerly (it also performs the one-time kernel attribute setup),
      // step 1 is captured; it does not execute during capture, so the replay loop starts from it
      JV_TRY(euler_step(st));
      first = 1;
      JV_HIP(hipEventRecord(w.ev_in, st));
      JV_HIP(hipStreamWaitEvent(w.gstream, w.ev_in, 0));
      JV_HIP(hipStreamBeginCapture(w.gstream, hipStreamCaptureModeThreadLocal));
      const int rc = euler_step(w.gstream);
      hipGraph_t graph = nullptr;
      const hipError_t ce = hipStreamEndCapture(w.gstream, &graph);
      if (rc != JV_OK) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc;
      }
      if (ce != hipSuccess || !graph) return fail(JV_ERR_HIP, "cfm_solve: stream capture of the Euler step failed");
      hipGraphExec_t exec = nullptr;
      if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGraphDestroy(graph);
        return fail(JV_ERR_HIP, "cfm_solve: hipGraphInstantiate failed");
      }
      if (w.graphs.size() >= 16) {   // bounded cache: drop the oldest geometry
        (void)hipGraphExecDestroy(w.graphs.front().exec);
        (void)hipGraphDestroy(w.graphs.front().graph);
        w.graphs.erase(w.graphs.begin());
      }
      w.graphs.push_back({B, T, c.attn_chunk, (int)g.temb_pre, graph, exec});
      sg = &w.graphs.back();
    } else {
      JV_HIP(hipEventRecord(w.ev_in, st));
      JV_HIP(hipStreamWaitEvent(w.gstream, w.ev_in, 0));
    }
    for (int s = first; s < n_timesteps; ++s) JV_HIP(hipGraphLaunch(sg->exec, w.gstream));
    JV_HIP(hipEventRecord(w.ev_out, w.gstream));
    JV_HIP(hipStreamWaitEvent(st, w.ev_out, 0));
  }
  if (w.rb_stamps && tuning_env("JV_RB_STAMPS")) {      // tuning aid: phase breakdown of the LAST rowblock launch (a block without q|k|v)
    JV_HIP(hipStreamSynchronize(st));
    const int nwg = (int)std::min<long>(1024, cdivl(g.M, 16 * std::max(1, rowgemm_tile((int)g.M))));
    std::vector<unsigned long long> hs((size_t)nwg * 48);
    for (int v = 0; v < 2; ++v) {
      JV_HIP(hipMemcpy(hs.data(), w.rb_stamps + v * 1024 * 48, hs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      fprintf(stderr, "[rowblock stamps%s] %d workgroups; median s_memtime ticks (100 MHz) since start:", v ? "" : ",qkv", nwg);
      for (int i = 1; i < 48; ++i) {
        std::vector<long> d;
        for (int b = 0; b < nwg; ++b)
          if (hs[(size_t)b * 48 + i] > hs[(size_t)b * 48]) d.push_back((long)(hs[(size_t)b * 48 + i] - hs[(size_t)b * 48]));
        if (d.empty()) continue;
        std::sort(d.begin(), d.end());
        fprintf(stderr, " %d:%ld", i, d[d.size() / 2]);
      }
      fprintf(stderr, "\n");
    }
    if (w.rc_stamps) {
      std::vector<unsigned long long> hc((size_t)nwg * 8);
      JV_HIP(hipMemcpy(hc.data(), w.rc_stamps, hc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      fprintf(stderr, "[rowconv stamps] %d workgroups; median cycles since start:", nwg);
      for (int i = 1; i < 8; ++i) {
        std::vector<long> d;
        for (int b = 0; b < nwg; ++b)
          if (hc[(size_t)b * 8 + i] > hc[(size_t)b * 8]) d.push_back((long)(hc[(size_t)b * 8 + i] - hc[(size_t)b * 8]));
        if (d.empty()) continue;
        std::sort(d.begin(), d.end());
        fprintf(stderr, " %d:%ld", i, d[d.size() / 2]);
      }
      fprintf(stderr, "\n");
    }
  }
  return rows_to_cf(w.x, 80, 0, FLOW_G, g.S, mel, 80L * T, B, 80, T, lens_dev ? w.lens2 : nullptr, st, g.uoff);
}

}  // namespace jv

namespace jv {
void flow_ws_destroy(Context& c) {
  if (c.flow) {
    flow_graphs_drop(c);
    if (c.flow->gstream) (void)hipStreamDestroy(c.flow->gstream);
    if (c.flow->ev_in) (void)hipEventDestroy(c.flow->ev_in);
    if (c.flow->ev_out) (void)hipEventDestroy(c.flow->ev_out);
    if (c.flow->h_lens) (void)hipHostFree(c.flow->h_lens);
    if (c.flow->h_uoff) (void)hipHostFree(c.flow->h_uoff);
  }
  delete c.flow;
  c.flow = nullptr;
}
}  // namespace jv
