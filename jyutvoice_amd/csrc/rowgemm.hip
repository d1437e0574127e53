// Host side of the row-owning fp16x3 GEMM (rowgemm_kernel.h): tile height choice and launch.
#include "rowconv_kernel.h"
#include "rowres_kernel.h"

namespace jv {

namespace {

__global__ void pack_wfrag_kernel(const unsigned short* __restrict__ w2, long w2_plane, int ldw, int NB, int KS,
                                  unsigned short* __restrict__ wf, long wf_plane) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= 2L * NB * KS * 64) return;
  const int lane = (int)(t & 63);
  const long f = t >> 6;
  const int ks = (int)(f % KS), nb = (int)((f / KS) % NB), pl = (int)(f / ((long)KS * NB));
  const uint4 v = *reinterpret_cast<const uint4*>(w2 + pl * w2_plane + (long)(nb * 16 + (lane & 15)) * ldw + ks * 32 + (lane >> 4) * 8);
  // k-step major: the 16 column blocks (x 1 KB) that a workgroup's eight waves load in one step are CONTIGUOUS.  Block major
  // ([N/16][KS]) put them KS KB apart -- 32 KB for ff.net.2 -- and every CU of an XCD then asked the same few L2 channels for
  // them at once: the phase stamps of rowblock_kernel read 1640 cycles per step on W2 against 840 on W1 (8 KB apart)
  *reinterpret_cast<uint4*>(wf + pl * wf_plane + (((long)ks * NB + nb) * 64 + lane) * 8) = v;
}

template <int RT, int EPI>
int rg_launch_wd(const RowGemmArgs& a, hipStream_t st) {
  static bool raised[64] = {};
  int dev = 0;
  JV_HIP(hipGetDevice(&dev));
  if (!raised[dev & 63]) {
    JV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&rowgemm_wd_kernel<RT, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               rgw_lds_bytes<RT>()));
    raised[dev & 63] = true;
  }
  hipLaunchKernelGGL((rowgemm_wd_kernel<RT, EPI>), dim3(cdiv(a.M, 16 * RT)), dim3(512), rgw_lds_bytes<RT>(), st, a);
  return JV_OK;
}

// A-resident form (rowgemm_wa_kernel): the multi-chunk epilogues when the whole A tile fits beside the slab
template <int RT, int EPI>
int rg_launch_wa(const RowGemmArgs& a, hipStream_t st) {
  static int raised[64] = {};      // per device: the LDS size the attribute was last raised to
  int dev = 0;
  JV_HIP(hipGetDevice(&dev));
  const int lds = rgwa_lds_bytes<RT>(a.K >> 5);
  if (raised[dev & 63] < lds) {
    JV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&rowgemm_wa_kernel<RT, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    raised[dev & 63] = lds;
  }
  const int nc = a.N >> 8;
  const int ns = a.nsplit > 1 ? (a.nsplit < nc ? a.nsplit : nc) : 1;      // column chunks dealt over grid.y (RowGemmArgs::nsplit)
  hipLaunchKernelGGL((rowgemm_wa_kernel<RT, EPI>), dim3(cdiv(a.M, 16 * RT), ns), dim3(512), lds, st, a);
  return JV_OK;
}

template <int RT, int EPI>
int rg_launch2(const RowGemmArgs& a, hipStream_t st) {
  static bool raised[64] = {};      // per device: the attribute belongs to the kernel's image on the current device
  int dev = 0;
  JV_HIP(hipGetDevice(&dev));
  if (!raised[dev & 63]) {
    JV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&rowgemm_kernel<RT, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               rg_lds_bytes<RT>()));
    raised[dev & 63] = true;
  }
  const bool prof = prof_on();
  if (prof) prof_begin(st);
  constexpr bool MULTI = EPI == RG_PLAIN || EPI == RG_GELU_PL || EPI == RG_QKV;
  const bool wdir = a.Wf && !(a.K & 63) && !dyn_env("JV_RG_WLDS");
  if (MULTI && wdir && a.N > 256 && rgwa_lds_bytes<RT>(a.K >> 5) <= 160 * 1024 && !dyn_env("JV_RG_NO_ARES")) {
    if constexpr (MULTI) JV_TRY((rg_launch_wa<RT, EPI>(a, st)));
  } else if (wdir) JV_TRY((rg_launch_wd<RT, EPI>(a, st)));
  else hipLaunchKernelGGL((rowgemm_kernel<RT, EPI>), dim3(cdiv(a.M, 16 * RT)), dim3(512), rg_lds_bytes<RT>(), st, a);
  if (prof) {
    static const std::string name = std::string("rowgemm_h3<") + std::to_string(16 * RT) + "x256" +
                                    (EPI == RG_GELU_PL ? ",gelu" : EPI == RG_RES ? ",res" : EPI == RG_RES_LN ? ",res,ln" : EPI == RG_QKV ? ",qkv" : "") + ">";
    const double rows = (double)(a.alg_rows > 0 ? a.alg_rows : a.M);
    const double bytes = 4.0 * (rows * a.K + (double)a.N * a.K + rows * a.N * ((EPI == RG_RES || EPI == RG_RES_LN) ? 2 : 1) +
                                (EPI == RG_RES_LN ? rows * 256 : 0));
    prof_end(st, name.c_str(), 2.0 * rows * a.N * a.K, bytes);
  }
  JV_HIP(hipGetLastError());
  return JV_OK;
}

template <int RT>
int rg_launch1(const RowGemmArgs& a, int epi, hipStream_t st) {
  switch (epi) {
    case RG_PLAIN: return rg_launch2<RT, RG_PLAIN>(a, st);
    case RG_GELU_PL: return rg_launch2<RT, RG_GELU_PL>(a, st);
    case RG_RES: return rg_launch2<RT, RG_RES>(a, st);
    case RG_RES_LN: return rg_launch2<RT, RG_RES_LN>(a, st);
    case RG_QKV: return rg_launch2<RT, RG_QKV>(a, st);
    default: return fail(JV_ERR_ARG, "rowgemm: unknown epilogue");
  }
}

}  // namespace

int pack_wfrag(const unsigned short* w2, long w2_plane, int ldw, int N, int K, unsigned short* wf, long wf_plane, hipStream_t st) {
  if (!w2 || !wf || (N & 15) || (K & 31) || (ldw & 7)) return fail(JV_ERR_ARG, "pack_wfrag: N % 16 == 0, K % 32 == 0, ldw % 8 == 0 required");
  const long total = 2L * (N >> 4) * (K >> 5) * 64;
  hipLaunchKernelGGL(pack_wfrag_kernel, dim3((unsigned)cdivl(total, 256)), dim3(256), 0, st, w2, w2_plane, ldw, N >> 4, K >> 5, wf,
                     wf_plane);
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// tile height (in 16-row units) for M rows: the fewest rounds of workgroups over the 256 CUs times the rows each round
// costs; ties go to the taller tile (each weight byte is then used for more rows).  0: use the tile kernels -- at most 2048
// rows (one to three utterances of 300 frames), where flow.hip splits the long contractions over K instead.
//
// Calibration (tools/regime_sweep.py, round 3, CFM loop alone at 300 frames, n = 4, one box; ms per pass, * = what this
// function picks):        utterances   tile kernels   rt 2     rt 3     rt 4     rt 5
//                              2          19.8*       24.6     26.4
//                              4          26.5        24.5*    27.8     30.4     34.0
//                              8          34.5        24.3*    29.2     32.9     36.8
//                             16          46.3        42.7     31.2*    35.4     39.5
//                             24          65.2        51.6     55.4     42.0*    46.0
//                             32          76.1        68.9     62.2     66.6     50.8*
//                             40          87.1        76.7     69.7*    74.2     80.6
//                             48         104.9        92.6     88.0     79.3*    85.9
//                             64         135.9       117.2    115.1    109.3     99.0*
// With the transformer block in one launch (rowblock_kernel.h) the row-owning form wins wherever it exists: round 2's two
// exceptions -- fewer than 96 workgroups, and more than one round of tiles shorter than 64 rows (the "hole" at 40
// utterances, which cost 25 % there) -- are gone, and the plain cost model picks the measured optimum at every point.
int rowgemm_tile(int M) {
  if (const char* f = dyn_env("JV_ROWGEMM_RT")) return atoi(f);
  if (M <= 2048) return 0;      // = flow.hip's PARTIAL_ROWS: the split-K regime
  int best = 0;
  long best_cost = 0;
  for (int rt = 2; rt <= 5; ++rt) {
    const long wgs = cdiv(M, 16 * rt);
    const long cost = cdivl(wgs, 256) * rt;
    if (!best || cost < best_cost || (cost == best_cost && rt > best)) { best = rt; best_cost = cost; }
  }
  return best;
}

// W-direct (rowconv_wd_kernel): fragment-order weights and an even number of 32-channel chunks
bool rowconv_w_direct(const RowConvArgs& a) { return a.Wf && !((a.Cin >> 5) & 1) && !dyn_env("JV_RG_WLDS"); }

namespace {
template <int RT>
int rc_launch(const RowConvArgs& a, hipStream_t st) {
  static bool raised[64] = {};
  int dev = 0;
  JV_HIP(hipGetDevice(&dev));
  if (!raised[dev & 63]) {
    JV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&rowconv_kernel<RT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               rc_lds_bytes<RT>()));
    raised[dev & 63] = true;
  }
  static bool raised_wd[64] = {};
  const bool wdir = rowconv_w_direct(a);
  if (a.ln2_out && !wdir) return fail(JV_ERR_ARG, "rowconv: the following LayerNorm exists on the W-direct kernel only");
  if (a.res_out && !wdir) return fail(JV_ERR_ARG, "rowconv: the folded res_conv exists on the W-direct kernel only");
  if (wdir && !raised_wd[dev & 63]) {
    JV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&rowconv_wd_kernel<RT, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               rcw_lds_bytes<RT>()));
    JV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&rowconv_wd_kernel<RT, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               rcw_lds_bytes<RT>()));
    raised_wd[dev & 63] = true;
  }
  const bool prof = prof_on();
  if (prof) prof_begin(st);
  if (wdir && a.res_out) hipLaunchKernelGGL((rowconv_wd_kernel<RT, true>), dim3(cdiv(a.M, 16 * RT)), dim3(512), rcw_lds_bytes<RT>(), st, a);
  else if (wdir) hipLaunchKernelGGL((rowconv_wd_kernel<RT, false>), dim3(cdiv(a.M, 16 * RT)), dim3(512), rcw_lds_bytes<RT>(), st, a);
  else hipLaunchKernelGGL((rowconv_kernel<RT>), dim3(cdiv(a.M, 16 * RT)), dim3(512), rc_lds_bytes<RT>(), st, a);
  if (prof) {
    static const std::string name = std::string("rowconv_h3<") + std::to_string(16 * RT) + "x256,k3>";
    static const std::string name_res = std::string("rowconv_h3<") + std::to_string(16 * RT) + "x256,k3+res>";
    const double rows = (double)(a.alg_rows > 0 ? a.alg_rows : a.M);
    const int taps = a.res_out ? 4 : 3;      // (the folded 1 x 1 res_conv is a fourth tap's worth of work and one more output)
    const double bytes = 4.0 * (rows * a.Cin + 256.0 * taps * a.Cin + rows * 256 * ((a.res ? 2 : 1) + (a.res_out ? 1 : 0)));
    prof_end(st, (a.res_out ? name_res : name).c_str(), 2.0 * rows * 256.0 * taps * a.Cin, bytes);
  }
  JV_HIP(hipGetLastError());
  return JV_OK;
}
}  // namespace

// causal k = 3 convolution to 256 channels with the row-wise tail in its epilogue (rowconv_kernel.h)
int rowconv(const RowConvArgs& a, hipStream_t st) {
  if (a.M <= 0) return JV_OK;
  if (const char* ab = tuning_env("JV_RG_ABLATE")) const_cast<RowConvArgs&>(a).ablate = atoi(ab);
  if (!a.A || !a.W2 || !a.colscale || !a.amax_in || !a.row_slot || !a.out)
    return fail(JV_ERR_ARG, "rowconv: needs A, W planes, colscale, the measured bound with its row slots, and an output");
  if ((a.Cin & 31) || a.Cin < 32 || (a.lda & 3) || (a.ldw & 7) || (a.ldo & 3) || (a.res && (a.ldr & 3)) || (a.rowvec && (a.rowvec_ld & 3)))
    return fail(JV_ERR_ARG, "rowconv: Cin % 32 == 0 and aligned strides required");
  if (a.ln && (!a.ln_g || !a.ln_b)) return fail(JV_ERR_ARG, "rowconv: LayerNorm needs gain and offset");
  if (a.res_out && !a.res_cs) return fail(JV_ERR_ARG, "rowconv: the folded res_conv needs its column scales");
  if (a.ln2_out && (!a.ln2_g || !a.ln2_b || !(a.ln2_scale > 0.f) || a.ln2_plane <= 0 || a.ldo != 256))
    return fail(JV_ERR_ARG, "rowconv: the following LayerNorm needs gain, offset, a scale, a plane stride and 256-wide output rows");
  int rt = rowgemm_tile(a.M);
  if (rt == 0) rt = 2;
  switch (rt) {
    case 1:
    case 2: return rc_launch<2>(a, st);
    case 3: return rc_launch<3>(a, st);
    case 4: return rc_launch<4>(a, st);
    case 5: return rc_launch<5>(a, st);
    default: return fail(JV_ERR_ARG, "rowconv: bad tile height");
  }
}

namespace {
template <int RT, bool QKV>
int rr_launch(const RowResArgs& a, hipStream_t st) {
  static bool raised[64] = {};
  int dev = 0;
  JV_HIP(hipGetDevice(&dev));
  if (!raised[dev & 63]) {
    JV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&rowres_kernel<RT, QKV>), hipFuncAttributeMaxDynamicSharedMemorySize, rr_lds_bytes<RT>()));
    raised[dev & 63] = true;
  }
  const bool prof = prof_on();
  if (prof) prof_begin(st);
  hipLaunchKernelGGL((rowres_kernel<RT, QKV>), dim3(cdiv(a.M, 16 * RT - 2)), dim3(512), rr_lds_bytes<RT>(), st, a);
  if (prof) {
    static const std::string name = std::string("rowres_h3<") + std::to_string(16 * RT) + "x256" + (QKV ? ",qkv>" : ">");
    const double rows = (double)(a.alg_rows > 0 ? a.alg_rows : a.M);
    // block1's three taps + res_conv, block2's three taps (+ the following block's to_q | to_k | to_v)
    const double macs = 256.0 * (4.0 * a.Cin + 3.0 * 256) + (QKV ? 256.0 * 1536 : 0.0);
    // x in, out rows, the weights (h2 and res never leave the chip) (+ q rows and the k / v planes out)
    const double bytes = 4.0 * (rows * (a.Cin + 256 + (QKV ? 1536 : 0)) + macs);
    prof_end(st, name.c_str(), 2.0 * rows * macs, bytes);
  }
  JV_HIP(hipGetLastError());
  return JV_OK;
}
}  // namespace

// does the whole-resnet launch keep the number of workgroup rounds of the two launches it replaces?  Its workgroups produce
// 16 rt - 2 rows each (block2's causal halo), so a batch that just fills a round with 16 rt-row tiles may spill into the next
bool rowres_fits(int M) {
  const int rt = rowgemm_tile(M);
  if (rt < 2) return false;
  return cdiv(cdiv(M, 16 * rt - 2), 256) == cdiv(cdiv(M, 16 * rt), 256);
}

// a whole CausalResnetBlock1D in one launch (rowres_kernel.h)
int rowres(const RowResArgs& a, hipStream_t st) {
  if (a.M <= 0) return JV_OK;
  if (!a.A || !a.Wf1 || !a.Wf2 || !a.cs1 || !a.cs2 || !a.csr || !a.amax_in || !a.out || !a.temb || !a.ln1_g || !a.ln1_b || !a.ln2_g || !a.ln2_b)
    return fail(JV_ERR_ARG, "rowres: needs x, both fragment streams, the three column scales, the measured bound, both LayerNorms, the time embedding and an output");
  if ((a.Cin & 63) || a.Cin < 64 || (a.lda & 3) || (a.ldo & 3) || !(a.h2_bound > 0.f)) return fail(JV_ERR_ARG, "rowres: Cin % 64 == 0, aligned strides and the h2 bound required");
  if (a.slot_S < 0 && !a.row_slot) return fail(JV_ERR_ARG, "rowres: the compact geometry needs the row -> utterance table");
  if (a.lnf_out && (!a.lnf_g || !a.lnf_b || !(a.lnf_scale > 0.f) || a.lnf_plane <= 0 || a.ldo != 256))
    return fail(JV_ERR_ARG, "rowres: the following LayerNorm needs gain, offset, a scale, a plane stride and 256-wide output rows");
  const bool qkv = a.Wqf != nullptr;
  if (qkv && (!a.csq || !a.q || !a.kv2 || a.kv2_plane <= 0 || !(a.k_scale > 0.f) || !(a.v_scale > 0.f) || !a.lnf_g || !a.lnf_b || !(a.lnf_scale > 0.f) ||
              a.wqf_plane <= 0 || a.ldo != 256))
    return fail(JV_ERR_ARG, "rowres: q | k | v needs its fragments, column scales, outputs, the attention scales and the following LayerNorm");
  const int rt = rowgemm_tile(a.M);
  // the kernel reads whole window rows up to the last tile's end: they must exist in the input buffer or read as masked (a_rows clamps)
  switch (rt) {
    case 2: return qkv ? rr_launch<2, true>(a, st) : rr_launch<2, false>(a, st);
    case 3: return qkv ? rr_launch<3, true>(a, st) : rr_launch<3, false>(a, st);
    case 4: return qkv ? rr_launch<4, true>(a, st) : rr_launch<4, false>(a, st);
    case 5: return qkv ? rr_launch<5, true>(a, st) : rr_launch<5, false>(a, st);
    default: return fail(JV_ERR_ARG, "rowres: no row-owning tile height for this row count");
  }
}

namespace {
template <int RT>
int rf_launch(const RowFfnArgs& a, hipStream_t st) {
  static bool raised[64] = {};
  int dev = 0;
  JV_HIP(hipGetDevice(&dev));
  if (!raised[dev & 63]) {
    JV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&rowffn_kernel<RT>), hipFuncAttributeMaxDynamicSharedMemorySize, rgf_lds_bytes<RT>()));
    raised[dev & 63] = true;
  }
  const bool prof = prof_on();
  if (prof) prof_begin(st);
  hipLaunchKernelGGL((rowffn_kernel<RT>), dim3(cdiv(a.M, 16 * RT)), dim3(512), rgf_lds_bytes<RT>(), st, a);
  if (prof) {
    static const std::string name = std::string("rowffn_h3<") + std::to_string(16 * RT) + "x256" + ">";
    const double rows = (double)(a.alg_rows > 0 ? a.alg_rows : a.M);
    // algorithmic bytes: LayerNorm planes in (rows x 256 x 4 B), both weight matrices, residual in, rows out (+ planes out)
    prof_end(st, name.c_str(), 2.0 * rows * 2.0 * 256.0 * 1024.0, 4.0 * (rows * 256 * (a.ln ? 4 : 3) + 2.0 * 256 * 1024));
  }
  JV_HIP(hipGetLastError());
  return JV_OK;
}
}  // namespace

// out = res + ff.net.2(gelu(ff.net.0(x))) (+ LayerNorm planes of it) in one launch (rowffn_kernel); 256 -> 1024 -> 256 only
int rowffn(const RowFfnArgs& a, hipStream_t st) {
  if (a.M <= 0) return JV_OK;
  if (!a.A2 || !a.W1f || !a.W2f || !a.cs1 || !a.cs2 || !(a.a_scale1 > 0.f) || !(a.h_scale > 0.f) || !a.out || !a.res)
    return fail(JV_ERR_ARG, "rowffn: needs the operand planes, both weight matrices in fragment order, their scales, out and res");
  if ((a.lda2 & 7) || (a.ldo & 3) || (a.ldr & 3) || (a.ln && (!a.out2 || (a.ldo2 & 3) || !a.ln_g || !a.ln_b)))
    return fail(JV_ERR_ARG, "rowffn: aligned strides (and gain / offset / plane buffer for the LayerNorm epilogue) required");
  int rt = rowgemm_tile(a.M);
  if (rt == 0) rt = 2;
  switch (rt) {
    case 1:
    case 2: return rf_launch<2>(a, st);
    case 3: return rf_launch<3>(a, st);
    case 4: return rf_launch<4>(a, st);
    case 5: return rf_launch<5>(a, st);
    default: return fail(JV_ERR_ARG, "rowffn: bad tile height");
  }
}

int rowgemm(const RowGemmArgs& a, int epi, hipStream_t st) {
  if (a.M <= 0) return JV_OK;
  if (const char* ab = tuning_env("JV_RG_ABLATE")) const_cast<RowGemmArgs&>(a).ablate = atoi(ab);
  if (!a.A2 || !a.W2 || !a.colscale || !(a.a_scale > 0.f)) return fail(JV_ERR_ARG, "rowgemm: needs A and W planes, colscale and a_scale");
  if ((a.K & 31) || a.K < 32 || (a.N & 255) || a.N < 256) return fail(JV_ERR_ARG, "rowgemm: K % 32 == 0 and N % 256 == 0 required");
  if ((a.lda2 & 7) || (a.ldw & 7)) return fail(JV_ERR_ARG, "rowgemm: lda2 / ldw must be multiples of 8 halves");
  if ((epi == RG_RES || epi == RG_RES_LN) && (a.N != 256 || !a.res || !a.out || (a.ldo & 3) || (a.ldr & 3)))
    return fail(JV_ERR_ARG, "rowgemm: the residual epilogues own whole 256-wide rows");
  if (epi == RG_PLAIN && (!a.out || (a.ldo & 3))) return fail(JV_ERR_ARG, "rowgemm: bad output");
  if (epi == RG_QKV && (a.N != 1536 || !a.out || !a.out2 || (a.ldo & 3) || (a.ldo2 & 3)))
    return fail(JV_ERR_ARG, "rowgemm: the qkv epilogue wants N = 1536, a q buffer and a K/V plane buffer");
  if ((epi == RG_GELU_PL || epi == RG_RES_LN) && (!a.out2 || (a.ldo2 & 3))) return fail(JV_ERR_ARG, "rowgemm: bad plane output");
  if (epi == RG_RES_LN && (!a.ln_g || !a.ln_b)) return fail(JV_ERR_ARG, "rowgemm: LayerNorm epilogue needs gain and offset");
  int rt = (a.rt >= 2 && a.rt <= 5) ? a.rt : rowgemm_tile(a.M);
  if (rt == 0) rt = 2;      // callers ask rowgemm_tile() first; a direct call still works
  switch (rt) {
    case 1:
    case 2: return rg_launch1<2>(a, epi, st);
    case 3: return rg_launch1<3>(a, epi, st);
    case 4: return rg_launch1<4>(a, epi, st);
    case 5: return rg_launch1<5>(a, epi, st);
    default: return fail(JV_ERR_ARG, "rowgemm: bad tile height");
  }
}

}  // namespace jv
