// Host side of the fused transformer-block launch (rowblock_kernel.h): argument checks, tile height, launch.
#include "rowblock_kernel.h"

namespace jv {

int rowgemm_tile(int M);      // rowgemm.hip

namespace {
template <int RT, bool QKV, bool STAG>
int rb_launch(const RowBlockArgs& a, hipStream_t st) {
  static bool raised[64] = {};
  int dev = 0;
  JV_HIP(hipGetDevice(&dev));
  if (!raised[dev & 63]) {
    JV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&rowblock_kernel<RT, QKV, STAG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               rb_lds_bytes<RT>()));
    raised[dev & 63] = true;
  }
  const bool prof = prof_on();
  if (prof) prof_begin(st);
  hipLaunchKernelGGL((rowblock_kernel<RT, QKV, STAG>), dim3(cdiv(a.M, 16 * RT)), dim3(512), rb_lds_bytes<RT>(), st, a);
  if (prof) {
    static const std::string name_plain = std::string("rowblock_h3<") + std::to_string(16 * RT) + "x256" + (QKV ? ",qkv>" : ">");
    static const std::string name_ln = std::string("rowblock_h3<") + std::to_string(16 * RT) + "x256,ln>";
    const std::string& name = (!QKV && a.ln_out) ? name_ln : name_plain;
    const double rows = (double)(a.alg_rows > 0 ? a.alg_rows : a.M);
    // algorithmic work: to_out (512 -> 256) + ff.net.0 (256 -> 1024) + ff.net.2 (1024 -> 256) (+ q|k|v: 256 -> 1536);
    // bytes: attention planes in (512 x 4 B per row), residual in, rows out (256 x 4 B each), the weights, (+ q|k|v out)
    const double macs = 256.0 * 512 + 2.0 * 256 * 1024 + (QKV ? 256.0 * 1536 : 0.0);
    const double bytes = 4.0 * (rows * (512 + 256 + 256 + (QKV ? 1536 : 0)) + macs);
    prof_end(st, name.c_str(), 2.0 * rows * macs, bytes);
  }
  JV_HIP(hipGetLastError());
  return JV_OK;
}

// JV_FF_STAGGER=1: the feed-forward with waves 0..3 half a hidden chunk ahead of waves 4..7 (rowblock_kernel.h, STAG; the same
// bits).  Measured SLOWER than every wave in the same phase (133.8 against 129.4 us per launch, same box): a wave that runs
// its MFMA steps alone on a SIMD takes ~930 cycles per step, not 480 -- its weight fragments are requested 1.5 steps ahead,
// which at one wave per SIMD is less than an L2 round trip -- so the half that computes beside the other half's GELU pass
// takes as long as both halves computing together.  Kept as a tested alternative.
bool ff_stagger() {
  static const bool on = getenv("JV_FF_STAGGER") != nullptr;
  const char* d = dyn_env("JV_FF_STAGGER");
  return on || (d && d[0] == '1');
}

template <int RT>
int rb_launch1(const RowBlockArgs& a, bool qkv, hipStream_t st) {
  if (ff_stagger()) return qkv ? rb_launch<RT, true, true>(a, st) : rb_launch<RT, false, true>(a, st);
  return qkv ? rb_launch<RT, true, false>(a, st) : rb_launch<RT, false, false>(a, st);
}
}  // namespace

// h += to_out(att); out = h + ff.net.2(gelu(ff.net.0(LayerNorm3(h)))); qkv: q | k | v = to_q/k/v_next(LayerNorm1_next(out))
int rowblock(const RowBlockArgs& a, bool qkv, hipStream_t st) {
  if (a.M <= 0) return JV_OK;
  if (!a.A2 || !a.Wof || !a.cso || !(a.a_scale_o > 0.f) || !a.h || !a.ln3_g || !a.ln3_b)
    return fail(JV_ERR_ARG, "rowblock: to_out needs the attention planes, fragment-order weights, scales, h and LayerNorm3");
  if (!a.W1f || !a.W2f || !a.cs1 || !a.cs2 || !(a.a_scale1 > 0.f) || !(a.h_scale > 0.f) || !a.out)
    return fail(JV_ERR_ARG, "rowblock: the feed-forward pair needs both weight matrices in fragment order, their scales and an output");
  if (a.ldo & 3) return fail(JV_ERR_ARG, "rowblock: aligned output stride required");
  if (qkv && (a.out != a.h || a.ldo != 256)) return fail(JV_ERR_ARG, "rowblock: the q|k|v phase follows a block that writes the trunk in place");
  if (qkv && (!a.Wqf || !a.csq || !(a.a_scale_q > 0.f) || !a.ln1_g || !a.ln1_b || !a.q || !a.kv2 || !(a.k_scale > 0.f) || !(a.v_scale > 0.f)))
    return fail(JV_ERR_ARG, "rowblock: the q|k|v phase needs LayerNorm1, fragment-order weights, scales, a q buffer and a K/V plane buffer");
  if (!qkv && a.ln_out && (a.out != a.h || a.ldo != 256 || !a.ln1_g || !a.ln1_b || !(a.a_scale_q > 0.f) || a.ln_out_plane <= 0))
    return fail(JV_ERR_ARG, "rowblock: LayerNorm1 planes to HBM need a block that writes the trunk in place, LayerNorm1 and its scale");
  int rt = rowgemm_tile(a.M);
  if (rt == 0) rt = 2;
  if (rt == 1) rt = 2;
  // the kernel reads whole tiles: rows up to the end of the last one must exist in every row buffer
  if ((long)cdiv(a.M, 16 * rt) * 16 * rt > a.a_rows) return fail(JV_ERR_ARG, "rowblock: the row buffers must hold whole tiles (a_rows)");
  RowBlockArgs b = a;      // the reciprocals of the (power-of-two) scales, for the kernel
  b.inv_a_scale_o = 1.0f / a.a_scale_o; b.inv_a_scale1 = 1.0f / a.a_scale1; b.inv_h_scale = 1.0f / a.h_scale;
  b.inv_a_scale_q = qkv ? 1.0f / a.a_scale_q : 0.f;
  if (const char* ab = tuning_env("JV_RB_ABLATE")) b.ablate = atoi(ab);
  switch (rt) {
    case 2: return rb_launch1<2>(b, qkv, st);
    case 3: return rb_launch1<3>(b, qkv, st);
    case 4: return rb_launch1<4>(b, qkv, st);
    case 5: return rb_launch1<5>(b, qkv, st);
    default: return fail(JV_ERR_ARG, "rowblock: bad tile height");
  }
}

}  // namespace jv
