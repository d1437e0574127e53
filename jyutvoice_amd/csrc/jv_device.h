// Device-side helpers shared by the kernels.
#pragma once
#include <math.h>

#include "jv_common.h"

namespace jv {

// activations as the reference's PyTorch ops compute them (exact erf GELU, mish = x tanh(log1p(exp x)), ...)
__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case ACT_RELU: return fmaxf(v, 0.f);
    case ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    case ACT_MISH: return v * tanhf(log1pf(expf(v)));
    case ACT_ELU: return v > 0.f ? v : expm1f(v);
    case ACT_SILU: return v / (1.f + expf(-v));
    default: return v;
  }
}

}  // namespace jv
