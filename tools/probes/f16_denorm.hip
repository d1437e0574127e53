// Probe: does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs, and does the fp32 -> fp16 conversion produce them?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/f16_denorm.hip -o /tmp/f16_denorm && /tmp/f16_denorm
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(float* out, float tiny) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)tiny; b[i] = (_Float16)1024.0f; }
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; }
}
int main() {
  float* d; hipMalloc(&d, 8);
  for (float tiny : {9.5367431640625e-07f /*2^-20*/, 5.9604644775390625e-08f /*2^-24*/, 1e-6f}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, tiny);
    float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("tiny %.6e: cvt -> %.6e, mfma sum16(tiny*1024) = %.6e (expect %.6e)\n", tiny, h[1], h[0], 16.0 * 1024.0 * h[1]);
  }
  return 0;
}
