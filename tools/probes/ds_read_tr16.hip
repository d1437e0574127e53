#include <hip/hip_runtime.h>
typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(const _Float16* in, _Float16* out) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x;
  // 16-lane group: lane 4q+p supplies row q, cols 4p..4p+3 of a [4 rows][16 cols] block, row stride 64 halves
  const int q = (lane & 15) >> 2, p = lane & 3, grp = lane >> 4;
  auto ptr = (__attribute__((address_space(3))) h4*)(lds + q * 64 + grp * 16 + 4 * p);
  h4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(ptr);
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
  _Float16 *in, *out; hipMalloc(&in, 8192); hipMalloc(&out, 512);
  _Float16 h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (_Float16)((i / 64) * 100 + (i % 64));
  hipMemcpy(in, h, 8192, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, in, out);
  _Float16 o[256]; hipMemcpy(o, out, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 1) { printf("lane %2d:", l); for (int e = 0; e < 4; ++e) printf(" %5.0f", (float)o[l*4+e]); printf("\n"); }
}
