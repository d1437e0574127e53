// Probe: cycles per v_mfma_f32_32x32x16_f16 with vector instructions issued in the gaps, one wave on a SIMD, by operand file
// (where the accumulator, the A and the B operand live) and by filler kind.  attention_s.hip is built on the answer.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_gap.hip -o /tmp/mfma_gap && /tmp/mfma_gap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int K>
__device__ __forceinline__ void mfma(f32x16& c, const u32x4& a, const u32x4& b) {
  if constexpr (K == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[64:67], %1, %0" : "+v"(c) : "v"(b));             // C vgpr, A agpr, B vgpr
  else if constexpr (K == 1) asm volatile("v_mfma_f32_32x32x16_f16 a[0:15], a[64:67], %0, a[0:15]" ::"v"(b));        // C agpr, A agpr, B vgpr
  else if constexpr (K == 2) asm volatile("v_mfma_f32_32x32x16_f16 a[0:15], %0, %1, a[0:15]" ::"v"(a), "v"(b));      // C agpr, A vgpr, B vgpr
  else if constexpr (K == 3) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[64:67], %0" : "+v"(c) : "v"(a));        // C vgpr, A vgpr, B agpr
  else if constexpr (K == 4) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));      // all vgpr
  else asm volatile("v_mfma_f32_32x32x16_f16 a[0:15], a[64:67], a[68:71], a[0:15]");                                // all agpr
}
template <int F>
__device__ __forceinline__ void filler(float (&x)[8], const float one) {
  if constexpr (F == 1) {
    asm volatile("v_add_f32 %0, %0, %5\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %5\n v_add_f32 %3, %3, %5\n v_add_f32 %4, %4, %5"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]) : "v"(one));
  } else if constexpr (F == 2) {      // one fp16 split of a pair + two adds (attention_s.hip's split tick)
    asm volatile("v_cvt_pk_f16_f32 %2, %0, %1\n v_fma_mix_f32 %3, %0, %6, -%2 op_sel_hi:[0,0,1]\n"
                 " v_fma_mix_f32 %4, %1, %6, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n v_cvt_pk_f16_f32 %5, %3, %4\n"
                 " v_add_f32 %3, %0, %1\n v_add_f32 %4, %4, %3"
                 : "+v"(x[0]), "+v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]) : "v"(one));
  } else if constexpr (F == 3) {      // two exps (attention_s.hip's exp tick)
    asm volatile("v_fma_f32 %0, %2, %4, %4\n v_fma_f32 %1, %3, %4, %4\n v_exp_f32 %0, %0\n v_exp_f32 %1, %1"
                 : "=&v"(x[0]), "=&v"(x[1]) : "v"(x[2]), "v"(x[3]), "v"(one));
  } else if constexpr (F == 4) {
    asm volatile("v_max3_f32 %0, %1, %2, %3\n v_max3_f32 %1, %2, %3, %4\n v_max3_f32 %2, %3, %4, %0\n v_max3_f32 %3, %4, %0, %1"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "v"(x[4]));
  } else if constexpr (F == 5) {      // the split alone (4 instructions)
    asm volatile("v_cvt_pk_f16_f32 %2, %0, %1\n v_fma_mix_f32 %3, %0, %6, -%2 op_sel_hi:[0,0,1]\n"
                 " v_fma_mix_f32 %4, %1, %6, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n v_cvt_pk_f16_f32 %5, %3, %4"
                 : "+v"(x[0]), "+v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]) : "v"(one));
  } else if constexpr (F == 6) {      // one exp + fma + add
    asm volatile("v_fma_f32 %0, %2, %3, %3\n v_exp_f32 %0, %0\n v_add_f32 %1, %1, %0" : "=&v"(x[0]), "+v"(x[1]) : "v"(x[2]), "v"(one));
  }
}
template <int K, int F>
__global__ __launch_bounds__(64, 1) void probe(unsigned long long* out, float one) {
  asm volatile("" ::: "a255");
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  u32x4 a = {threadIdx.x, 1u, 2u, 3u}, b = {4u, 5u, 6u, threadIdx.x};
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = one * i;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      mfma<K>(c, a, b);
      filler<F>(x, one);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_nop 15\n s_nop 15");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)s; }
}
template <int K, int F>
void run(unsigned long long* d) {
  hipLaunchKernelGGL((probe<K, F>), dim3(1), dim3(64), 0, 0, d, 1.0f);
  hipLaunchKernelGGL((probe<K, F>), dim3(1), dim3(64), 0, 0, d, 1.0f);
  unsigned long long h[2];
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("%6.1f ", (double)h[0] / 2048.0);
}
template <int K>
void row(unsigned long long* d, const char* name) {
  printf("%-28s", name);
  run<K, 0>(d); run<K, 1>(d); run<K, 4>(d); run<K, 5>(d); run<K, 2>(d); run<K, 6>(d); run<K, 3>(d);
  printf("\n");
}
int main() {
  unsigned long long* d;
  hipMalloc(&d, 16);
  printf("cycles (s_memtime) per MFMA; fillers per gap:  none  5 add  4 max3 split split+2add fma+exp+add 2x(fma,exp)\n");
  row<0>(d, "C vgpr  A agpr  B vgpr");
  row<1>(d, "C agpr  A agpr  B vgpr");
  row<2>(d, "C agpr  A vgpr  B vgpr");
  row<3>(d, "C vgpr  A vgpr  B agpr");
  row<4>(d, "C vgpr  A vgpr  B vgpr");
  row<5>(d, "C agpr  A agpr  B agpr");
  return 0;
}
