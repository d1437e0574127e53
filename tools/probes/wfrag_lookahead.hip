// Probe: how should a row-owning workgroup's waves request their weight fragments?  The loops of rowblock_kernel.h /
// rowres_kernel.h keep TWO register buffers per wave (a step's buffer is refilled in two halves, mid-step and at its end:
// 1.5 steps of lookahead), and DESIGN.md 5 (Round 4) reads the 930 cycles a lone wave needs per step as that lookahead being
// shorter than an L2 round trip.  Four buffers inside rowres_kernel were SLOWER (profiles/r04_deep_weight_buffer_ab.json).
// This probe isolates the loop: 256 workgroups x 8 waves (one workgroup per CU: the LDS allocation forces it), every wave
// owns 2 of 16 column blocks x RT = 5 row tiles, A fragments from an LDS image (a whole step's, double-buffered across steps,
// as the kernels do), weight fragments from a 1.5 MB fragment-order matrix in L2 (48 steps x 16 blocks x 1 KB x 2 planes,
// walked REP times), 30 v_mfma_f32_16x16x32_f16 per step and wave.  Swept one at a time:
//   NB     buffers per wave (2, 3, 4)
//   SPLIT  1: a buffer's first half is requested right behind the first column block's MFMAs, the second half at the end
//             of the step (the kernels' pattern); 0: all four loads at the end of the step
//   LONE   1: waves 4..7 leave at once -- one wave per SIMD; 2: waves 4..7 run independent v_fma_f32 with vector-register
//             operands for as long as the MFMA waves run (the staggered schedule's situation: the other half in its GELU pass);
//             3: the same with the fma's operands in SGPRs; 4 / 5: 2 + two / eight dword LDS stores per 16 fmas; 6: 2 + two v_exp_f32
// Output: median over all waves of (s_memtime at loop end - at loop start) / steps, and the kernel's time by HIP events.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/wfrag_lookahead.hip -o /tmp/wfrag_lookahead && /tmp/wfrag_lookahead
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <utility>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int RT = 5, KS = 48, NBLK = 16, REP = 8, STEPS = KS * REP;
constexpr int STAGE = 2 * 16 * RT * 64;      // one 32-deep k-step of 80 rows as two fp16 planes
constexpr int LDS_BYTES = 150 * 1024;       // > half of the CU's 160 KB: one workgroup per CU

template <class F, int... Is>
__device__ __forceinline__ void for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void load_frag(u32x4& dst, const unsigned short* ptr) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(dst) : "v"(ptr) : "memory");
}
__device__ __forceinline__ void landed(u32x4& a, u32x4& b, u32x4& c, u32x4& d) { asm volatile("" : "+v"(a), "+v"(b)::"memory"); asm volatile("" : "+v"(c), "+v"(d)::"memory"); }

template <int NB, int SPLIT>
__global__ __launch_bounds__(512, 2) void probe(const unsigned short* __restrict__ W, long plane, unsigned long long* __restrict__ t,
                                                float* __restrict__ sink, int lone) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
  for (int i = tid; i < 8 * STAGE / 4; i += 512) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u + (i & 255);      // halves near 1.0
  __syncthreads();
  if (lone == 1 && wave >= 4) return;
  if (lone >= 2 && wave >= 4) {
    // the SIMD's other wave does vector work for as long as the MFMA wave runs (the staggered schedule's GELU pass):
    // lone == 2: sixteen independent fp32 fmas per iteration (two register operands + one accumulator each);
    // lone == 3: the same from four operands held in SGPRs (no vector-register reads besides the accumulator)
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = lane * 1e-3f + i;
    float c0 = 1.0001f, c1 = 0.5f;
    asm volatile("" : "+v"(c0), "+v"(c1));
    float k0 = 1.0001f, k1 = 0.5f;
    asm volatile("" : "+s"(k0), "+s"(k1));
#pragma unroll 1
    for (int it = 0; it < 5200; ++it) {
      if (lone == 3) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "s"(k0));
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c0), "v"(c1));
      }
      // lone == 4: + two dword LDS stores per 16 fmas (the GELU pass's ratio: the hidden tile's planes leave as dwords);
      // lone == 5: + eight; lone == 6: + two v_exp_f32 (quarter rate)
      if (lone == 4 || lone == 5) {
        unsigned* const dst = reinterpret_cast<unsigned*>(lds + 8 * STAGE) + (wave - 4) * 4096 + ((it & 15) * 256) + lane;
        const int n = lone == 4 ? 2 : 8;
        for (int i = 0; i < n; ++i) dst[i * 64] = __float_as_uint(x[i]);
      }
      if (lone == 6) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(x[0]));
        asm volatile("v_exp_f32 %0, %0" : "+v"(x[1]));
      }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += x[i];
    if (s == 12345.678f) sink[1] = s;
    return;
  }
  const unsigned short* wb[2][2];
  for (int nt = 0; nt < 2; ++nt)
    for (int pl = 0; pl < 2; ++pl) wb[nt][pl] = W + (long)pl * plane + (long)(wave * 2 + nt) * 512 + lane * 8;
  long woff = 0;      // the NEXT step to request
  int wk = 0;
  auto advance = [&]() {
    if (++wk == KS) { wk = 0; woff = 0; } else woff += (long)NBLK * 512;
  };
  u32x4 bq[NB][2][2];
  for (int b = 0; b < NB; ++b)
    for (int i = 0; i < 4; ++i) bq[b][i >> 1][i & 1] = u32x4{0u, 0u, 0u, 0u};
  f32x4 acc[RT][2];
  for (int mt = 0; mt < RT; ++mt)
    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the first NB steps' fragments
  static_for<NB>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    load_frag(bq[b][0][0], wb[0][0] + woff); load_frag(bq[b][0][1], wb[0][1] + woff);
    load_frag(bq[b][1][0], wb[1][0] + woff); load_frag(bq[b][1][1], wb[1][1] + woff);
    advance();
  });
  wait_vmcnt<0>();
  static_for<NB>([&](auto bc) { constexpr int b = decltype(bc)::value; landed(bq[b][0][0], bq[b][0][1], bq[b][1][0], bq[b][1][1]); });
  u32x4 af[2][RT][2];
  const int a_off = r16 * 64 + ((kq ^ (((r16 >> 2) & 1) << 1)) << 4);
  auto read_a = [&](auto par_tag, const int ks) {
    constexpr int par = decltype(par_tag)::value;
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) af[par][mt][pl] = *reinterpret_cast<const u32x4*>(lds + (ks & 7) * STAGE + a_off + pl * (16 * RT * 64) + mt * 1024);
  };
  read_a(std::integral_constant<int, 0>{}, 0);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  constexpr int UNR = NB % 2 ? 2 * NB : NB;      // buffer = step mod NB and A parity = step mod 2: both compile-time
#pragma unroll 1
  for (int s0 = 0; s0 < STEPS; s0 += UNR) {
    static_for<UNR>([&](auto uc) {
      constexpr int u = decltype(uc)::value, b = u % NB, par = u & 1;
      // this step's buffer: everything requested after it may still be in flight (SPLIT: its second half too, until mid-step)
      if constexpr (SPLIT) {
        wait_vmcnt<4 * (NB - 1) + 2>();
        landed(bq[b][0][0], bq[b][0][1], bq[b][0][0], bq[b][0][1]);
      } else {
        wait_vmcnt<4 * (NB - 1)>();
        landed(bq[b][0][0], bq[b][0][1], bq[b][1][0], bq[b][1][1]);
      }
      auto block = [&](auto ntc) {
        constexpr int nt = decltype(ntc)::value;
#pragma unroll
        for (int mt = 0; mt < RT; ++mt) {
          f32x4 tt = acc[mt][nt];
          auto mm = [&](const u32x4& x, const u32x4& y) {
            tt = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), tt, 0, 0, 0);
          };
          mm(af[par][mt][1], bq[b][nt][0]);
          mm(af[par][mt][0], bq[b][nt][1]);
          mm(af[par][mt][0], bq[b][nt][0]);
          acc[mt][nt] = tt;
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      block(std::integral_constant<int, 0>{});
      if constexpr (SPLIT) {
        load_frag(bq[b][0][0], wb[0][0] + woff); load_frag(bq[b][0][1], wb[0][1] + woff);
        wait_vmcnt<4 * (NB - 1) + 2>();      // the second half: behind it, the other buffers' four loads each and the two just issued
        landed(bq[b][1][0], bq[b][1][1], bq[b][1][0], bq[b][1][1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      read_a(std::integral_constant<int, par ^ 1>{}, s0 + u + 1);
      __builtin_amdgcn_sched_barrier(0);
      block(std::integral_constant<int, 1>{});
      if constexpr (!SPLIT) { load_frag(bq[b][0][0], wb[0][0] + woff); load_frag(bq[b][0][1], wb[0][1] + woff); }
      load_frag(bq[b][1][0], wb[1][0] + woff); load_frag(bq[b][1][1], wb[1][1] + woff);
      advance();
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  wait_vmcnt<0>();
  static_for<NB>([&](auto bc) { constexpr int b = decltype(bc)::value; landed(bq[b][0][0], bq[b][0][1], bq[b][1][0], bq[b][1][1]); });
  if (lane == 0) t[blockIdx.x * 8 + wave] = t1 - t0;
  float s = 0.f;
  for (int mt = 0; mt < RT; ++mt)
    for (int nt = 0; nt < 2; ++nt) s += acc[mt][nt][0] + acc[mt][nt][3];
  if (s == 12345.678f) sink[0] = s;
}

template <int NB, int SPLIT>
static void run(const unsigned short* W, long plane, unsigned long long* t, float* sink, int lone) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<NB, SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  std::vector<unsigned long long> h(256 * 8);
  double med = 0;
  for (int it = 0; it < 4; ++it) {
    hipMemset(t, 0, sizeof(unsigned long long) * 256 * 8);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<NB, SPLIT>), dim3(256), dim3(512), LDS_BYTES, 0, W, plane, t, sink, lone);
    hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) { printf("launch failed\n"); exit(1); }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (it == 0) continue;      // warm-up: weights into L2
    best = std::min(best, ms);
    hipMemcpy(h.data(), t, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    std::vector<double> v;
    for (auto x : h) if (x) v.push_back((double)x / STEPS);
    std::sort(v.begin(), v.end());
    med = v[v.size() / 2];
  }
  printf("NB=%d split=%d lone=%d : %7.1f s_memtime ticks per step (median over waves), kernel %8.1f us = %6.1f ns per step\n", NB, SPLIT, lone, med,
         1e3 * best, 1e6 * best / STEPS);
}

int main() {
  const long plane = (long)KS * NBLK * 512;      // halves
  unsigned short* W;
  unsigned long long* t;
  float* sink;
  hipMalloc(&W, sizeof(unsigned short) * 2 * plane);
  hipMalloc(&t, sizeof(unsigned long long) * 256 * 8);
  hipMalloc(&sink, 8);
  std::vector<unsigned short> hw(2 * plane);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0x3800 + (i % 97);      // halves near 0.5
  hipMemcpy(W, hw.data(), sizeof(unsigned short) * hw.size(), hipMemcpyHostToDevice);
  for (int lone = 0; lone < 2; ++lone) {
    run<2, 1>(W, plane, t, sink, lone);
    run<2, 0>(W, plane, t, sink, lone);
    run<3, 1>(W, plane, t, sink, lone);
    run<3, 0>(W, plane, t, sink, lone);
    run<4, 1>(W, plane, t, sink, lone);
    run<4, 0>(W, plane, t, sink, lone);
  }
  for (int lone = 2; lone < 7; ++lone) {
    run<2, 1>(W, plane, t, sink, lone);
    run<4, 0>(W, plane, t, sink, lone);
  }
  printf("(30 MFMAs of 16 cycles per step and wave: 480 cycles of issue; two waves per SIMD: 960 per step pair)\n");
  return 0;
}
