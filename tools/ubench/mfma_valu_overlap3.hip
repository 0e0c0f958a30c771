// Micro-benchmark 3: 512-thread work-groups, one per CU: waves 0-3 (one per SIMD) run ONLY MFMAs, waves 4-7 (their SIMD partners)
// run ONLY v_fma_f32.  Does the SIMD overlap the two waves?  time(both) ~ max -> separate pipes; ~ sum -> shared.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>  // 1 fp32 MFMA, 2 bf16 MFMA
__global__ __launch_bounds__(512) void k(float* out, int iters_m, int iters_v, float seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float s = 0;
  if (wave < 4) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
      for (int q = 0; q < 16; ++q) acc[i][q] = seed * q;
    float a = seed * lane, b = seed + 1.0f;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) ab[i] = (short)(lane + i), bb[i] = (short)(lane * 3 + i);
    for (int it = 0; it < iters_m; ++it) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if constexpr (KIND == 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
        if constexpr (KIND == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(ab), "v"(bb));
      }
    }
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
  } else {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed + lane * 0.001f + i;
    const float c = 1.0001f * seed;
    for (int it = 0; it < iters_v; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j]) : "v"(c));
    }
    for (int i = 0; i < 16; ++i) s += v[i];
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int KIND>
float run(float* out, int im, int iv) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  k<KIND><<<256, 512>>>(out, 10, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<KIND><<<256, 512>>>(out, im, iv, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  // iteration counts chosen so that each role alone takes about the same time
  const int im32 = 20000, iv32 = 64000;  // 4 x 64 cyc = 256 cyc/iter vs 16 fma ~ 80 cyc/iter
  printf("fp32 MFMA: mfma-only %.3f ms, valu-only %.3f ms, both %.3f ms\n", run<1>(out, im32, 0), run<1>(out, 0, iv32), run<1>(out, im32, iv32));
  const int im16 = 40000, iv16 = 64000;
  printf("bf16 MFMA: mfma-only %.3f ms, valu-only %.3f ms, both %.3f ms\n", run<2>(out, im16, 0), run<2>(out, 0, iv16), run<2>(out, im16, iv16));
  return 0;
}
