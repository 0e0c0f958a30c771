// Micro-benchmark 2 (inline asm, exact instruction streams): cycles per {1 MFMA + N independent v_fma_f32} for one wave per SIMD,
// fp32-input MFMA (v_mfma_f32_32x32x2_f32, 64-cycle) and bf16 MFMA (v_mfma_f32_32x32x16_bf16, 32-cycle).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define FMA1(r) "v_fma_f32 " r ", " r ", %[c], %[c]\n"
template <int N>
__device__ __forceinline__ void fmas(float (&v)[16], float c) {
  if constexpr (N >= 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[0]) : "v"(c));
  if constexpr (N >= 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[1]) : "v"(c));
  if constexpr (N >= 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[2]) : "v"(c));
  if constexpr (N >= 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[3]) : "v"(c));
  if constexpr (N >= 5) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[4]) : "v"(c));
  if constexpr (N >= 6) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[5]) : "v"(c));
  if constexpr (N >= 7) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[6]) : "v"(c));
  if constexpr (N >= 8) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[7]) : "v"(c));
  if constexpr (N >= 9) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[8]) : "v"(c));
  if constexpr (N >= 10) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[9]) : "v"(c));
  if constexpr (N >= 11) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[10]) : "v"(c));
  if constexpr (N >= 12) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[11]) : "v"(c));
  if constexpr (N >= 13) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[12]) : "v"(c));
  if constexpr (N >= 14) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[13]) : "v"(c));
  if constexpr (N >= 15) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[14]) : "v"(c));
  if constexpr (N >= 16) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[15]) : "v"(c));
}

template <int KIND, int N>  // KIND 0: no MFMA, 1: fp32 MFMA, 2: bf16 MFMA
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int q = 0; q < 16; ++q) acc[i][q] = seed * q;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = seed + lane * 0.001f + i;
  float a = seed * lane, b = seed + 1.0f, c = 1.0001f * seed;
  bf16x8 ab, bb;
  for (int i = 0; i < 8; ++i) ab[i] = (short)(lane + i), bb[i] = (short)(lane * 3 + i);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if constexpr (KIND == 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
      if constexpr (KIND == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(ab), "v"(bb));
      fmas<N>(v, c);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND, int N>
void run(float* out, unsigned long long* cyc, int blocks, const char* name) {
  const int iters = 5000;
  k<KIND, N><<<blocks, 256>>>(out, 10, 1.0f, cyc);
  hipDeviceSynchronize();
  k<KIND, N><<<blocks, 256>>>(out, iters, 1.0f, cyc);
  hipDeviceSynchronize();
  printf("  %-6s N=%2d: %7.1f cycles per (MFMA + N fma)\n", name, N, (double)*cyc / (iters * 4.0));
}

int main() {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 4096 * 256 * 4);
  hipHostMalloc(&cyc, 8);
  for (int blocks : {256, 512}) {
    printf("blocks=%d (%d wave(s) per SIMD)\n", blocks, blocks / 256);
    run<0, 8>(out, cyc, blocks, "none"), run<0, 16>(out, cyc, blocks, "none");
    run<1, 0>(out, cyc, blocks, "fp32"), run<1, 4>(out, cyc, blocks, "fp32"), run<1, 8>(out, cyc, blocks, "fp32");
    run<1, 12>(out, cyc, blocks, "fp32"), run<1, 16>(out, cyc, blocks, "fp32");
    run<2, 0>(out, cyc, blocks, "bf16"), run<2, 2>(out, cyc, blocks, "bf16"), run<2, 4>(out, cyc, blocks, "bf16");
    run<2, 6>(out, cyc, blocks, "bf16"), run<2, 8>(out, cyc, blocks, "bf16"), run<2, 12>(out, cyc, blocks, "bf16");
    run<2, 16>(out, cyc, blocks, "bf16");
  }
  return 0;
}
