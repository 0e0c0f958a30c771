// Micro-benchmark: how many cycles does a SIMD-32 of gfx950 need per wave64 VALU instruction, as a function of the number of waves
// resident on that SIMD?  Settles the "2 or 4 cycles" question behind DESIGN.md section 3's "VALU-issue bound" reading of the eval
// kernel (round-4 review, item 1c): MI355X_MICROARCH.md says a wave64 VALU instruction issues over 2 cycles on the SIMD-32, this
// repository's round-1 ubench measured 4.7 cycles per INDEPENDENT v_fma_f32 from ONE wave.
//
// One work-group per CU (96 KB of LDS keeps a second one out), 256 * W threads = W waves per SIMD (W = 1 .. 4; waves of a work-group go
// round-robin over the four SIMDs).  Every wave runs ITERS iterations of a block of 64 instructions of one kind on 16 independent
// register chains (so a chain's next instruction is 16 issue slots behind its producer: no dependency stall), bracketed by s_memtime.
// Printed per kind and W: cycles per wave-instruction as ONE wave sees them (its own cadence), and cycles per instruction per SIMD
// (= wave cycles / instructions / W: the pipe's throughput).  If the pipe needs 2 cycles per instruction, the second column bottoms out
// at 2 once W >= 2; if it needs 4, it stays at 4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Kind { FMA = 0, MUL, PKFMA, PKMUL, EXP, RCP, DPP_SHR, DPP_BCAST, CNDMASK, CVT_PK_F16, FMA_DEP, MIX_FMA_EXP, N_KINDS };
static const char* kNames[N_KINDS] = {"v_fma_f32", "v_mul_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_exp_f32", "v_rcp_f32",
                                      "v_add_f32 dpp row_shr:1", "v_add_f32 dpp row_bcast:15", "v_cndmask_b32", "v_cvt_pk_f16_f32",
                                      "v_fma_f32 (ONE dependent chain)", "3 v_fma_f32 : 1 v_exp_f32"};

__device__ __forceinline__ uint64_t memtime() {
  uint64_t t;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, uint64_t* cycles, int iters, float seed) {
  extern __shared__ float pad[];
  const int lane = threadIdx.x & 63;
  float v[16];
  f32x2 p[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = seed + lane * 0.001f + i, p[i] = f32x2{v[i], v[i] + 0.5f};
  const float c = 1.0001f * seed;
  const f32x2 cc = {c, c};
  if (threadIdx.x == 0) pad[0] = seed;   // keeps the LDS allocation alive
  __syncthreads();
  const uint64_t t0 = memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if constexpr (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j]) : "v"(c));
        if constexpr (KIND == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[j]) : "v"(c));
        if constexpr (KIND == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[j]) : "v"(cc));
        if constexpr (KIND == PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[j]) : "v"(cc));
        if constexpr (KIND == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
        if constexpr (KIND == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[j]));
        if constexpr (KIND == DPP_SHR) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[j]));
        if constexpr (KIND == DPP_BCAST) asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v[j]));
        if constexpr (KIND == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[j]) : "v"(c));
        if constexpr (KIND == CVT_PK_F16) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[j]) : "v"(c));
        if constexpr (KIND == FMA_DEP) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[0]) : "v"(c));
        if constexpr (KIND == MIX_FMA_EXP) {
          if ((j & 3) == 3)
            asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
          else
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j]) : "v"(c));
        }
      }
    }
  }
  const uint64_t t1 = memtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i] + p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
void run(float* out, uint64_t* cyc_d, int n_cu, double* wave_cpi, double* simd_cpi, double* ms_out, int W) {
  const int iters = 4000, threads = 256 * W, n_waves = n_cu * 4 * W;
  const size_t lds = 96 * 1024;
  hipFuncSetAttribute((const void*)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  k<KIND><<<n_cu, threads, lds>>>(out, cyc_d, 16, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<KIND><<<n_cu, threads, lds>>>(out, cyc_d, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  static uint64_t host[256 * 16];
  hipMemcpy(host, cyc_d, n_waves * sizeof(uint64_t), hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < n_waves; ++i) sum += (double)host[i];
  const double insts = (double)iters * 64;
  *wave_cpi = sum / n_waves / insts;
  *simd_cpi = *wave_cpi / W;
  *ms_out = ms;
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount;
  float* out;
  uint64_t* cyc;
  hipMalloc(&out, (size_t)n_cu * 1024 * 4);
  hipMalloc(&cyc, (size_t)n_cu * 16 * 8);
  printf("# %s, %d CUs, clock %d MHz (s_memtime ticks; the wall-clock column says what a tick is)\n", prop.gcnArchName, n_cu, prop.clockRate / 1000);
  printf("# one work-group of 256*W threads per CU = W waves per SIMD; 4000 x 64 instructions per wave on 16 independent chains\n");
  printf("%-34s %2s %12s %12s %10s %14s\n", "instruction", "W", "cyc/inst/wave", "cyc/inst/SIMD", "ms", "ticks per ns");
  for (int kind = 0; kind < N_KINDS; ++kind)
    for (int W = 1; W <= 4; ++W) {
      double w, s, ms;
      switch (kind) {
#define CASE(K_) case K_: run<K_>(out, cyc, n_cu, &w, &s, &ms, W); break;
        CASE(FMA) CASE(MUL) CASE(PKFMA) CASE(PKMUL) CASE(EXP) CASE(RCP) CASE(DPP_SHR) CASE(DPP_BCAST) CASE(CNDMASK) CASE(CVT_PK_F16) CASE(FMA_DEP)
        CASE(MIX_FMA_EXP)
#undef CASE
      }
      printf("%-34s %2d %12.2f %12.2f %10.3f %14.3f\n", kNames[kind], W, w, s, ms, w * 4000 * 64 / (ms * 1e6));
    }
  return 0;
}
