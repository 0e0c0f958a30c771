// Micro-benchmark: how many cycles does a SIMD-32 of gfx950 need per wave64 VALU instruction, as a function of the number of waves
// resident on that SIMD?  Settles the "2 or 4 cycles" question behind DESIGN.md section 3's "VALU-issue bound" reading of the eval
// kernel (round-4 review, item 1c): MI355X_MICROARCH.md says a wave64 VALU instruction issues over 2 cycles on the SIMD-32, this
// repository's round-1 ubench measured 4.7 cycles per INDEPENDENT v_fma_f32 from ONE wave.
//
// One work-group per CU (96 KB of LDS keeps a second one out), 256 * W threads = W waves per SIMD (W = 1 .. 4; waves of a work-group go
// round-robin over the four SIMDs).  Every wave runs ITERS iterations of a block of 64 instructions of one kind on 16 independent
// register chains (so a chain's next instruction is 16 issue slots behind its producer: no dependency stall), bracketed by s_memtime.
// Every wave also records the SIMD it ran on (HW_ID) and the constant 100 MHz counter (s_memrealtime).  Printed per kind and W: cycles per
// wave-instruction as ONE wave sees them (its own cadence); cycles per instruction per SIMD measured PER SIMD as (last wave's end - first
// wave's start) / instructions issued on that SIMD -- the pipe's throughput whatever the placement of the waves; the waves-per-SIMD
// histogram (is the placement what the launch intended?); the shader clock (s_memtime ticks per s_memrealtime tick x 100 MHz) and the
// wall-clock ns per instruction per SIMD.  If the pipe needs 2 cycles per instruction, the throughput column bottoms out near 2 once
// W >= 2; if it needs 4, it stays at 4.  (Version 1 of this file divided the mean wave time by W and is kept in profiles/r05a.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Kind { FMA = 0, MUL, PKFMA, PKMUL, EXP, RCP, DPP_SHR, DPP_BCAST, CNDMASK, CVT_PK_F16, FMA_DEP, MIX_FMA_EXP,
            ADD, FMAC, MUL_E64, FMA3, CNDMASK_SET, CNDMASK_E64, MAX, MOV, LSHL, CVT_F16, MED3, FMA_MIX, MUL_2W, N_KINDS };
static const char* kNames[N_KINDS] = {"v_fma_f32", "v_mul_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_exp_f32", "v_rcp_f32",
                                      "v_add_f32 dpp row_shr:1", "v_add_f32 dpp row_bcast:15", "v_cndmask_b32", "v_cvt_pk_f16_f32",
                                      "v_fma_f32 (ONE dependent chain)", "3 v_fma_f32 : 1 v_exp_f32",
                                      "v_add_f32 (e32)", "v_fmac_f32 (e32)", "v_mul_f32_e64 (VOP3 encoding)", "v_fma_f32, 3 distinct sources",
                                      "v_cndmask_b32 e32, vcc written first", "v_cndmask_b32_e64, SGPR-pair mask", "v_max_f32 (e32)",
                                      "v_mov_b32 (e32)", "v_lshlrev_b32 (e32)", "v_cvt_f16_f32 (e32)", "v_med3_f32 (VOP3)",
                                      "v_fma_mix_f32 (VOP3P)", "v_mul_f32 e32, 2 distinct sources"};

__device__ __forceinline__ uint64_t memtime() {
  uint64_t t;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
__device__ __forceinline__ uint64_t realtime() {
  uint64_t t;
  asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
__device__ __forceinline__ unsigned hw_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
  return v;
}
struct WaveRec {
  uint64_t t0, t1, r0, r1;
  unsigned hw, pad;
};

template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, WaveRec* recs, int iters, float seed) {
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
  unsigned long long mask = 0x5555aaaa3333ccccull ^ (unsigned long long)iters;
  if constexpr (KIND == CNDMASK_SET) asm volatile("v_cmp_gt_f32 vcc, %0, %1" ::"v"(v[0]), "v"(c) : "vcc");
  const uint64_t r0 = realtime();
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
        if constexpr (KIND == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j]) : "v"(c));
        if constexpr (KIND == FMAC) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(v[j]) : "v"(c));
        if constexpr (KIND == MUL_E64) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(v[j]) : "v"(c));
        if constexpr (KIND == FMA3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(c), "v"(v[(j + 5) & 15]));
        if constexpr (KIND == CNDMASK_SET) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[j]) : "v"(c));
        if constexpr (KIND == CNDMASK_E64) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[j]) : "v"(c), "s"(mask));
        if constexpr (KIND == MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[j]) : "v"(c));
        if constexpr (KIND == MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(v[j]) : "v"(c));
        if constexpr (KIND == LSHL) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(v[j]));
        if constexpr (KIND == CVT_F16) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(v[j]));
        if constexpr (KIND == MED3) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(v[j]) : "v"(c));
        if constexpr (KIND == FMA_MIX) asm volatile("v_fma_mix_f32 %0, %0, %1, %1" : "+v"(v[j]) : "v"(c));
        if constexpr (KIND == MUL_2W) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[j]) : "v"(v[(j + 5) & 15]));
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
  const uint64_t r1 = realtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i] + p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) {
    WaveRec r;
    r.t0 = t0, r.t1 = t1, r.r0 = r0, r.r1 = r1, r.hw = hw_id(), r.pad = 0;
    recs[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = r;
  }
}

struct Result {
  double wave_cpi, simd_cpi, ms, ghz, ns_per_inst;
  int hist[9];
};

template <int KIND>
Result run(float* out, WaveRec* rec_d, int n_cu, int W) {
  const int iters = 4000, threads = 256 * W, n_waves = n_cu * 4 * W;
  const size_t lds = 96 * 1024;
  hipFuncSetAttribute((const void*)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  k<KIND><<<n_cu, threads, lds>>>(out, rec_d, 16, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<KIND><<<n_cu, threads, lds>>>(out, rec_d, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  static WaveRec host[256 * 16];
  hipMemcpy(host, rec_d, n_waves * sizeof(WaveRec), hipMemcpyDeviceToHost);
  const double insts = (double)iters * 64;
  Result r;
  memset(&r, 0, sizeof(r));
  double wave_sum = 0, simd_sum = 0, clk_sum = 0;
  int n_simd = 0;
  for (int b = 0; b < n_cu; ++b) {
    // the waves of work-group b (= CU b: one work-group fits a CU) by SIMD
    for (int simd = 0; simd < 4; ++simd) {
      uint64_t first = ~0ull, lastt = 0;
      int cnt = 0;
      for (int w = 0; w < 4 * W; ++w) {
        const WaveRec& q = host[b * 4 * W + w];
        if ((int)((q.hw >> 4) & 3) != simd) continue;
        ++cnt;
        if (q.t0 < first) first = q.t0;
        if (q.t1 > lastt) lastt = q.t1;
      }
      r.hist[cnt < 8 ? cnt : 8]++;
      if (cnt) simd_sum += (double)(lastt - first) / (cnt * insts), ++n_simd;
    }
    for (int w = 0; w < 4 * W; ++w) {
      const WaveRec& q = host[b * 4 * W + w];
      wave_sum += (double)(q.t1 - q.t0);
      clk_sum += (double)(q.t1 - q.t0) / (double)(q.r1 - q.r0);
    }
  }
  r.wave_cpi = wave_sum / n_waves / insts;
  r.simd_cpi = simd_sum / n_simd;
  r.ms = ms;
  r.ghz = clk_sum / n_waves * 0.1;       // s_memrealtime: 100 MHz
  r.ns_per_inst = r.simd_cpi / r.ghz;
  return r;
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount;
  float* out;
  WaveRec* rec;
  hipMalloc(&out, (size_t)n_cu * 1024 * 4);
  hipMalloc(&rec, (size_t)n_cu * 16 * sizeof(WaveRec));
  printf("# %s, %d CUs, nominal clock %d MHz\n", prop.gcnArchName, n_cu, prop.clockRate / 1000);
  printf("# one work-group of 256*W threads per CU = W waves per SIMD intended; 4000 x 64 instructions per wave on 16 independent chains\n");
  printf("# cyc/inst/wave: one wave's own cadence.  cyc/inst/SIMD: per SIMD (last end - first start) / instructions issued there, mean over the SIMDs.\n");
  printf("# waves/SIMD histogram: number of SIMDs that hosted 0,1,2,... waves.  GHz: s_memtime ticks per s_memrealtime tick x 100 MHz.\n");
  printf("%-34s %2s %13s %13s %8s %8s %12s  %s\n", "instruction", "W", "cyc/inst/wave", "cyc/inst/SIMD", "GHz", "ms", "ns/inst/SIMD", "waves/SIMD histogram");
  for (int kind = 0; kind < N_KINDS; ++kind)
    for (int W = 1; W <= 4; W += (kind > MIX_FMA_EXP ? 1 : 1)) {
      if (kind > MIX_FMA_EXP && W == 3) continue;
      Result r;
      switch (kind) {
#define CASE(K_) case K_: r = run<K_>(out, rec, n_cu, W); break;
        CASE(FMA) CASE(MUL) CASE(PKFMA) CASE(PKMUL) CASE(EXP) CASE(RCP) CASE(DPP_SHR) CASE(DPP_BCAST) CASE(CNDMASK) CASE(CVT_PK_F16) CASE(FMA_DEP)
        CASE(MIX_FMA_EXP) CASE(ADD) CASE(FMAC) CASE(MUL_E64) CASE(FMA3) CASE(CNDMASK_SET) CASE(CNDMASK_E64) CASE(MAX) CASE(MOV) CASE(LSHL) CASE(CVT_F16)
        CASE(MED3) CASE(FMA_MIX) CASE(MUL_2W)
#undef CASE
      }
      printf("%-34s %2d %13.2f %13.2f %8.3f %8.3f %12.3f  ", kNames[kind], W, r.wave_cpi, r.simd_cpi, r.ghz, r.ms, r.ns_per_inst);
      for (int i = 0; i <= 8; ++i)
        if (r.hist[i]) printf("%d:%d ", i, r.hist[i]);
      printf("\n");
    }
  return 0;
}
