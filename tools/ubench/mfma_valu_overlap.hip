// Micro-benchmark: do v_mfma_f32_32x32x2_f32 (fp32-input MFMA) and plain fp32 VALU work overlap on one SIMD of gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_overlap.hip -o gpurun_out/mfma_valu && gpurun_out/mfma_valu
// Modes: 0 = MFMA only, 1 = VALU only, 2 = both interleaved in ONE wave (per MFMA: NV fmas), 3 = waves alternate roles (even waves
// MFMA-only, odd waves VALU-only; 2 waves per SIMD), 4 = bf16 MFMA only, 5 = bf16 MFMA + VALU in one wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.0f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed + lane * 0.001f + i;
  float a = seed * lane, b = seed + 1.0f;
  bf16x8 ab, bb;
  for (int i = 0; i < 8; ++i) ab[i] = (short)(lane + i), bb[i] = (short)(lane * 3 + i);
  const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && !(wave & 1)) || MODE == 4 || MODE == 5;
  const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 1)) || MODE == 5;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (do_m) {
        if (MODE >= 4) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[m], 0, 0, 0);
        else acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
      }
      if (do_v) {
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], 1.0001f, 0.5f);
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NV>
float run(float* out, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  k<MODE, NV><<<blocks, 256>>>(out, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE, NV><<<blocks, 256>>>(out, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 20000;
  for (int blocks : {256, 512}) {  // 1 / 2 work-groups per CU = 1 / 2 waves per SIMD
    printf("blocks=%d (%d waves/SIMD), %d iters x 4 MFMA, NV fmas per MFMA\n", blocks, blocks / 256, iters);
    printf("  fp32 MFMA only          : %.3f ms\n", run<0, 16>(out, blocks, iters));
    printf("  VALU only  NV=16        : %.3f ms\n", run<1, 16>(out, blocks, iters));
    printf("  VALU only  NV=32        : %.3f ms\n", run<1, 32>(out, blocks, iters));
    printf("  one wave both NV=16     : %.3f ms\n", run<2, 16>(out, blocks, iters));
    printf("  one wave both NV=32     : %.3f ms\n", run<2, 32>(out, blocks, iters));
    printf("  alternate waves NV=16   : %.3f ms\n", run<3, 16>(out, blocks, iters));
    printf("  alternate waves NV=32   : %.3f ms\n", run<3, 32>(out, blocks, iters));
    printf("  bf16 MFMA only          : %.3f ms\n", run<4, 16>(out, blocks, iters));
    printf("  bf16 MFMA + VALU NV=8   : %.3f ms\n", run<5, 8>(out, blocks, iters));
    printf("  bf16 MFMA + VALU NV=16  : %.3f ms\n", run<5, 16>(out, blocks, iters));
  }
  return 0;
}
