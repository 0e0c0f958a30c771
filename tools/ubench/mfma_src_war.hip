// Does an MFMA latch its A/B source registers at issue?  Overwrite srcA (or srcB) with the very next VALU instruction and compare
// the result with the undisturbed one.  fp32-input v_mfma_f32_32x32x2_f32 (16 passes) and v_mfma_f32_32x32x16_f16 (8 passes).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int MODE>  // 0 clean fp32, 1 fp32 + overwrite A next instr, 2 fp32 + overwrite B, 3 fp32 + overwrite A after 2 nops.. ; 10/11/12 f16
__global__ void k(float* out, int nwaves_busy) {
  const int lane = threadIdx.x & 63;
  f32x16 acc;
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  float a = 1.0f + 0.01f * lane, b = 2.0f - 0.02f * lane;
  h8 ah, bh;
  for (int e = 0; e < 8; ++e) ah[e] = (_Float16)(1.0f + 0.01f * lane + e), bh[e] = (_Float16)(2.0f - 0.02f * lane - e);
  if (threadIdx.x >= 64) {  // partner waves keep the SIMDs busy with MFMAs of their own
    f32x16 t = acc;
    for (int i = 0; i < nwaves_busy; ++i) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(t) : "v"(a), "v"(b));
    if (t[0] == 12345.f) out[0] = t[1];
    return;
  }
  for (int rep = 0; rep < 64; ++rep) {
    float a2 = a, b2 = b;
    h8 ah2 = ah, bh2 = bh;
    if (MODE == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n s_nop 15\n s_nop 15" : "+v"(acc), "+v"(a2), "+v"(b2));
    if (MODE == 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n v_mov_b32 %1, 0\n s_nop 15\n s_nop 15" : "+v"(acc), "+v"(a2), "+v"(b2));
    if (MODE == 2) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n v_mov_b32 %2, 0\n s_nop 15\n s_nop 15" : "+v"(acc), "+v"(a2), "+v"(b2));
    if (MODE == 3) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n s_nop 7\n v_mov_b32 %1, 0\n s_nop 15\n s_nop 15" : "+v"(acc), "+v"(a2), "+v"(b2));
    if (MODE == 4) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n s_nop 15\n s_nop 15\n v_mov_b32 %1, 0\n s_nop 15\n s_nop 15" : "+v"(acc), "+v"(a2), "+v"(b2));
    if (MODE == 10) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n s_nop 15\n s_nop 15" : "+v"(acc), "+v"(ah2), "+v"(bh2));
  }
  for (int q = 0; q < 16; ++q) out[(blockIdx.x * 64 + lane) * 16 + q] = acc[q];
}

template <int MODE>
void run(const char* name, float* d, float* ref, bool is_ref) {
  static float h[256 * 64 * 16];
  k<MODE><<<256, 512>>>(d, 200);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  if (is_ref) {
    for (int i = 0; i < 256 * 64 * 16; ++i) ref[i] = h[i];
    printf("%-44s reference\n", name);
    return;
  }
  int bad = 0, badlane[4] = {0, 0, 0, 0};
  for (int i = 0; i < 256 * 64 * 16; ++i)
    if (h[i] != ref[i]) bad++, badlane[((i / 16) % 64) / 16]++;
  printf("%-44s mismatching values %d (lanes 0-15: %d, 16-31: %d, 32-47: %d, 48-63: %d)\n", name, bad, badlane[0], badlane[1], badlane[2], badlane[3]);
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 64 * 16 * 4);
  static float ref32[256 * 64 * 16], ref16[256 * 64 * 16];
  run<0>("fp32 MFMA, sources untouched", d, ref32, true);
  run<0>("fp32 MFMA, sources untouched (again)", d, ref32, false);
  run<1>("fp32 MFMA, srcA overwritten next instruction", d, ref32, false);
  run<2>("fp32 MFMA, srcB overwritten next instruction", d, ref32, false);
  run<3>("fp32 MFMA, srcA overwritten after s_nop 7", d, ref32, false);
  run<4>("fp32 MFMA, srcA overwritten after 32 nops", d, ref32, false);
  run<10>("f16 MFMA, sources untouched", d, ref16, true);
  run<10>("f16 MFMA, sources untouched (again)", d, ref16, false);
  return 0;
}
