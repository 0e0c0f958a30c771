// Is an MFMA whose DESTINATION tuple overlaps its own A / B source registers computed correctly on MI355X?
// hipcc (ROCm 7.2) emits exactly that when srcC is the inline constant 0 and the sources die at the instruction
// (seen: v_mfma_f32_32x32x2_f32 v[2:17], v2, v3, 0), so the answer decides whether tools/check_mfma_overlap.py guards against a
// hardware hazard or against nothing.  Every lane evaluates the same product twice -- once with dst = v[34:49], A = v35, B = v36
// (overlapping), once with A = v50, B = v51 (disjoint) -- with explicit physical registers through inline asm, bit-compares the 16
// results and counts mismatches.  Partner waves on the same SIMDs keep the matrix and vector pipes busy (timing-dependent
// failures need contention).  Modes: single MFMA; MFMA followed back-to-back by a dependent accumulate into the same tuple (the
// pattern of a lin_in k-loop); the f16 32x32x16 form with dst overlapping its 4-register A / B tuples.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_dst_overlap.hip -o mfma_dst_overlap && ./mfma_dst_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(const float* __restrict__ in, unsigned long long* __restrict__ bad, int iters, int busy) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= 4) {   // contention: independent MFMA + VALU streams on the same SIMDs
    f32x16 t;
    for (int q = 0; q < 16; ++q) t[q] = (float)q;
    float a = in[lane], b = in[64 + lane], c = 0.f;
    for (int i = 0; i < busy; ++i) {
      t = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, t, 0, 0, 0);
      c = __builtin_fmaf(c, a, b);
    }
    if (t[3] == 12345.f || c == 54321.f) bad[1] = 1;
    return;
  }
  unsigned long long mism = 0;
  for (int it = 0; it < iters; ++it) {
    const float a = in[(it * 131 + lane + blockIdx.x * 17) & 4095], b = in[(it * 71 + lane * 3 + wave * 5 + 11) & 4095];
    const float a2 = in[(it * 37 + lane + 5) & 4095], b2 = in[(it * 97 + lane * 7 + 3) & 4095];
    f32x16 ov, cl;
    if constexpr (MODE == 0) {   // single fp32-input MFMA
      asm volatile("s_nop 4\n v_mfma_f32_32x32x2_f32 v[34:49], v35, v36, 0\n s_nop 15\n s_nop 7" : "={v[34:49]}"(ov) : "{v35}"(a), "{v36}"(b));
      asm volatile("s_nop 4\n v_mfma_f32_32x32x2_f32 v[34:49], v50, v51, 0\n s_nop 15\n s_nop 7" : "={v[34:49]}"(cl) : "{v50}"(a), "{v51}"(b));
    } else if constexpr (MODE == 1) {   // first MFMA of a chain (overlapping) + a dependent accumulate right behind it
      asm volatile("s_nop 4\n v_mfma_f32_32x32x2_f32 v[34:49], v35, v36, 0\n v_mfma_f32_32x32x2_f32 v[34:49], v52, v53, v[34:49]\n s_nop 15\n s_nop 7"
                   : "={v[34:49]}"(ov) : "{v35}"(a), "{v36}"(b), "{v52}"(a2), "{v53}"(b2));
      asm volatile("s_nop 4\n v_mfma_f32_32x32x2_f32 v[34:49], v50, v51, 0\n v_mfma_f32_32x32x2_f32 v[34:49], v52, v53, v[34:49]\n s_nop 15\n s_nop 7"
                   : "={v[34:49]}"(cl) : "{v50}"(a), "{v51}"(b), "{v52}"(a2), "{v53}"(b2));
    } else if constexpr (MODE == 2) {   // overlap with the LAST registers of the tuple (the ones written last)
      asm volatile("s_nop 4\n v_mfma_f32_32x32x2_f32 v[34:49], v48, v49, 0\n s_nop 15\n s_nop 7" : "={v[34:49]}"(ov) : "{v48}"(a), "{v49}"(b));
      asm volatile("s_nop 4\n v_mfma_f32_32x32x2_f32 v[34:49], v50, v51, 0\n s_nop 15\n s_nop 7" : "={v[34:49]}"(cl) : "{v50}"(a), "{v51}"(b));
    } else {   // f16 form: A = v[36:39], B = v[40:43] inside dst v[34:49]
      f32x4 av = {a, b, a2, b2}, bv = {b2, a2, b, a};   // arbitrary bit patterns are fine: both evaluations see the same halves
      asm volatile("s_nop 4\n v_mfma_f32_32x32x16_f16 v[34:49], v[36:39], v[40:43], 0\n s_nop 15\n s_nop 7" : "={v[34:49]}"(ov) : "{v[36:39]}"(av), "{v[40:43]}"(bv));
      asm volatile("s_nop 4\n v_mfma_f32_32x32x16_f16 v[34:49], v[52:55], v[56:59], 0\n s_nop 15\n s_nop 7" : "={v[34:49]}"(cl) : "{v[52:55]}"(av), "{v[56:59]}"(bv));
    }
    for (int q = 0; q < 16; ++q) mism += __float_as_uint(ov[q]) != __float_as_uint(cl[q]);
  }
  if (mism) atomicAdd(bad, mism);
}

template <int MODE>
void run(const char* name, const float* in, unsigned long long* d_bad) {
  hipMemset(d_bad, 0, 16);
  const int iters = 4000;
  k<MODE><<<1024, 512>>>(in, d_bad, iters, 3000);
  unsigned long long h[2];
  hipMemcpy(h, d_bad, 16, hipMemcpyDeviceToHost);
  printf("%-72s %llu mismatching values of %llu\n", name, h[0], 1024ull * 256 * 16 * iters);
}

int main() {
  float* in;
  unsigned long long* bad;
  hipMalloc(&in, 4096 * 4), hipMalloc(&bad, 16);
  static float h[4096];
  unsigned s = 12345;
  for (int i = 0; i < 4096; ++i) s = s * 1664525u + 1013904223u, h[i] = ((s >> 8) & 0xffff) / 4096.0f - 8.0f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  run<0>("fp32 32x32x2: dst v[34:49] overlaps A=v35, B=v36 (srcC = 0)", in, bad);
  run<1>("same + dependent accumulate into the tuple back to back", in, bad);
  run<2>("fp32 32x32x2: overlap with the tuple's last registers A=v48, B=v49", in, bad);
  run<3>("f16 32x32x16: dst v[34:49] overlaps A=v[36:39], B=v[40:43]", in, bad);
  return 0;
}
