// Validates the split-precision trick used by the render kernel: a fp32 GEMM tile D[32x32] = A[32xK] . B[Kx32] evaluated on the
// f16 matrix pipe (v_mfma_f32_32x32x16_f16) as hi.hi + lo.hi + hi.lo with x = hi + lo, hi = f16_rne(x), lo = f16_rne(x - hi),
// against an fp64 host reference and against the fp32-input MFMA.  Checks layout, subnormal handling and accuracy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__global__ void k(const float* A, const float* B, float* D16, float* D32, int K, float scale_a) {
  const int lane = threadIdx.x, i = lane & 31, g = lane >> 5;
  f32x16 acc16, acc32;
  for (int q = 0; q < 16; ++q) acc16[q] = 0.f, acc32[q] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    h8 ah, al, bh, bl;
    for (int e = 0; e < 8; ++e) {
      const float a = A[i * K + k0 + 8 * g + e] * scale_a;   // A[i][k]
      const float b = B[(k0 + 8 * g + e) * 32 + i];           // B[k][j=i]
      const _Float16 a_hi = (_Float16)a, b_hi = (_Float16)b;
      ah[e] = a_hi, al[e] = (_Float16)(a - (float)a_hi);
      bh[e] = b_hi, bl[e] = (_Float16)(b - (float)b_hi);
    }
    acc16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc16, 0, 0, 0);
    acc16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc16, 0, 0, 0);
    acc16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc16, 0, 0, 0);
  }
  for (int k0 = 0; k0 < K; k0 += 2) {
    acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k0 + g], B[(k0 + g) * 32 + i], acc32, 0, 0, 0);
  }
  for (int q = 0; q < 16; ++q) {
    const int row = (q & 3) + 8 * (q >> 2) + 4 * g;
    D16[row * 32 + i] = acc16[q] / scale_a;
    D32[row * 32 + i] = acc32[q];
  }
}

int main() {
  const int K = 48;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 0.14f);
  std::uniform_real_distribution<float> ud(-1.f, 1.f);
  for (int test = 0; test < 3; ++test) {
    std::vector<float> A(32 * K), B(K * 32), D16(1024), D32(1024);
    for (auto& a : A) a = nd(rng) * (test == 2 ? 1e-3f : 1.f);
    for (auto& b : B) b = test == 1 ? ud(rng) * 1e-3f : sinf(ud(rng) * 50.f);   // test 1: tiny PE values (subnormal lo parts)
    float *dA, *dB, *d16, *d32;
    hipMalloc(&dA, A.size() * 4), hipMalloc(&dB, B.size() * 4), hipMalloc(&d16, 4096), hipMalloc(&d32, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice), hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, d16, d32, K, 256.f);
    hipMemcpy(D16.data(), d16, 4096, hipMemcpyDeviceToHost), hipMemcpy(D32.data(), d32, 4096, hipMemcpyDeviceToHost);
    double e16 = 0, e32 = 0, mx = 0, r16 = 0, r32 = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        double ref = 0;
        for (int kk = 0; kk < K; ++kk) ref += (double)A[i * K + kk] * (double)B[kk * 32 + j];
        e16 = fmax(e16, fabs(D16[i * 32 + j] - ref)), e32 = fmax(e32, fabs(D32[i * 32 + j] - ref)), mx = fmax(mx, fabs(ref));
        r16 += (D16[i * 32 + j] - ref) * (D16[i * 32 + j] - ref), r32 += (D32[i * 32 + j] - ref) * (D32[i * 32 + j] - ref);
      }
    printf("test %d: max|D| %.3e  f16-split: max err %.3e rms %.3e | fp32 MFMA: max err %.3e rms %.3e\n", test, mx, e16, sqrt(r16 / 1024),
           e32, sqrt(r32 / 1024));
  }
  return 0;
}
