// hipcc fuses  (_Float16)fmaf(a, b, c)  into v_fma_mixlo_f16 / v_fma_mixhi_f16 (one rounding instead of two).  The split-precision
// operands of lin_in are  hi = f16(e), lo = f16(e - float(hi))  with e = fmaf(-d, s, c) (the reference's "cos" entry) or e = t * q
// (angle doubling).  Is hi + lo == e to 2^-22 with the fused code hipcc generates?  Every (hi, lo) pair is checked in fp64 on the host.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/ubench/f16_split_fusion.hip -o f16_split_fusion
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <bool OPAQUE>
__global__ void k(const float* __restrict__ d, const float* __restrict__ s, const float* __restrict__ c, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float e0 = __builtin_fmaf(-d[i], s[i], c[i]);   // "cos" entry
  float e1 = (s[i] + s[i]) * c[i];                // doubled sine
  if (OPAQUE) asm("" : "+v"(e0), "+v"(e1));
  const _Float16 h0 = (_Float16)e0, h1 = (_Float16)e1;
  const _Float16 l0 = (_Float16)(e0 - (float)h0), l1 = (_Float16)(e1 - (float)h1);
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  // keep the packed (hi0, hi1) / (lo0, lo1) registers the kernel builds, then unpack
  h2 ph = {h0, h1}, pl = {l0, l1};
  asm volatile("" : "+v"(ph), "+v"(pl));
  out[6 * i + 0] = e0, out[6 * i + 1] = (float)ph[0], out[6 * i + 2] = (float)pl[0];
  out[6 * i + 3] = e1, out[6 * i + 4] = (float)ph[1], out[6 * i + 5] = (float)pl[1];
}

int main() {
  const int n = 1 << 24;
  std::vector<float> d(n), s(n), c(n), o(6ul * n);
  unsigned r = 12345;
  auto rnd = [&]() { r = r * 1664525u + 1013904223u; return (r >> 8) * (1.0f / 16777216.0f); };
  for (int i = 0; i < n; ++i) {
    const float a = (rnd() * 2 - 1) * 3.14159f * 16;
    s[i] = sinf(a), c[i] = cosf(a), d[i] = (rnd() * 2 - 1) * 4e-6f;
  }
  float *dd, *ds, *dc, *dout;
  hipMalloc(&dd, 4ul * n), hipMalloc(&ds, 4ul * n), hipMalloc(&dc, 4ul * n), hipMalloc(&dout, 24ul * n);
  hipMemcpy(dd, d.data(), 4ul * n, hipMemcpyHostToDevice), hipMemcpy(ds, s.data(), 4ul * n, hipMemcpyHostToDevice), hipMemcpy(dc, c.data(), 4ul * n, hipMemcpyHostToDevice);
  for (int opaque = 0; opaque < 2; ++opaque) {
    if (opaque) k<true><<<n / 256, 256>>>(dd, ds, dc, dout, n); else k<false><<<n / 256, 256>>>(dd, ds, dc, dout, n);
    hipMemcpy(o.data(), dout, 24ul * n, hipMemcpyDeviceToHost);
    double worst[2] = {0, 0};
    long bad[2] = {0, 0};
    int ex[2] = {-1, -1};
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < 2; ++j) {
        const double e = o[6ul * i + 3 * j], err = fabs((double)o[6ul * i + 3 * j + 1] + (double)o[6ul * i + 3 * j + 2] - e);
        if (err > worst[j]) worst[j] = err, ex[j] = i;
        bad[j] += err > 2.4e-7 * fabs(e) + 6e-8;
      }
    printf("%s: max |hi + lo - e| = %.3e (fma entry), %.3e (product entry); pairs worse than 2^-22 |e| + 2^-24: %ld, %ld of %d\n",
           opaque ? "fp32 value made opaque (no fusion)" : "as hipcc compiles it (v_fma_mix fusion)", worst[0], worst[1], bad[0], bad[1], n);
    for (int j = 0; j < 2; ++j)
      if (ex[j] >= 0 && worst[j] > 1e-6) printf("   e.g. e = %.9g hi = %.9g lo = %.9g\n", o[6ul * ex[j] + 3 * j], o[6ul * ex[j] + 3 * j + 1], o[6ul * ex[j] + 3 * j + 2]);
  }
  return 0;
}
