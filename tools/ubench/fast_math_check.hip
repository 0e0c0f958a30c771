// How accurate are the hand-rolled exp / softplus / sigmoid / reciprocal / quotient of bts_common.h (v_exp_f32, v_log_f32, v_rcp_f32
// plus explicit correction steps) against fp64, next to libm's expf / log1pf and the compiler's IEEE division?  Dense sweeps over the
// ranges the renderer meets: s in [-40, 25] (pre-softplus MLP output), x = -delta sigma in [-1e4, 0], denominators in [1e-3, 1e3].
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -I behindthescenes_amd/csrc tools/ubench/fast_math_check.hip -o fast_math_check
#include "bts_common.h"

#include <cmath>
#include <cstdio>
#include <vector>

using namespace bts;

__device__ __forceinline__ float softplus_libm(float s) { return s > 20.0f ? s : log1pf(expf(s)); }
__device__ __forceinline__ float sigmoid_libm(float s) { return 1.0f / (1.0f + expf(-s)); }

// mode 0: softplus, 1: exp (x <= 0), 2: sigmoid, 3: quotient a / b and reciprocal
__global__ void k(int mode, const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ fast, float* __restrict__ lib,
                  double* __restrict__ ref, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = x[i], b = y[i];
  if (mode == 0) fast[i] = softplus(a), lib[i] = softplus_libm(a), ref[i] = a > 20.0f ? (double)a : log1p(exp((double)a));
  if (mode == 1) fast[i] = transmittance(a, 1.0f), lib[i] = expf(-fabsf(a)), ref[i] = exp(-fabs((double)a));
  if (mode == 2) fast[i] = sigmoidf(a), lib[i] = sigmoid_libm(a), ref[i] = 1.0 / (1.0 + exp(-(double)a));
  if (mode == 3) fast[i] = div_by(a, b, rcp_nr(b)), lib[i] = a / b, ref[i] = (double)a / (double)b;
  if (mode == 4) fast[i] = rcp_nr(b), lib[i] = 1.0f / b, ref[i] = 1.0 / (double)b;
}

static double ulp_of(double v) {
  float f = (float)fabs(v);
  if (f < 1.17549435e-38f) return 1.4e-45;
  int e;
  frexpf(f, &e);
  return ldexp(1.0, e - 24);
}

int main() {
  const int n = 1 << 24;
  std::vector<float> x(n), y(n), f(n), l(n);
  std::vector<double> r(n);
  float *dx, *dy, *df, *dl;
  double* dr;
  hipMalloc(&dx, 4ul * n), hipMalloc(&dy, 4ul * n), hipMalloc(&df, 4ul * n), hipMalloc(&dl, 4ul * n), hipMalloc(&dr, 8ul * n);
  unsigned s = 777;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0 / 16777216.0); };
  const char* names[5] = {"softplus(s), s in [-40, 25]", "exp(-x), x in [0, 1e4] (log-uniform) ", "sigmoid(s), s in [-40, 25]", "a / b, |a| <= 1e3, b in [1e-3, 1e3]",
                          "1 / b, b in [1e-3, 1e3]"};
  for (int mode = 0; mode < 5; ++mode) {
    for (int i = 0; i < n; ++i) {
      if (mode == 0 || mode == 2) x[i] = (float)(-40.0 + 65.0 * (i + rnd()) / n);
      if (mode == 1) x[i] = (float)exp(log(1e-6) + (log(1e4) - log(1e-6)) * (i + rnd()) / n);
      if (mode >= 3) x[i] = (float)((rnd() * 2 - 1) * 1e3), y[i] = (float)exp(log(1e-3) + (log(1e3) - log(1e-3)) * rnd());
    }
    hipMemcpy(dx, x.data(), 4ul * n, hipMemcpyHostToDevice), hipMemcpy(dy, y.data(), 4ul * n, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(mode, dx, dy, df, dl, dr, n);
    hipMemcpy(f.data(), df, 4ul * n, hipMemcpyDeviceToHost), hipMemcpy(l.data(), dl, 4ul * n, hipMemcpyDeviceToHost);
    hipMemcpy(r.data(), dr, 8ul * n, hipMemcpyDeviceToHost);
    double wf = 0, wl = 0, af = 0, al = 0, sf = 0, sl = 0;
    long differ = 0, nonfinite = 0;
    int ef = 0, el = 0;
    for (int i = 0; i < n; ++i) {
      if (!std::isfinite(f[i])) { ++nonfinite; continue; }
      const double u = ulp_of(r[i]);
      const double e0 = fabs((double)f[i] - r[i]), e1 = fabs((double)l[i] - r[i]);
      if (e0 / u > wf) wf = e0 / u, ef = i;
      if (e1 / u > wl) wl = e1 / u, el = i;
      af = fmax(af, e0), al = fmax(al, e1), sf += (e0 / u) * (e0 / u), sl += (e1 / u) * (e1 / u);
      differ += f[i] != l[i];
    }
    printf("%-40s fast: max %.3f ulp (at %.9g) rms %.3f ulp max abs %.3e | libm / IEEE: max %.3f ulp (at %.9g) rms %.3f ulp max abs %.3e | bitwise different: %ld of %d, non-finite: %ld\n",
           names[mode], wf, x[ef], sqrt(sf / n), af, wl, x[el], sqrt(sl / n), al, differ, n, nonfinite);
  }
  // the compositing edge cases: delta = 1e10 (last sample), sigma 0 / tiny / huge
  {
    const float xs[8] = {0.0f, 1e-30f, 1e-10f, 1.0f, 3e4f, 1e10f, 3e20f, 3.3e38f};
    for (int i = 0; i < 8; ++i) x[i] = xs[i];
    hipMemcpy(dx, x.data(), 32, hipMemcpyHostToDevice);
    k<<<1, 8>>>(1, dx, dy, df, dl, dr, 8);
    hipMemcpy(f.data(), df, 32, hipMemcpyDeviceToHost), hipMemcpy(l.data(), dl, 32, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) printf("  exp(-%g): fast %.9g libm %.9g\n", xs[i], f[i], l[i]);
    const float ss[8] = {-200.0f, -104.0f, -88.0f, -87.0f, 19.999f, 20.0f, 20.001f, 1e30f};
    for (int i = 0; i < 8; ++i) x[i] = ss[i];
    hipMemcpy(dx, x.data(), 32, hipMemcpyHostToDevice);
    for (int mode = 0; mode <= 2; mode += 2) {
      k<<<1, 8>>>(mode, dx, dy, df, dl, dr, 8);
      hipMemcpy(f.data(), df, 32, hipMemcpyDeviceToHost), hipMemcpy(l.data(), dl, 32, hipMemcpyDeviceToHost);
      for (int i = 0; i < 8; ++i) printf("  %s(%g): fast %.9g libm %.9g\n", mode ? "sigmoid" : "softplus", ss[i], f[i], l[i]);
    }
  }
  return 0;
}
