// Do the three fp32 -> f16 conversions hipcc mixes freely on gfx950 agree bit for bit?
//   v_cvt_f16_f32 (scalar), v_cvt_pk_f16_f32 (new on gfx950, two at a time), v_fma_mixlo_f16 x, 1.0, 0 (fused multiply + convert)
// The split-precision lin_in computes  hi = f16(e), lo = f16(e - float(hi))  and the compiler is free to evaluate the two f16(e) with
// DIFFERENT instructions (one packed for the MFMA operand, one scalar for the subtraction).  If they ever disagree, hi + lo is off by
// one f16 ulp of e.  All 2^32 bit patterns are checked.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/cvt_f16_forms.hip -o cvt_f16_forms && ./cvt_f16_forms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

__global__ void k(unsigned long long* out) {
  const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long bad_pk_lo = 0, bad_pk_hi = 0, bad_mix = 0;
  for (unsigned long long i = tid; i < (1ull << 32); i += (unsigned long long)gridDim.x * blockDim.x) {
    const float x = __uint_as_float((unsigned)i);
    unsigned a, b, c;
    asm volatile("v_cvt_f16_f32 %0, %3\n v_cvt_pk_f16_f32 %1, %3, %3\n v_fma_mixlo_f16 %2, %3, 1.0, 0" : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(x));
    a &= 0xffff;
    const unsigned lo = b & 0xffff, hi = b >> 16, m = c & 0xffff;
    const bool nan = (x != x);
    if (!nan && lo != a) { if (!bad_pk_lo++ && atomicAdd(out + 3, 1ull) == 0) out[4] = (i << 32) | (a << 16) | lo; }
    if (!nan && hi != a) bad_pk_hi++;
    if (!nan && m != a) { if (!bad_mix++ && atomicAdd(out + 5, 1ull) == 0) out[6] = (i << 32) | (a << 16) | m; }
  }
  if (bad_pk_lo) atomicAdd(out, bad_pk_lo);
  if (bad_pk_hi) atomicAdd(out + 1, bad_pk_hi);
  if (bad_mix) atomicAdd(out + 2, bad_mix);
}

int main() {
  unsigned long long *d, h[8];
  if (hipMalloc(&d, 64) != hipSuccess) return 1;
  (void)hipMemset(d, 0, 64);
  k<<<4096, 256>>>(d);
  (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
  printf("non-NaN inputs where v_cvt_pk_f16_f32 (low half / high half) or v_fma_mixlo_f16 differ from v_cvt_f16_f32: %llu / %llu / %llu of 2^32\n", h[0], h[1], h[2]);
  auto ex = [](unsigned long long v, const char* what) {
    unsigned u = (unsigned)(v >> 32); float x; memcpy(&x, &u, 4);
    printf("  e.g. x = %.9g (0x%08x): v_cvt_f16_f32 -> 0x%04x, %s -> 0x%04x\n", x, u, (unsigned)(v >> 16) & 0xffff, what, (unsigned)v & 0xffff);
  };
  if (h[0]) ex(h[4], "v_cvt_pk_f16_f32");
  if (h[2]) ex(h[6], "v_fma_mixlo_f16");
  return 0;
}
