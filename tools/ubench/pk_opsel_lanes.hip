// Packed-FP32 operand selection on MI355X: which op_sel / op_sel_hi forms of v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 / v_pk_mov_b32
// return wrong values, in which lanes, and under which activity of the OTHER wave on the same SIMD.
// Found in round 2 while bisecting a timing-dependent colour error of bts::render_kernel_p: the compiler's SLP vectoriser turns
// scalar bilinear blends into v_pk_mul_f32 ... op_sel:[0,1] (low result = src0.lo * src1.HI); that form returned 0 in the low result of
// lanes 48-63 a few per cent of the time while a partner wave ran MFMAs -- no memory instruction involved (operands written by v_mov
// dozens of cycles earlier).  Every lane compares the packed instruction (explicit registers, inline asm) with scalar arithmetic.
// (-fno-slp-vectorize: the scalar reference arithmetic must not be turned into the very instructions under test.)
//   hipcc --offload-arch=gfx950 -O2 -fno-slp-vectorize tools/ubench/pk_opsel_lanes.hip -o pk_opsel_lanes && ./pk_opsel_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define PRE_ "v_mov_b32 v12, %2\n v_mov_b32 v13, %3\n v_mov_b32 v28, %4\n v_mov_b32 v29, %5\n v_mov_b32 v30, %6\n v_mov_b32 v31, %7\n v_mov_b32 v2, 0x7fc00000\n v_mov_b32 v3, 0x7fc00000\n s_nop 15\n"
#define POST_ "s_nop 15\n s_nop 15\n v_mov_b32 %0, v2\n v_mov_b32 %1, v3"
#define OPS_ : "=&v"(r0), "=&v"(r1) : "v"(s0[0]), "v"(s0[1]), "v"(s1[0]), "v"(s1[1]), "v"(s2[0]), "v"(s2[1]) : "v2", "v3", "v12", "v13", "v28", "v29", "v30", "v31"

#define OPS2_ : "=&v"(r0), "=&v"(r1) : "v"(s0[0]), "v"(s0[1]), "v"(s1[0]), "v"(s1[1]), "v"(s2[0]), "v"(s2[1]) : "v2", "v3", "v12", "v13", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55"

__global__ __launch_bounds__(512) void k(int mode, int partner, unsigned long long* __restrict__ out, int iters) {
  const int lane = threadIdx.x & 63;
  __shared__ volatile int done;
  __shared__ float lds_w[4096];
  if (threadIdx.x == 0) done = 0;
  for (int i = threadIdx.x; i < 4096; i += 512) lds_w[i] = 0.001f * i;
  __syncthreads();
  if (threadIdx.x >= 256) {   // partner waves: one per SIMD, busy until the probe waves are done
    f32x16 acc;
    for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
    h8 a, b;
    for (int e = 0; e < 8; ++e) a[e] = (_Float16)(0.01f * (lane + e)), b[e] = (_Float16)(0.02f * (lane - e));
    f32x2 p = {1.0f, 2.0f}, q2 = {0.5f, 0.25f};
    float f = 1.0f, g = 0.5f;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
    typedef int i32x16 __attribute__((ext_vector_type(16)));
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
    h4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
    bf8 ab, bb;
    for (int e = 0; e < 8; ++e) ab[e] = (__bf16)(0.01f * (lane + e)), bb[e] = (__bf16)(0.02f * (lane - e));
    i32x16 acci;
    for (int q = 0; q < 16; ++q) acci[q] = 0;
    i32x4 ai = {lane, lane + 1, lane + 2, lane + 3}, bi = {1, 2, 3, lane};
    int guard = 0;
    while (!done && guard < (1 << 22)) {
      for (int r = 0; r < 16; ++r) {
        if (partner == 1 || partner == 6) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        if (partner == 5) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f, g, acc, 0, 0, 0);
        if (partner == 7) acc4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc4, 0, 0, 0);
        if (partner == 8) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc, 0, 0, 0);
        if (partner == 9) acc = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, acc, 0, 0, 0);
        if (partner == 10) acc4 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc4, 0, 0, 0);
        if (partner == 11) acci = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai, bi, acci, 0, 0, 0);
        if (partner == 2 || partner == 6) p = __builtin_elementwise_fma(p, q2, (f32x2){0.5f, 0.25f}), q2 = __builtin_elementwise_fma(q2, p, (f32x2){0.125f, 0.5f});
        if (partner == 3 || partner == 6) {
          const float4 wv = *reinterpret_cast<const float4*>(&lds_w[((lane + r + guard) & 1023) * 4]);
          f += wv.x + wv.w;
        }
        if (partner == 4) f = __builtin_fmaf(f, g, 0.5f), g = __builtin_fmaf(g, f, 0.25f);
        if (partner == 0) __builtin_amdgcn_s_sleep(2);
      }
      guard += 16;
    }
    if (acc[3] == 12345.0f || acc4[1] == 4321.0f || acci[2] == 424242 || q2[1] == 777.0f || f == 999.0f || g == 998.0f) out[7] = 1;
    return;
  }
  unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  unsigned long long bad_lo[4] = {0, 0, 0, 0}, bad_hi[4] = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    const f32x2 s0 = {1.0f + (float)(s & 1023) * 0.01f, 3.0f + (float)((s >> 10) & 1023) * 0.02f};
    const f32x2 s1 = {1.5f + 0.001f * lane, 2.0f + 0.003f * (it & 15)};
    const f32x2 s2 = {0.25f + (float)((s >> 20) & 255), 7.0f - 0.1f * (lane & 7)};
    float r0, r1, e0 = 0.f, e1 = 0.f;
    switch (mode) {
      case 0: asm volatile(PRE_ "v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[0,0] op_sel_hi:[0,0]\n" POST_ OPS_); e0 = s0[0] * s1[0], e1 = s0[0] * s1[0]; break;
      case 1: asm volatile(PRE_ "v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[0,0] op_sel_hi:[0,1]\n" POST_ OPS_); e0 = s0[0] * s1[0], e1 = s0[0] * s1[1]; break;
      case 2: asm volatile(PRE_ "v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[0,0] op_sel_hi:[1,0]\n" POST_ OPS_); e0 = s0[0] * s1[0], e1 = s0[1] * s1[0]; break;
      case 3: asm volatile(PRE_ "v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[0,0] op_sel_hi:[1,1]\n" POST_ OPS_); e0 = s0[0] * s1[0], e1 = s0[1] * s1[1]; break;
      case 4: asm volatile(PRE_ "v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[0,1] op_sel_hi:[0,0]\n" POST_ OPS_); e0 = s0[0] * s1[1], e1 = s0[0] * s1[0]; break;
      case 5: asm volatile(PRE_ "v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[0,1] op_sel_hi:[0,1]\n" POST_ OPS_); e0 = s0[0] * s1[1], e1 = s0[0] * s1[1]; break;
      case 6: asm volatile(PRE_ "v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[0,1] op_sel_hi:[1,0]\n" POST_ OPS_); e0 = s0[0] * s1[1], e1 = s0[1] * s1[0]; break;
      case 7: asm volatile(PRE_ "v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[0,1] op_sel_hi:[1,1]\n" POST_ OPS_); e0 = s0[0] * s1[1], e1 = s0[1] * s1[1]; break;
      case 8: asm volatile(PRE_ "v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[1,0] op_sel_hi:[0,0]\n" POST_ OPS_); e0 = s0[1] * s1[0], e1 = s0[0] * s1[0]; break;
      case 9: asm volatile(PRE_ "v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[1,0] op_sel_hi:[0,1]\n" POST_ OPS_); e0 = s0[1] * s1[0], e1 = s0[0] * s1[1]; break;
      case 10: asm volatile(PRE_ "v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[1,0] op_sel_hi:[1,0]\n" POST_ OPS_); e0 = s0[1] * s1[0], e1 = s0[1] * s1[0]; break;
      case 11: asm volatile(PRE_ "v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[1,0] op_sel_hi:[1,1]\n" POST_ OPS_); e0 = s0[1] * s1[0], e1 = s0[1] * s1[1]; break;
      case 12: asm volatile(PRE_ "v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[1,1] op_sel_hi:[0,0]\n" POST_ OPS_); e0 = s0[1] * s1[1], e1 = s0[0] * s1[0]; break;
      case 13: asm volatile(PRE_ "v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[1,1] op_sel_hi:[0,1]\n" POST_ OPS_); e0 = s0[1] * s1[1], e1 = s0[0] * s1[1]; break;
      case 14: asm volatile(PRE_ "v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[1,1] op_sel_hi:[1,0]\n" POST_ OPS_); e0 = s0[1] * s1[1], e1 = s0[1] * s1[0]; break;
      case 15: asm volatile(PRE_ "v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[1,1] op_sel_hi:[1,1]\n" POST_ OPS_); e0 = s0[1] * s1[1], e1 = s0[1] * s1[1]; break;
      case 16: asm volatile(PRE_ "v_pk_add_f32 v[2:3], v[12:13], v[28:29] op_sel:[0,1]\n" POST_ OPS_); e0 = s0[0] + s1[1], e1 = s0[1] + s1[1]; break;
      case 17: asm volatile(PRE_ "v_pk_fma_f32 v[2:3], v[12:13], v[28:29], v[30:31] op_sel:[0,1,0]\n" POST_ OPS_); e0 = __builtin_fmaf(s0[0], s1[1], s2[0]), e1 = __builtin_fmaf(s0[1], s1[1], s2[1]); break;
      case 18: asm volatile(PRE_ "v_pk_fma_f32 v[2:3], v[12:13], v[28:29], v[30:31] op_sel:[0,0,1]\n" POST_ OPS_); e0 = __builtin_fmaf(s0[0], s1[0], s2[1]), e1 = __builtin_fmaf(s0[1], s1[1], s2[1]); break;
      case 19: asm volatile(PRE_ "v_pk_mov_b32 v[2:3], v[12:13], v[28:29] op_sel:[1,0]\n" POST_ OPS_); e0 = s0[1], e1 = s1[0]; break;
      case 20: asm volatile(PRE_ "v_pk_mov_b32 v[2:3], v[12:13], v[28:29] op_sel:[0,1]\n" POST_ OPS_); e0 = s0[0], e1 = s1[1]; break;
      case 21: asm volatile(PRE_ "v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1]\n" POST_ OPS_); e0 = s0[0] * -s1[1], e1 = s0[1] * s1[1]; break;
      // mixed-precision FMAs (VOP3P encoded as well): f16(a * b + c) from fp32 sources, low / high half of the destination
      case 22: asm volatile(PRE_ "v_fma_mixlo_f16 v2, v12, v28, v30\n v_cvt_f32_f16 v2, v2\n v_mov_b32 v3, 0\n" POST_ OPS_); e0 = (float)(_Float16)__builtin_fmaf(s0[0], s1[0], s2[0]), e1 = 0.0f; break;
      case 23: asm volatile(PRE_ "v_mov_b32 v2, 0\n v_fma_mixhi_f16 v2, v12, v28, v30\n v_lshrrev_b32 v2, 16, v2\n v_cvt_f32_f16 v2, v2\n v_mov_b32 v3, 0\n" POST_ OPS_); e0 = (float)(_Float16)__builtin_fmaf(s0[0], s1[0], s2[0]), e1 = 0.0f; break;
      case 24: asm volatile(PRE_ "v_fma_mixlo_f16 v2, -v12, v28, v30\n v_cvt_f32_f16 v2, v2\n v_mov_b32 v3, 0\n" POST_ OPS_); e0 = (float)(_Float16)__builtin_fmaf(-s0[0], s1[0], s2[0]), e1 = 0.0f; break;
      case 25: asm volatile(PRE_ "v_fma_mixlo_f16 v2, v12, v28, 0\n v_cvt_f32_f16 v2, v2\n v_mov_b32 v3, 0\n" POST_ OPS_); e0 = (float)(_Float16)(s0[0] * s1[0]), e1 = 0.0f; break;
      case 26: asm volatile(PRE_ "v_cvt_pk_f16_f32 v2, v12, v28\n v_lshrrev_b32 v3, 16, v2\n v_cvt_f32_f16 v2, v2\n v_cvt_f32_f16 v3, v3\n" POST_ OPS_); e0 = (float)(_Float16)s0[0], e1 = (float)(_Float16)s1[0]; break;
      // 100 + d: the probe wave's OWN wide MFMA issued, then d idle cycles, then the packed multiply (partners idle)
      case 100: asm volatile(PRE_ "v_mfma_f32_32x32x16_f16 v[40:55], v[32:35], v[36:39], 0\n v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[0,1]\n" POST_ OPS2_); e0 = s0[0] * s1[1], e1 = s0[1] * s1[1]; break;
      case 104: asm volatile(PRE_ "v_mfma_f32_32x32x16_f16 v[40:55], v[32:35], v[36:39], 0\n s_nop 3\n v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[0,1]\n" POST_ OPS2_); e0 = s0[0] * s1[1], e1 = s0[1] * s1[1]; break;
      case 108: asm volatile(PRE_ "v_mfma_f32_32x32x16_f16 v[40:55], v[32:35], v[36:39], 0\n s_nop 7\n v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[0,1]\n" POST_ OPS2_); e0 = s0[0] * s1[1], e1 = s0[1] * s1[1]; break;
      case 116: asm volatile(PRE_ "v_mfma_f32_32x32x16_f16 v[40:55], v[32:35], v[36:39], 0\n s_nop 15\n v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[0,1]\n" POST_ OPS2_); e0 = s0[0] * s1[1], e1 = s0[1] * s1[1]; break;
      case 124: asm volatile(PRE_ "v_mfma_f32_32x32x16_f16 v[40:55], v[32:35], v[36:39], 0\n s_nop 15\n s_nop 7\n v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[0,1]\n" POST_ OPS2_); e0 = s0[0] * s1[1], e1 = s0[1] * s1[1]; break;
      case 132: asm volatile(PRE_ "v_mfma_f32_32x32x16_f16 v[40:55], v[32:35], v[36:39], 0\n s_nop 15\n s_nop 15\n v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[0,1]\n" POST_ OPS2_); e0 = s0[0] * s1[1], e1 = s0[1] * s1[1]; break;
      case 148: asm volatile(PRE_ "v_mfma_f32_32x32x16_f16 v[40:55], v[32:35], v[36:39], 0\n s_nop 15\n s_nop 15\n s_nop 15\n v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[0,1]\n" POST_ OPS2_); e0 = s0[0] * s1[1], e1 = s0[1] * s1[1]; break;
      default: return;
    }
    bad_lo[lane >> 4] += __float_as_uint(r0) != __float_as_uint(e0);
    bad_hi[lane >> 4] += __float_as_uint(r1) != __float_as_uint(e1);
    if (__float_as_uint(r0) != __float_as_uint(e0) && atomicAdd(out + 8, 1ull) == 0) out[9] = ((unsigned long long)__float_as_uint(r0) << 32) | __float_as_uint(e0);
  }
  for (int q = 0; q < 4; ++q) {
    if (bad_lo[q]) atomicAdd(out + q, bad_lo[q]);
    if (bad_hi[q]) atomicAdd(out + 4 + q, bad_hi[q]);
  }
  __builtin_amdgcn_s_waitcnt(0);
  done = 1;
}

static const char* kNames[] = {/*22..26 appended below*/"v_pk_mul_f32 op_sel:[0,0] op_sel_hi:[0,0]", "v_pk_mul_f32 op_sel:[0,0] op_sel_hi:[0,1]", "v_pk_mul_f32 op_sel:[0,0] op_sel_hi:[1,0]", "v_pk_mul_f32 op_sel:[0,0] op_sel_hi:[1,1]", "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[0,0]", "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[0,1]", "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,1]", "v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,0]", "v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1]", "v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[1,0]", "v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[1,1]", "v_pk_mul_f32 op_sel:[1,1] op_sel_hi:[0,0]", "v_pk_mul_f32 op_sel:[1,1] op_sel_hi:[0,1]", "v_pk_mul_f32 op_sel:[1,1] op_sel_hi:[1,0]", "v_pk_mul_f32 op_sel:[1,1] op_sel_hi:[1,1]", "v_pk_add_f32 op_sel:[0,1]", "v_pk_fma_f32 op_sel:[0,1,0]", "v_pk_fma_f32 op_sel:[0,0,1]", "v_pk_mov_b32 op_sel:[1,0]", "v_pk_mov_b32 op_sel:[0,1]", "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1]", "v_fma_mixlo_f16 a, b, c (fp32 sources)", "v_fma_mixhi_f16 a, b, c", "v_fma_mixlo_f16 -a, b, c", "v_fma_mixlo_f16 a, b, 0", "v_cvt_pk_f16_f32"};
static const char* kPartners[] = {"idle (s_sleep)", "v_mfma_f32_32x32x16_f16", "v_pk_fma_f32", "ds_read_b128", "v_fma_f32", "v_mfma_f32_32x32x2_f32", "f16 MFMA + v_pk_fma + ds_read",
                                  "v_mfma_f32_16x16x32_f16", "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_32x32x8_f16", "v_mfma_f32_16x16x16_f16", "v_mfma_i32_32x32x32_i8"};

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  unsigned long long* d;
  if (hipMalloc(&d, 128) != hipSuccess) return 1;
  const int iters = 1000, blocks = 1024;
  const double per_group = (double)blocks * 256 * iters / 4;
  printf("%d probe lane-iterations per 16-lane group and cell; cells: wrong LOW results in lanes 0-15 / 16-31 / 32-47 / 48-63 | wrong HIGH results\n", (int)per_group);
  auto cell = [&](int mode, int partner) {
    unsigned long long h[16];
    (void)hipMemset(d, 0, 128);
    k<<<blocks, 512>>>(mode, partner, d, iters);
    (void)hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
    printf("  %-34s %llu/%llu/%llu/%llu | %llu/%llu/%llu/%llu", kPartners[partner], h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7] & ~0ull);
    if (h[8]) {
      float got, want; unsigned u = (unsigned)(h[9] >> 32), v = (unsigned)h[9];
      memcpy(&got, &u, 4), memcpy(&want, &v, 4);
      printf("   e.g. low result %.7g instead of %.7g", got, want);
    }
    printf("\n");
  };
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  // 1. every selection of v_pk_mul_f32 (and the other packed instructions) next to the full partner load
  for (int m = 0; m < 27; ++m) {
    if (only >= 0 && only != m) continue;
    printf("%s\n", kNames[m]);
    cell(m, 6);
  }
  // 2. which partner activity triggers it (the form the vectoriser emitted: v_pk_mul_f32 op_sel:[0,1])
  if (only < 0) {
    printf("--- v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,1] against each partner activity\n");
    for (int p = 0; p < 12; ++p) cell(7, p);
    printf("--- the probe wave's OWN v_mfma_f32_32x32x16_f16, d idle cycles, then v_pk_mul_f32 op_sel:[0,1]; partner waves idle\n");
    for (int dd : {0, 4, 8, 16, 24, 32, 48}) {
      printf("  d = %2d:", dd);
      cell(100 + dd, 0);
    }
  }
  return 0;
}
