// Does s_waitcnt vmcnt(N) on MI355X release a wave before the LAST 16 lanes of a returning global load are visible to the very next
// VALU instruction?  Observed in bts::render_kernel_p (round 2): a v_pk_mul_f32 issued right behind `s_waitcnt vmcnt(3)` saw, in
// lanes 48-63 only and timing-dependently, the OLD contents of the load's destination registers (which were the load's own address
// registers) -- every other intermediate, including the loaded texels read a few instructions later, was intact.
// This reproducer issues the same instruction shapes with explicit registers through inline asm and checks the consumer's result
// against the value recomputed from the registers a few dozen cycles later.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/vmem_return_race.hip -o vmem_return_race && ./vmem_return_race
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE 0: dwordx3, dst overlaps the 64-bit address pair, consumer v_pk_mul_f32 op_sel:[0,1]      (the failing shape)
// MODE 1: same, consumer plain v_mul_f32
// MODE 2: dwordx3, dst disjoint from the address, consumer v_pk_mul_f32
// MODE 3: dwordx4, dst overlaps the address, consumer v_pk_mul_f32
// MODE 4: four loads + vmcnt(3),(2),(1),(0) + four pk_mul exactly as in the kernel
template <int MODE>
__global__ __launch_bounds__(512) void k(const float* __restrict__ buf, unsigned n_vec4, unsigned long long* __restrict__ out, int iters) {
  const int lane = threadIdx.x & 63;
  __shared__ volatile int done;
  __shared__ float lds_w[4096];
  if (threadIdx.x == 0) done = 0;
  for (int i = threadIdx.x; i < 4096; i += 512) lds_w[i] = 0.001f * i;
  __syncthreads();
  if (threadIdx.x >= 256) {   // partner waves (one per SIMD): the renderer's MFMA phase -- f16 MFMAs, packed FMAs, LDS reads -- until the probes finish
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 acc;
    for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
    h8 a, b;
    for (int e = 0; e < 8; ++e) a[e] = (_Float16)(0.01f * (lane + e)), b[e] = (_Float16)(0.02f * (lane - e));
    f32x2 p = {1.0f, 2.0f}, q2 = {0.5f, 0.25f};
    int guard = 0;
    while (!done && guard < (1 << 22)) {
      for (int r = 0; r < 16; ++r) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        const float4 wv = *reinterpret_cast<const float4*>(&lds_w[((lane + r + guard) & 1023) * 4]);
        p = __builtin_elementwise_fma(p, q2, (f32x2){wv.x, wv.y});
        q2 = __builtin_elementwise_fma(q2, p, (f32x2){wv.z, wv.w});
        acc[r] += p[0];
      }
      guard += 16;
    }
    if (acc[3] == 12345.0f || q2[1] == 777.0f) out[3] = 1;
    return;
  }
  unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  unsigned long long bad = 0, bad_hi = 0, stale_addr = 0;
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    // neighbouring lanes mostly hit neighbouring texels (like the renderer), sometimes far away
    const unsigned idx = ((s >> 7) % 4096u == 0 ? (s >> 3) : (blockIdx.x * 977u + it * 13u + lane / 4)) % n_vec4;
    const unsigned long addr = (unsigned long)(buf + 4ul * idx);
    const f32x2 w = {1.0f + 0.001f * lane, 2.0f + 0.003f * (it & 15)};
    unsigned long addr_io = addr;
    float r0, r1, x, y;
    if constexpr (MODE == 0) {
      asm volatile("global_load_dwordx3 v[8:10], v[8:9], off\n s_waitcnt vmcnt(0)\n v_pk_mul_f32 v[2:3], v[8:9], v[28:29] op_sel:[0,1]\n"
                   "s_nop 15\n s_nop 15\n v_mov_b32 %0, v2\n v_mov_b32 %1, v3\n v_mov_b32 %2, v8\n v_mov_b32 %3, v9"
                   : "=&v"(r0), "=&v"(r1), "=&v"(x), "=&v"(y), "+{v[8:9]}"(addr_io) : "{v[28:29]}"(w) : "v2", "v3", "v10", "memory");
    } else if constexpr (MODE == 1) {
      asm volatile("global_load_dwordx3 v[8:10], v[8:9], off\n s_waitcnt vmcnt(0)\n v_mul_f32 v2, v8, v29\n v_mul_f32 v3, v9, v29\n"
                   "s_nop 15\n s_nop 15\n v_mov_b32 %0, v2\n v_mov_b32 %1, v3\n v_mov_b32 %2, v8\n v_mov_b32 %3, v9"
                   : "=&v"(r0), "=&v"(r1), "=&v"(x), "=&v"(y), "+{v[8:9]}"(addr_io) : "{v[28:29]}"(w) : "v2", "v3", "v10", "memory");
    } else if constexpr (MODE == 2) {
      asm volatile("v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n global_load_dwordx3 v[12:14], v[8:9], off\n s_waitcnt vmcnt(0)\n v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[0,1]\n"
                   "s_nop 15\n s_nop 15\n v_mov_b32 %0, v2\n v_mov_b32 %1, v3\n v_mov_b32 %2, v12\n v_mov_b32 %3, v13"
                   : "=&v"(r0), "=&v"(r1), "=&v"(x), "=&v"(y) : "{v[8:9]}"(addr), "{v[28:29]}"(w) : "v2", "v3", "v12", "v13", "v14", "memory");
    } else if constexpr (MODE == 3) {
      asm volatile("global_load_dwordx4 v[8:11], v[8:9], off\n s_waitcnt vmcnt(0)\n v_pk_mul_f32 v[2:3], v[8:9], v[28:29] op_sel:[0,1]\n"
                   "s_nop 15\n s_nop 15\n v_mov_b32 %0, v2\n v_mov_b32 %1, v3\n v_mov_b32 %2, v8\n v_mov_b32 %3, v9"
                   : "=&v"(r0), "=&v"(r1), "=&v"(x), "=&v"(y), "+{v[8:9]}"(addr_io) : "{v[28:29]}"(w) : "v2", "v3", "v10", "v11", "memory");
    } else if constexpr (MODE >= 5 && MODE <= 12) {
      // single dwordx3 load into v[12:14] (disjoint from the address), different consumers / delays
      float e0v = 0.f, e1v = 0.f;
#define LOAD_ "v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n global_load_dwordx3 v[12:14], v[8:9], off\n s_waitcnt vmcnt(0)\n"
#define TAIL_ "s_nop 15\n s_nop 15\n v_mov_b32 %0, v2\n v_mov_b32 %1, v3\n v_mov_b32 %2, v12\n v_mov_b32 %3, v13"
#define OPS_ : "=&v"(r0), "=&v"(r1), "=&v"(x), "=&v"(y), "+{v[8:9]}"(addr_io) : "{v[28:29]}"(w) : "v2", "v3", "v12", "v13", "v14", "memory"
      if constexpr (MODE == 5) {   // no load at all: the operands come from VALU moves
        asm volatile("global_load_dwordx3 v[12:14], v[8:9], off\n s_waitcnt vmcnt(0)\n s_nop 15\n v_mov_b32 v12, v12\n v_mov_b32 v13, v13\n s_nop 15\n"
                     "v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[0,1]\n" TAIL_ OPS_);
      } else if constexpr (MODE == 6) {   // 8 idle cycles between the wait and the consumer
        asm volatile(LOAD_ "s_nop 7\n v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[0,1]\n" TAIL_ OPS_);
      } else if constexpr (MODE == 7) {   // default op_sel: lo*lo, hi*hi
        asm volatile(LOAD_ "v_pk_mul_f32 v[2:3], v[12:13], v[28:29]\n" TAIL_ OPS_);
      } else if constexpr (MODE == 8) {   // the gather blend's shape: v_pk_fma_f32 data, weight(lo broadcast), acc
        asm volatile(LOAD_ "v_mov_b32 v2, 0\n v_mov_b32 v3, 0\n" LOAD_ "v_pk_fma_f32 v[2:3], v[12:13], v[28:29], v[2:3] op_sel_hi:[1,0,1]\n" TAIL_ OPS_);
      } else if constexpr (MODE == 9) {   // one ordinary VALU instruction between the wait and the consumer
        asm volatile(LOAD_ "v_mov_b32 v4, v28\n v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[0,1]\n" TAIL_ OPS_ , "v4");
      } else if constexpr (MODE == 10) {  // 1 idle cycle
        asm volatile(LOAD_ "s_nop 0\n v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[0,1]\n" TAIL_ OPS_);
      } else if constexpr (MODE == 11) {  // 4 idle cycles
        asm volatile(LOAD_ "s_nop 3\n v_pk_mul_f32 v[2:3], v[12:13], v[28:29] op_sel:[0,1]\n" TAIL_ OPS_);
      } else {                            // loaded data as src1
        asm volatile(LOAD_ "v_pk_mul_f32 v[2:3], v[28:29], v[12:13] op_sel:[1,0]\n" TAIL_ OPS_);
      }
      if (MODE == 7) e0v = x * w[0], e1v = y * w[1];
      else if (MODE == 8) e0v = __builtin_fmaf(x, w[0], 0.0f), e1v = __builtin_fmaf(y, w[0], 0.0f);
      else e0v = x * w[1], e1v = y * w[1];
      if (__float_as_uint(r0) != __float_as_uint(e0v) || __float_as_uint(r1) != __float_as_uint(e1v)) {
        if (!bad && lane >= 48 && atomicAdd(out + 4, 1ull) == 0) {
          out[5] = ((unsigned long long)__float_as_uint(r0) << 32) | __float_as_uint(r1);
          out[6] = ((unsigned long long)__float_as_uint(x) << 32) | __float_as_uint(y);
          out[7] = ((unsigned long long)__float_as_uint(w[0]) << 32) | __float_as_uint(w[1]);
        }
        bad++, bad_hi += lane >= 48;
      }
      if (x != buf[4ul * idx] || y != buf[4ul * idx + 1]) bad += 1ull << 32;
      continue;
    } else {
      unsigned long a1 = (unsigned long)(buf + 4ul * ((idx + 1) % n_vec4)), a2 = (unsigned long)(buf + 4ul * ((idx + 160) % n_vec4)),
                          a3 = (unsigned long)(buf + 4ul * ((idx + 161) % n_vec4));
      asm volatile("global_load_dwordx3 v[8:10], v[8:9], off\n s_nop 0\n global_load_dwordx3 v[12:14], v[12:13], off\n s_nop 0\n"
                   "global_load_dwordx3 v[16:18], v[16:17], off\n s_nop 0\n global_load_dwordx3 v[20:22], v[20:21], off\n"
                   "s_waitcnt vmcnt(3)\n v_pk_mul_f32 v[2:3], v[8:9], v[28:29] op_sel:[0,1]\n"
                   "s_waitcnt vmcnt(2)\n v_pk_mul_f32 v[32:33], v[12:13], v[28:29] op_sel_hi:[1,0]\n"
                   "s_waitcnt vmcnt(1)\n v_pk_mul_f32 v[34:35], v[16:17], v[28:29] op_sel:[0,1]\n"
                   "s_waitcnt vmcnt(0)\n v_pk_mul_f32 v[36:37], v[20:21], v[28:29] op_sel_hi:[1,0]\n"
                   "s_nop 15\n s_nop 15\n"
                   // fold the four products' checks into (r0, x) / (r1, y): r0 = p0.lo + p2.lo, x = v8 + v16 scaled the same way below
                   "v_mul_f32 v4, v8, v29\n v_mul_f32 v5, v12, v28\n v_mul_f32 v6, v16, v29\n v_mul_f32 v7, v20, v28\n"
                   "v_cmp_neq_f32 vcc, v2, v4\n v_cndmask_b32 %0, 0, 1, vcc\n v_cmp_neq_f32 vcc, v32, v5\n v_cndmask_b32 %1, 0, 1, vcc\n"
                   "v_cmp_neq_f32 vcc, v34, v6\n v_cndmask_b32 %2, 0, 1, vcc\n v_cmp_neq_f32 vcc, v36, v7\n v_cndmask_b32 %3, 0, 1, vcc"
                   : "=&v"(r0), "=&v"(r1), "=&v"(x), "=&v"(y), "+{v[8:9]}"(addr_io), "+{v[12:13]}"(a1), "+{v[16:17]}"(a2), "+{v[20:21]}"(a3)
                   : "{v[28:29]}"(w)
                   : "v2", "v3", "v4", "v5", "v6", "v7", "v10", "v14", "v18", "v22", "v32", "v33", "v34", "v35", "v36", "v37", "vcc", "memory");
      const unsigned m = __float_as_uint(r0) | __float_as_uint(r1) | __float_as_uint(x) | __float_as_uint(y);
      if (m) bad++, bad_hi += lane >= 48;
      continue;
    }
    const float e0 = x * w[1], e1 = y * w[1];
    if (__float_as_uint(r0) != __float_as_uint(e0) || __float_as_uint(r1) != __float_as_uint(e1)) {
      bad++, bad_hi += lane >= 48;
      // did the consumer see the address bits instead of the data?
      const float s0 = __uint_as_float((unsigned)addr) * w[1];
      stale_addr += __float_as_uint(r0) == __float_as_uint(s0);
    }
    if (x != buf[4ul * idx] || y != buf[4ul * idx + 1]) bad += 1ull << 32;   // the load itself must be right
  }
  if (bad) atomicAdd(out, bad), atomicAdd(out + 1, bad_hi), atomicAdd(out + 2, stale_addr);
  __builtin_amdgcn_s_waitcnt(0);
  done = 1;
}

template <int MODE>
void run(const char* name, const float* buf, unsigned n_vec4, unsigned long long* d) {
  hipMemset(d, 0, 64);
  const int iters = 2000, blocks = 2048;
  k<MODE><<<blocks, 512>>>(buf, n_vec4, d, iters);
  unsigned long long h[8];
  const hipError_t e = hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
  if (e != hipSuccess) printf("%s: %s\n", name, hipGetErrorString(e));
  printf("%-78s wrong consumer results %llu (lanes 48-63: %llu, = stale address bits: %llu), wrong loads %llu, of %llu\n", name,
         h[0] & 0xffffffffull, h[1], h[2], h[0] >> 32, (unsigned long long)blocks * 256 * iters);
  if (h[4]) {
    auto f = [](unsigned long long v, int hi) { unsigned u = hi ? (unsigned)(v >> 32) : (unsigned)v; float r; memcpy(&r, &u, 4); return r; };
    printf("    example (a lane >= 48): got (%.9g, %.9g) from data (%.9g, %.9g) and weights (%.9g, %.9g): got / weight = (%.9g, %.9g)\n", f(h[5], 1), f(h[5], 0),
           f(h[6], 1), f(h[6], 0), f(h[7], 1), f(h[7], 0), f(h[5], 1) / f(h[7], 0), f(h[5], 0) / f(h[7], 0));
  }
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  const unsigned n_vec4 = 16u << 20;   // 256 MB of rgb0 texels
  float* buf;
  unsigned long long* d;
  hipMalloc(&buf, 16ul * n_vec4), hipMalloc(&d, 64);
  float* h = (float*)malloc(16ul * n_vec4);
  for (unsigned long i = 0; i < 4ul * n_vec4; ++i) h[i] = 1.0f + (float)(i % 65521u) * 0.001f;
  hipMemcpy(buf, h, 16ul * n_vec4, hipMemcpyHostToDevice);
  if (only < 0 || only == 0) run<0>("dwordx3, dst overlaps address, consumer v_pk_mul_f32 op_sel:[0,1]", buf, n_vec4, d);
  if (only < 0 || only == 1) run<1>("dwordx3, dst overlaps address, consumer v_mul_f32", buf, n_vec4, d);
  if (only < 0 || only == 2) run<2>("dwordx3, dst disjoint from address, consumer v_pk_mul_f32", buf, n_vec4, d);
  if (only < 0 || only == 3) run<3>("dwordx4, dst overlaps address, consumer v_pk_mul_f32", buf, n_vec4, d);
  if (only < 0 || only == 4) run<4>("4 x dwordx3 + vmcnt(3..0) + 4 x v_pk_mul_f32 (kernel sequence)", buf, n_vec4, d);
  if (only < 0 || only == 5) run<5>("operands re-written by v_mov long after the load, then v_pk_mul_f32", buf, n_vec4, d);
  if (only < 0 || only == 10) run<10>("dwordx3, wait, s_nop 0, v_pk_mul_f32 op_sel:[0,1]", buf, n_vec4, d);
  if (only < 0 || only == 11) run<11>("dwordx3, wait, s_nop 3, v_pk_mul_f32 op_sel:[0,1]", buf, n_vec4, d);
  if (only < 0 || only == 6) run<6>("dwordx3, wait, s_nop 7, v_pk_mul_f32 op_sel:[0,1]", buf, n_vec4, d);
  if (only < 0 || only == 9) run<9>("dwordx3, wait, one v_mov_b32, v_pk_mul_f32 op_sel:[0,1]", buf, n_vec4, d);
  if (only < 0 || only == 7) run<7>("dwordx3, wait, v_pk_mul_f32 (default op_sel)", buf, n_vec4, d);
  if (only < 0 || only == 8) run<8>("dwordx3, wait, v_pk_fma_f32 data, weight, acc op_sel_hi:[1,0,1] (blend shape)", buf, n_vec4, d);
  if (only < 0 || only == 12) run<12>("dwordx3, wait, v_pk_mul_f32 weight, data op_sel:[1,0] (data as src1)", buf, n_vec4, d);
  return 0;
}
