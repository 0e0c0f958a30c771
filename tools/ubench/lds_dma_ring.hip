// Does the gather ring of render_kernel_p (three 4 KB LDS slots per wave filled by global_load_lds_dwordx4, read back as ds_read_b128,
// bts_render_kernel.h: GatherLds / gl_*) deliver the right rows under BOTH orders of its steady state?
//   late order   (default build):          blend block T, issue block T + 3 into T's slot, THEN request block T + 1 from LDS
//   early order  (-DBTS_GL_FETCH_EARLY):   request block T + 1 from LDS, blend block T, issue block T + 3 into T's slot
// This stand-alone driver runs the VERY SAME device functions (it includes the kernel header) with synthetic taps, unit blend weights
// and a feature map whose rows are known, so that every accumulator value can be checked against the host -- not only compared
// between runs -- with matrix-pipe traffic between the steps (MFMA = 1) and one or two waves per SIMD.
// History: rounds 2 - 3 used it to hunt the run-to-run differences of the late order on the RE10K shapes.  It stayed clean in every
// configuration: the hazard needs a deep LDS queue AND a compiler schedule that reorders the four row reads -- see
// lds_dma_overtake.hip, which shows the hardware behaviour in isolation, and gl_issue, which now closes it by construction.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -ffp-contract=off [-DBTS_GL_FETCH_EARLY] -DRING_HD=32 \
//         tools/ubench/lds_dma_ring.hip -o lds_dma_ring && ./lds_dma_ring
#define BTS_NO_LAUNCH_GLUE
#include "../../behindthescenes_amd/csrc/bts_render_kernel.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

#ifndef RING_HD
#define RING_HD 32
#endif
using namespace bts;

namespace bts {
void set_error(const char*, const char*, long, long, long) {}
}

__host__ __device__ inline unsigned mix(unsigned x) {
  x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
  return x;
}

template <int HD, int T, int NBLK>
__device__ __forceinline__ void consume_all(f32x16 (&acc)[HD / 32][2], const GatherLds& gl, GRows& rows, const float4* G, const float (&wq)[2][4],
                                            unsigned (&off_next)[4], f32x16& busy, h8 ma, h8 mb, int with_mfma) {
  if constexpr (T < NBLK) {
    if (with_mfma && (T & 3) == 0) {   // what an encoding region puts between the steps: a burst of wide MFMAs + some VALU
#pragma unroll
      for (int i = 0; i < 6; ++i) busy = __builtin_amdgcn_mfma_f32_32x32x16_f16(ma, mb, busy, 0, 0, 0);
    }
    gl_consume<HD, T>(acc, gl, rows, G, wq, off_next);
    consume_all<HD, T + 1, NBLK>(acc, gl, rows, G, wq, off_next, busy, ma, mb, with_mfma);
  }
}

template <int HD>
__global__ __launch_bounds__(256, 2) void ring_kernel(const float4* __restrict__ G, int Wt, int Ht, float* __restrict__ out, int rays_per_wave,
                                                      int with_mfma, unsigned seed) {
  constexpr int HT = HD / 32;
  constexpr int NBLK = 8 * HT;
  __shared__ float pad[7000];   // the weights' share of LDS in the real kernel (two work-groups per CU either way)
  extern __shared__ __attribute__((aligned(128))) char gather_lds[];
  const int lane = threadIdx.x & 63, h0 = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (threadIdx.x < 64) pad[threadIdx.x * 100] = (float)threadIdx.x;
  __syncthreads();
  GatherLds gl;
  {
    char* base = gather_lds + wave * kGatherLdsPerWave;
    gl.ring = base;
    gl.ring_m0 = (unsigned)(unsigned long)base;
    gl.tab = reinterpret_cast<unsigned*>(base + 3 * 4096);
    gl.m = lane >> 3;
    gl.piece16 = 16u * (unsigned)(((lane & 7) + (lane >> 4)) & 7);
    gl.piece16x = gl.piece16 ^ 64u;
    const int col = lane & 31;
#pragma unroll
    for (int q = 0; q < 4; ++q) gl.rd[q] = (unsigned)(col * 128 + ((4 * h0 + q - (col >> 1)) & 7) * 16);
  }
  f32x16 busy = zero_acc();
  const h8 ma = {(_Float16)1, (_Float16)0.5f, (_Float16)0.25f, (_Float16)2, (_Float16)1, (_Float16)1, (_Float16)0.125f, (_Float16)3};
  const h8 mb = {(_Float16)0.5f, (_Float16)1, (_Float16)1, (_Float16)0.25f, (_Float16)2, (_Float16)0.5f, (_Float16)1, (_Float16)1};
  const long wave_id = (long)blockIdx.x * 4 + wave;
  for (int r = 0; r < rays_per_wave; ++r) {
    const long ray = wave_id * rays_per_wave + r;
    // this lane's sample: a pseudo-random texel with its three neighbours (o11 = o10 + o01 - o00, as make_taps)
    const unsigned hsh = mix((unsigned)ray * 64u + (unsigned)lane + seed);
    const int x0 = (int)(hsh % (unsigned)(Wt - 1)), y0 = (int)((hsh >> 12) % (unsigned)(Ht - 1));
    const unsigned o00 = (unsigned)(y0 * Wt + x0), o01 = o00 + 1, o10 = o00 + (unsigned)Wt;
    float wq[2][4];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
      for (int t = 0; t < 4; ++t) wq[pt][t] = 1.0f;
    unsigned off_next[4];
    GRows rows;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    gl.tab[lane * 3 + 0] = o00 * (HD * 4u), gl.tab[lane * 3 + 1] = o01 * (HD * 4u), gl.tab[lane * 3 + 2] = o10 * (HD * 4u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    gl_prologue<HD>(gl, rows, G, off_next);
    f32x16 acc[HT][2];
#pragma unroll
    for (int ht = 0; ht < HT; ++ht) acc[ht][0] = zero_acc(), acc[ht][1] = zero_acc();
    gl_fetch<HD, 0, 2>(gl, rows);
    consume_all<HD, 0, NBLK>(acc, gl, rows, G, wq, off_next, busy, ma, mb, with_mfma);
    // out[ray][pt][sample col][channel ht*32 + 16 h + i]: this lane's 16 accumulator rows of each tile
#pragma unroll
    for (int ht = 0; ht < HT; ++ht)
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        float* dst = out + ((ray * 2 + pt) * 32 + (lane & 31)) * HD + ht * 32 + 16 * h0;
#pragma unroll
        for (int i = 0; i < 16; ++i) dst[i] = acc[ht][pt][i];
      }
  }
  if (busy[3] == 12345.0f) out[0] = pad[lane * 100];
}

int main(int argc, char** argv) {
  constexpr int HD = RING_HD;
  const int Wt = 384, Ht = 256;                 // the RE10K map
  const int launches = argc > 1 ? atoi(argv[1]) : 40;
  int cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) == hipSuccess) cus = prop.multiProcessorCount;
  const int rays_per_wave = 12;
  std::vector<float> hG((size_t)Wt * Ht * HD);
  for (size_t i = 0; i < hG.size(); ++i) hG[i] = (float)(mix((unsigned)i * 2654435761u) & 0xFFFF) * (1.0f / 64.0f) - 300.0f;
  float* dG;
  hipMalloc(&dG, hG.size() * 4);
  hipMemcpy(dG, hG.data(), hG.size() * 4, hipMemcpyHostToDevice);
  const int dyn = 4 * kGatherLdsPerWave;
  hipFuncSetAttribute(reinterpret_cast<const void*>(ring_kernel<HD>), hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
#if defined(BTS_GL_FETCH_EARLY)
  const char* order = "early";
#else
  const char* order = "late";
#endif
  const char* drain = "";
  for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu)
    for (int with_mfma = 0; with_mfma <= 1; ++with_mfma) {
      const int grid = cus * wgs_per_cu;
      const long rays = (long)grid * 4 * rays_per_wave;
      const size_t n_out = (size_t)rays * 64 * HD;
      float* dOut;
      hipMalloc(&dOut, n_out * 4);
      std::vector<float> got(n_out), want(n_out);
      long bad_rays_total = 0, bad_vals_total = 0, bad_launches = 0;
      for (int L = 0; L < launches; ++L) {
        const unsigned seed = 1000u * (unsigned)L + 17u;
        hipMemset(dOut, 0xFF, n_out * 4);
        ring_kernel<HD><<<grid, 256, dyn, 0>>>(reinterpret_cast<const float4*>(dG), Wt, Ht, dOut, rays_per_wave, with_mfma, seed);
        if (hipDeviceSynchronize() != hipSuccess) {
          printf("launch failed: %s\n", hipGetErrorString(hipGetLastError()));
          return 1;
        }
        hipMemcpy(got.data(), dOut, n_out * 4, hipMemcpyDeviceToHost);
        long bad_rays = 0, bad_vals = 0;
        for (long ray = 0; ray < rays; ++ray) {
          bool bad = false;
          for (int s = 0; s < 64; ++s) {
            const unsigned hsh = mix((unsigned)ray * 64u + (unsigned)s + seed);
            const int x0 = (int)(hsh % (unsigned)(Wt - 1)), y0 = (int)((hsh >> 12) % (unsigned)(Ht - 1));
            const size_t o[4] = {(size_t)y0 * Wt + x0, (size_t)y0 * Wt + x0 + 1, (size_t)(y0 + 1) * Wt + x0, (size_t)(y0 + 1) * Wt + x0 + 1};
            const float* g = &got[((size_t)(ray * 2 + s / 32) * 32 + s % 32) * HD];
            for (int c = 0; c < HD; ++c) {
              float a = 0.0f;
              for (int t = 0; t < 4; ++t) a = fmaf(hG[o[t] * HD + c], 1.0f, a);
              if (g[c] != a) ++bad_vals, bad = true;
            }
          }
          bad_rays += bad;
        }
        bad_rays_total += bad_rays, bad_vals_total += bad_vals, bad_launches += bad_rays > 0;
      }
      printf("d_hidden %d  order %-7s%s  %d work-group(s) / CU  matrix-pipe traffic %d:  %ld of %d launches wrong, %ld wrong rays of %ld, %ld wrong values\n",
             HD, order, drain, wgs_per_cu, with_mfma, bad_launches, launches, bad_rays_total, rays * launches, bad_vals_total);
      hipFree(dOut);
    }
  return 0;
}
