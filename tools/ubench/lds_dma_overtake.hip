// Can the LDS write of a global_load_lds_dwordx4 (LDS-DMA) overtake a ds_read_b128 of the SAME wave that was issued BEFORE it and is
// still waiting in the LDS queue?  This is the hazard behind the run-to-run differences of the gather ring's "late" order on the RE10K
// shapes (bts_render_kernel.h: gl_issue, profiles/r03_experiments/r03i - r03k): the program order  ds_read(slot) ... global_load_lds(slot)  is kept,
// yet nothing makes the DMA wait for the read to RETURN -- lgkmcnt counts the read, vmcnt the DMA, and the two travel separately.
//
// One wave per work-group plays the ring: it fills a 4 KB slot with pattern A (ds_write, drained), then in ONE asm block
//     DELAY x ds_read_b128 with a 64-way bank conflict     (stands for the eight waves' weight reads that keep the queue deep)
//     4 x ds_read_b128 of the slot                         (the rows of the block that lives there)
//     [s_waitcnt lgkmcnt(0)]                               (mode "guarded": what the four dependency operands of gl_issue enforce)
//     4 x global_load_lds_dwordx4 into the slot            (pattern B from a buffer that sits in L2; the instruction offset moves the
//                                                           global AND the LDS address, M0 stays)
//     s_waitcnt vmcnt(0) lgkmcnt(0)
// and counts the 16-byte pieces that came back as B.  The other waves of the work-group (HAMMER = 1) add their own conflicted reads.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_dma_overtake.hip -o lds_dma_overtake && ./lds_dma_overtake
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int DELAY, bool GUARDED>
__global__ __launch_bounds__(256) void overtake_kernel(const uint4* __restrict__ B, unsigned* __restrict__ overtaken, unsigned* __restrict__ other,
                                                       int iters, int hammer) {
  __shared__ __attribute__((aligned(128))) uint4 slot[256];      // 4 KB: four DMA instructions of 1 KB
  __shared__ __attribute__((aligned(128))) uint4 conflict[1024];  // lane L reads conflict[16 L]: every lane in the same banks
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 1024; i += 256) conflict[i] = make_uint4(i, i, i, i);
  __syncthreads();
  unsigned sink = 0;
  if (wave != 0) {
    if (hammer) {
      for (int it = 0; it < iters * 4; ++it) {
        const uint4 v = conflict[16 * lane + (it & 15)];
        sink += v.x;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (sink == 0xFFFFFFFFu) other[0] = sink;
    return;
  }
  const unsigned slot_m0 = (unsigned)(unsigned long)slot;
  const unsigned rd = slot_m0 + 16u * (unsigned)lane;
  const unsigned cf = (unsigned)(unsigned long)conflict + 256u * (unsigned)lane;
  const unsigned goff = 16u * (unsigned)lane;
  unsigned n_b = 0, n_other = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned tagA = 0x10000000u + (unsigned)it * 4096u;
#pragma unroll
    for (int q = 0; q < 4; ++q) slot[64 * q + lane] = make_uint4(tagA + 64 * q + lane, 1, 2, 3);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    uint4 r0, r1, r2, r3, d0, d1, d2, d3, d4, d5, d6, d7;
    if constexpr (DELAY == 8) {
      asm volatile(
          "ds_read_b128 %4, %12\n\tds_read_b128 %5, %12 offset:16\n\tds_read_b128 %6, %12 offset:32\n\tds_read_b128 %7, %12 offset:48\n\t"
          "ds_read_b128 %8, %12 offset:64\n\tds_read_b128 %9, %12 offset:80\n\tds_read_b128 %10, %12 offset:96\n\tds_read_b128 %11, %12 offset:112\n\t"
          "ds_read_b128 %0, %13\n\tds_read_b128 %1, %13 offset:1024\n\tds_read_b128 %2, %13 offset:2048\n\tds_read_b128 %3, %13 offset:3072\n\t"
          "s_waitcnt lgkmcnt(%17)\n\t"
          "s_mov_b32 m0, %14\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %15, %16\n\t"
          "global_load_lds_dwordx4 %15, %16 offset:1024\n\t"
          "global_load_lds_dwordx4 %15, %16 offset:2048\n\t"
          "global_load_lds_dwordx4 %15, %16 offset:3072\n\t"
          "s_waitcnt vmcnt(0) lgkmcnt(0)"
          : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)
          : "v"(cf), "v"(rd), "s"(slot_m0), "v"(goff), "s"(B), "n"(GUARDED ? 0 : 15)
          : "memory", "m0");
      sink += d0.x + d1.x + d2.x + d3.x + d4.x + d5.x + d6.x + d7.x;
    } else {
      asm volatile(
          "ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\t"
          "s_waitcnt lgkmcnt(%8)\n\t"
          "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %7\n\t"
          "global_load_lds_dwordx4 %6, %7 offset:1024\n\t"
          "global_load_lds_dwordx4 %6, %7 offset:2048\n\t"
          "global_load_lds_dwordx4 %6, %7 offset:3072\n\t"
          "s_waitcnt vmcnt(0) lgkmcnt(0)"
          : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
          : "v"(rd), "s"(slot_m0), "v"(goff), "s"(B), "n"(GUARDED ? 0 : 15)
          : "memory", "m0");
    }
    const uint4 r[4] = {r0, r1, r2, r3};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool isA = r[q].x == tagA + 64 * q + lane && r[q].y == 1 && r[q].z == 2 && r[q].w == 3;
      const bool isB = r[q].x == 0xB0000000u + 64 * q + lane && r[q].y == 0xB1 && r[q].z == 0xB2 && r[q].w == 0xB3;
      n_b += isB, n_other += !isA && !isB;
    }
    // what the DMA left in the slot must be B in any case
#pragma unroll
    for (int q = 0; q < 4; ++q) n_other += slot[64 * q + lane].x != 0xB0000000u + 64 * q + lane;
  }
  if (sink == 0xFFFFFFFFu) other[1] = sink;
  atomicAdd(overtaken, n_b), atomicAdd(other, n_other);
}

template <int DELAY, bool GUARDED>
static void run(const uint4* dB, unsigned* dCnt, int grid, int iters, int hammer) {
  hipMemset(dCnt, 0, 8);
  overtake_kernel<DELAY, GUARDED><<<grid, 256>>>(dB, dCnt, dCnt + 1, iters, hammer);
  if (hipDeviceSynchronize() != hipSuccess) {
    printf("launch failed: %s\n", hipGetErrorString(hipGetLastError()));
    exit(1);
  }
  unsigned h[2];
  hipMemcpy(h, dCnt, 8, hipMemcpyDeviceToHost);
  printf("queue delay %d conflicted reads, other waves hammering %d, %-9s: %10u of %ld row pieces came back as the NEW block, %u garbage\n", DELAY, hammer,
         GUARDED ? "guarded" : "unguarded", h[0], (long)grid * iters * 4 * 64, h[1]);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  int cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) == hipSuccess) cus = prop.multiProcessorCount;
  std::vector<uint4> hB(256);
  for (int i = 0; i < 256; ++i) hB[i] = make_uint4(0xB0000000u + i, 0xB1, 0xB2, 0xB3);
  uint4* dB;
  unsigned* dCnt;
  hipMalloc(&dB, 4096), hipMalloc(&dCnt, 8);
  hipMemcpy(dB, hB.data(), 4096, hipMemcpyHostToDevice);
  for (int hammer = 0; hammer <= 1; ++hammer) {
    run<0, false>(dB, dCnt, cus * 2, iters, hammer);
    run<0, true>(dB, dCnt, cus * 2, iters, hammer);
    run<8, false>(dB, dCnt, cus * 2, iters, hammer);
    run<8, true>(dB, dCnt, cus * 2, iters, hammer);
  }
  return 0;
}
