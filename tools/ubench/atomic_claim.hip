// What does a "claim the next unit of work" atomic cost on MI355X?  2 048 waves (512 work-groups x 4, the persistent grids of this
// repository) each make M dependent claims (lane 0: atomic add with return, then readfirstlane) on counters spread over A addresses,
// at agent scope (what atomicAdd means: coherent across the 8 XCDs, performed beyond the XCD's L2) and at work-group scope (performed
// in the XCD's own L2; only meaningful when all users of a counter sit on one XCD -- here counter = XCC_ID based).
//   hipcc --offload-arch=gfx950 -O3 -o atomic_claim atomic_claim.hip && ./atomic_claim
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
  return v;
}

template <int SCOPE>   // 0 agent, 1 workgroup (L2-local), by XCC_ID
__global__ __launch_bounds__(256) void claim_kernel(unsigned* counters, int n_addr, int claims, unsigned* sink, unsigned* xcc_seen) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const unsigned xcc = xcc_id();
  unsigned* c = SCOPE == 1 ? counters + 16 * (xcc + 8 * (wave % (n_addr >= 8 ? n_addr / 8 : 1))) : counters + 16 * (wave % n_addr);
  unsigned acc = 0;
  for (int i = 0; i < claims; ++i) {
    unsigned v = 0;
    if (lane == 0) {
      if (SCOPE == 0) v = atomicAdd(c, 1u);
      else v = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    acc += (unsigned)__builtin_amdgcn_readfirstlane((int)v);
  }
  if (lane == 0) sink[wave] = acc;
  if (threadIdx.x == 0) xcc_seen[blockIdx.x] = xcc;
}

int main() {
  const int grid = 512, waves = grid * 4;
  unsigned *counters, *sink, *xs;
  hipMalloc(&counters, 16 * 4 * 4096);
  hipMalloc(&sink, waves * 4);
  hipMalloc(&xs, grid * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int scope = 0; scope < 2; ++scope)
    for (int n_addr : {1, 8, 64, 512, 2048}) {
      if (scope == 1 && n_addr < 8) continue;
      for (int claims : {15, 60}) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
          hipMemset(counters, 0, 16 * 4 * 4096);
          hipDeviceSynchronize();
          hipEventRecord(e0);
          if (scope == 0) claim_kernel<0><<<grid, 256>>>(counters, n_addr, claims, sink, xs);
          else claim_kernel<1><<<grid, 256>>>(counters, n_addr, claims, sink, xs);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          if (ms < best) best = ms;
        }
        // check: the counters must add up to waves * claims (work-group scope: only true if every user of a counter shares an L2)
        std::vector<unsigned> h(16 * 4096);
        hipMemcpy(h.data(), counters, 16 * 4 * 4096, hipMemcpyDeviceToHost);
        unsigned long total = 0;
        for (int i = 0; i < 4096; ++i) total += h[16 * i];
        printf("scope %-9s addresses %4d claims/wave %2d : %8.2f us  (%6.1f ns per claim overall, %7.1f ns per claim and address)  counted %lu of %d\n",
               scope ? "workgroup" : "agent", n_addr, claims, best * 1e3, best * 1e6 / (waves * claims), best * 1e6 / (waves * claims) * n_addr, total,
               waves * claims);
      }
    }
  std::vector<unsigned> hx(grid);
  hipMemcpy(hx.data(), xs, grid * 4, hipMemcpyDeviceToHost);
  int mism = 0;
  for (int b = 0; b < grid; ++b) mism += (hx[b] != (unsigned)(b & 7));
  printf("XCC_ID == blockIdx %% 8 for %d of %d work-groups\n", grid - mism, grid);
  return 0;
}
