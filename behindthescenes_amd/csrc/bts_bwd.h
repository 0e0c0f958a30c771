// Shared between the backward's translation units (bts_bwd.hip: lane = ray pass + scatter; bts_bwd_rows.hip: lane = sample passes).
#pragma once
#include "bts_field_kernel.h"

namespace bts {

struct BwdParams {
  FwdParams f;            // field + rays (+ sigma_raw, trans as inputs)
  const float* g_rgb;     // (B, nv*3)
  const float* g_depth;   // (B)
  const float* g_weights; // (B, K)
  const float* g_alphas;  // (B, K)
  float* d_proj;          // (n,H,W,HD)
  float* d_mlp;           // packed
  float* d_empty_proj;    // (HD)
  float* gh_ws;           // lane = ray path: (groups, K, 64, HD) g_h rows for the dG scatter pass, or null: scatter with direct atomics
  float* gs_ws;           // lane = sample path (bts_bwd_rows.hip): (n*Bp, K) gradient at the pre-softplus density
  unsigned* mask_ws;      //                    (n*Bp, HD/32, K) relu gates of lin_in's output per sample, one bit per channel
  uint2* pmask_ws;        //                    (n*Bp, HD) the same gates per channel, one bit per sample of the ray
  unsigned char* tiles;   // (n, tiles_per_img) dirty flags of d_proj's 64-texel tiles (BtsRenderGrads.d_proj_tiles), or null
  int tiles_per_img;
  float* flush_ws;        // kFlushSlots x (40 x HD) floats, zeroed by the launcher: pass C's dW_pe partial sums (dwpe_flush below)
};

// Pass C ends with every work-group adding its 40 x HD partial sums of dW_pe / db_in into the SAME 40 x HD addresses of d_mlp: ~500-770
// float atomics per address, issued by work-groups that all finish within microseconds of each other.  Measured (profiles/r04j, flush
// ablated): 0.16 of dwpe_kernel's 0.33 ms and 0.09 of dwpe_rows_kernel's 0.18 ms were these atomics queueing at L2.  The work-groups
// therefore add into one of kFlushSlots copies (work-group b -> slot b % kFlushSlots: 1/8 of the queue per address), and a 40 x HD
// thread kernel (dwpe_reduce_kernel) folds the copies into d_mlp.
constexpr int kFlushSlots = 8;
constexpr int kFlushRows = kPeDim + 1;   // 40 encoding inputs incl. the bias row

__device__ __forceinline__ void atomic_add_f32(float* p, float v) {

  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the parameter-gradient flushes at the end of the backward kernels (every work-group -> the same few thousand addresses)
__device__ __forceinline__ void flush_add_f32(float* p, float v) {
#ifndef BTS_ABL_NOFLUSH   // timing ablation: what do the same-address atomics of ~500 work-groups cost?
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  if (v == 1.2345e-30f) *p = v;
#endif
}

}  // namespace bts
