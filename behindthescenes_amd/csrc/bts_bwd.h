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
};

__device__ __forceinline__ void atomic_add_f32(float* p, float v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace bts
