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
  int tile_tw;            // tile_cols(H >> fs, W >> fs, cfg->tile_blocks): blocks per row of the 16 x 4 tile form, 0 = runs of 64 texels
#ifdef BTS_TICKS
  unsigned long long* ticks;   // diagnostic build: [waves][16] cycles per section (rows_kernel: tools/bwd_ticks.py; scatter_kernel behind them)
#endif
  float* flush_ws;        // kFlushSlots x (40 x HD) floats, zeroed by the launcher: pass C's dW_pe partial sums (dwpe_flush below)
  bool flush_clean;       // the caller already zeroed flush_ws on this stream (bts_train_step_bwd: one prep launch does every scale's)
};

// diagnostic build (-DBTS_TICKS, behindthescenes_amd/variants): s_memtime at the section borders of a kernel's iteration.  Reading the counter
// drains lgkmcnt, so the sections are somewhat longer than in the product; their shares are what the numbers are for.
#ifdef BTS_TICKS
#define BW_TICK(i)                                                   \
  {                                                                  \
    const unsigned long long t_now = __builtin_readcyclecounter();   \
    t_acc[i] += t_now - t_last;                                      \
    t_last = t_now;                                                  \
  }
constexpr long kTicksScatterOffset = 4096L * 16;   // scatter_kernel's records start behind 4096 wave records of rows_kernel
#else
#define BW_TICK(i)
#endif

// Pass C ends with every work-group adding its 40 x HD partial sums of dW_pe / db_in into the SAME 40 x HD addresses of d_mlp: ~500-770
// float atomics per address, issued by work-groups that all finish within microseconds of each other.  Measured (profiles/r04j, flush
// ablated): 0.16 of dwpe_kernel's 0.33 ms and 0.09 of dwpe_rows_kernel's 0.18 ms were these atomics queueing at L2.  The work-groups
// therefore add into one of kFlushSlots copies (work-group b -> slot b % kFlushSlots: 1/8 of the queue per address), and a 40 x HD
// thread kernel (dwpe_reduce_kernel) folds the copies into d_mlp.
// the status of the launches issued so far, read ONCE (hipGetLastError clears it): the HIP error string goes to bts_last_error()
inline int launch_status() {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return BTS_OK;
  set_error("%s: backward kernel launch failed (%ld)", hipGetErrorString(e), (long)e);
  return BTS_E_LAUNCH;
}

constexpr int kFlushSlots = 8;
constexpr int kFlushRows = kPeDim + 1;   // 40 encoding inputs incl. the bias row

// A ray's wave-uniform inputs of the backward -- origin, direction (8 floats of `rays`), its upstream colour gradient (nv * 3 floats)
// and depth gradient -- fetched ONE ITERATION AHEAD as one vector load: lane i < 8 holds float i of the ray, lanes 8 .. 8 + 3 NVMAX - 1 the
// colour gradient (index clamped to the nv * 3 that exist), lane 48 the depth gradient.  As scalar loads at the top of the iteration they
// were a full memory round trip with the wave idle (7 % of rowsb_kernel's iteration, profiles/r03_experiments/r03v section 10); issued early as scalar
// loads they would turn every LDS wait behind them into lgkmcnt(0).  The iteration's head moves the lanes into SGPRs (v_readlane).
template <class Q>
__device__ __forceinline__ float fetch_ray_record(Q q, long ray, int nv3, int lane) {
  const float* a = q->f.rays + ray * 8 + (lane & 7);
  const float* b = q->g_rgb ? q->g_rgb + ray * nv3 + min(max(lane - 8, 0), max(nv3 - 1, 0)) : a;
  const float* c = q->g_depth ? q->g_depth + ray : a;
  return *(lane < 8 ? a : (lane < 48 ? b : c));
}
template <int NV3>
struct RayIn {
  float o[3], d[3], g_rgb[NV3], g_bkgd, g_depth;
};
template <int NV3, class Q>
__device__ __forceinline__ RayIn<NV3> unpack_ray_record(Q q, float rec, int nv3) {
  RayIn<NV3> r;
  r.o[0] = lane_value(rec, 0), r.o[1] = lane_value(rec, 1), r.o[2] = lane_value(rec, 2);
  r.d[0] = lane_value(rec, 3), r.d[1] = lane_value(rec, 4), r.d[2] = lane_value(rec, 5);
  const bool has = q->g_rgb != nullptr;
  r.g_bkgd = 0.0f;
#pragma unroll
  for (int i = 0; i < NV3; ++i) {
    r.g_rgb[i] = (has && i < nv3) ? lane_value(rec, 8 + i) : 0.0f;
    r.g_bkgd -= r.g_rgb[i];
  }
  r.g_depth = q->g_depth ? lane_value(rec, 48) : 0.0f;
  return r;
}

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
