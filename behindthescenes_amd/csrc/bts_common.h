// Shared device-side pieces of the gfx950 renderer: projection, bilinear taps, positional encoding, MFMA helpers.
// Written for CDNA4 only (wave64, v_mfma_f32_32x32x2_f32, v_permlane32_swap); compiled with -ffp-contract=off so that
// the scalar geometry follows the reference's unfused fp32 op sequence (only explicit __builtin_fmaf fuses).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bts_render.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define BTS_EPS 1e-3f  // models_bts.py:14

// work-groups per CU the forward kernel is compiled and launched for.  2 is the product; 3 exists for ONE experiment (review item 4: "is a
// third wave per SIMD worth 168 VGPRs?", profiles/r05m) and only fits with -DBTS_GATHER_REGS (three rings do not fit the CU's LDS)
#ifndef BTS_FWD_WAVES
#define BTS_FWD_WAVES 2
#endif

namespace bts {

constexpr int kNumFreqs = 6;            // every shipped config (configs/*.yaml `code.num_freqs`)
constexpr int kPeDim = 3 + 6 * kNumFreqs;  // 39

// ---------------------------------------------------------------------------------------------------------------
// wave-level helpers
// ---------------------------------------------------------------------------------------------------------------
// v_permlane32_swap: afterwards a = {a.lo, b.lo}, b = {a.hi, b.hi} (lo = lanes 0-31, hi = lanes 32-63).
// With one ray per lane and (a, b) = the ray's inputs (2s, 2s+1) this yields exactly the two B operands of
// v_mfma_f32_32x32x2_f32 for point tile 0 (rays 0-31) and point tile 1 (rays 32-63): B[k = lane>>5][j = lane&31].
__device__ __forceinline__ void swap32(float& a, float& b) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}

// row of the 32x32 MFMA C/D tile held in accumulator register r by a lane of half h (= lane >> 5)
__device__ __forceinline__ constexpr int mfma_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// A fresh accumulator.  (Round 1 kept this opaque to the compiler on the theory that an MFMA whose destination overlaps its own
// A / B registers -- which hipcc emits for srcC = 0 -- misbehaves on MI355X; tools/ubench/mfma_dst_overlap.hip disproves that:
// 0 mismatches in 1.7e10 values.  The nondeterminism that prompted it was the packed-FP32 erratum, see tools/ubench/pk_opsel_lanes.hip.)
__device__ __forceinline__ f32x16 zero_acc() {
  f32x16 z;
#pragma unroll
  for (int q = 0; q < 16; ++q) z[q] = 0.0f;
  return z;
}

// ---------------------------------------------------------------------------------------------------------------
// camera: rows of w2c[:3,:4] and K(3x3) kept in scalar registers (wave-uniform batch element)
// ---------------------------------------------------------------------------------------------------------------
struct Cam {
  float r[12];  // w2c rows 0..2, 4 entries each
  float k[9];
};

// Read-only kernel inputs addressed wave-uniformly (cameras, a wave's ray, scalars of the MLP) are read through the constant
// address space: the compiler then emits s_load into SGPRs instead of vector loads that pin ~50 VGPRs per wave.  Only valid for
// buffers the kernel never writes.
typedef const float __attribute__((address_space(4)))* cfp;
__device__ __forceinline__ cfp as_const(const float* p) { return (cfp)(unsigned long)p; }

__device__ __forceinline__ Cam load_cam(const float* w2c_, const float* K_) {
  const cfp w2c = as_const(w2c_), K = as_const(K_);
  Cam c;
#pragma unroll
  for (int i = 0; i < 12; ++i) c.r[i] = w2c[i];
#pragma unroll
  for (int i = 0; i < 9; ++i) c.k[i] = K[i];
  return c;
}

// A kernel's by-value parameter struct as it lies in the kernarg segment (offset 0: every kernel here takes ONE struct).  The
// persistent kernels read their parameters through this view, laundered once per iteration (asm volatile "+s"): hipcc otherwise loads
// all ~60 parameter dwords before the loop, keeps them in SGPRs for its whole life and -- 106 SGPRs do not hold them next to the
// cameras, the ray and the loop state -- spills ~120 of them to VGPR lanes: ~300 v_writelane / v_readlane per iteration of the render
// kernel, VALU instructions in a VALU-issue-bound loop.  Reloading a field where it is used is an s_load from the scalar cache (SMEM
// issues beside the VALU); the spill traffic drops to ~100 lane operations per iteration.
template <class T>
__device__ __forceinline__ const T __attribute__((address_space(4)))* kernarg_view() {
  return (const T __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
}

// lane `i` of a register as a wave-uniform value (an SGPR)
__device__ __forceinline__ float lane_value(float v, int i) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i));
}

// Element `idx` (a 32-bit lane value) of an array whose base is WAVE-UNIFORM: the byte offset is formed in 32 bits, so the access is
// the scalar-base form `global_load v, v_off32, s[base:base+1]` -- no 64-bit per-lane address (two VGPRs per access, computed ahead of
// the load and, in the register-starved kernels, spilled: a reload from scratch waits with vmcnt(0) for every load in flight).
// base[idx] written plainly is base + (zext(idx) << 2), which the compiler cannot narrow to 32 bits.  Arrays far below 4 GB only.
template <class T>
__device__ __forceinline__ T& at32(T* base, unsigned idx) {
  return *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + idx * (unsigned)sizeof(T));
}
template <class T>
__device__ __forceinline__ const T& at32(const T* base, unsigned idx) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + idx * (unsigned)sizeof(T));
}

// same through ordinary (vector) loads: the values land in VGPRs
__device__ __forceinline__ Cam load_cam_v(const float* __restrict__ w2c, const float* __restrict__ K) {
  Cam c;
#pragma unroll
  for (int i = 0; i < 12; ++i) c.r[i] = w2c[i];
#pragma unroll
  for (int i = 0; i < 9; ++i) c.k[i] = K[i];
  return c;
}

struct Proj {
  float x, y;     // normalised image coordinates
  float z;        // q.z  (depth after K)
  float rz;       // 1 / max(q.z, EPS)
  float dist;     // |R p + t|   (only meaningful when requested)
  bool invalid;
};

// ---------------------------------------------------------------------------------------------------------------
// Divisions and exponentials on the transcendental unit (v_rcp_f32 / v_exp_f32 / v_log_f32, 1 ulp each) with explicit correction
// steps instead of the compiler's IEEE sequences (v_div_scale x2, v_rcp, 4 fma, v_div_fmas, v_div_fixup = 10 instructions per quotient)
// and libm's expf / log1pf (~70 instructions each with their range handling).  The operands of this path are far inside the normal
// range (depths clamped to >= 1e-3, |x| of exp <= 1e4 after the clamp), so the scaling steps those sequences exist for are not needed.
// Accuracy against fp64 is measured on the device by tools/ubench/fast_math_check.hip (profiles/r04*/fast_math_check.txt).
// ---------------------------------------------------------------------------------------------------------------
// 1 / b: hardware reciprocal + one Newton step (error <= 0.5 ulp + 2^-46 relative: the IEEE quotient in all but near-tie cases)
__device__ __forceinline__ float rcp_nr(float b) {
  const float r = __builtin_amdgcn_rcpf(b);
  return __builtin_fmaf(r, __builtin_fmaf(-b, r, 1.0f), r);
}
// a / b given rb ~ 1 / b (within 1 ulp): quotient estimate + one residual correction -- the correctly rounded quotient (Markstein)
__device__ __forceinline__ float div_by(float a, float b, float rb) {
  const float q = a * rb;
  return __builtin_fmaf(__builtin_fmaf(-b, q, a), rb, q);
}
// e^x, x <= ~88: 2^(x log2 e) with the product's rounding error carried into a first-order correction.  <= 1.5 ulp; results below
// 2^-126 flush to 0 (v_exp_f32 has no denormal results): 1e-38 against 1 in 1 - exp(.) and in log(1 + exp(.)).
__device__ __forceinline__ float exp_fast(float x) {
  constexpr float c = 1.44269502162933349609375f, cc = 1.925963033500011e-8f;   // log2(e) = c + cc
  const float ph = x * c;
  const float pl = __builtin_fmaf(x, cc, __builtin_fmaf(x, c, -ph));
  const float r = __builtin_amdgcn_exp2f(ph);
  return __builtin_fmaf(r * pl, 0.693147182464599609375f, r);   // 2^(ph + pl) = r (1 + ln2 pl)
}
// F.softplus (beta 1, threshold 20) = log1p(exp(s)):  u = fl(1 + e) with its exact rounding error err (= e - (u - 1): both differences
// are exact in fp32), log(u + err) = ln2 log2(u) + err / u, the product with ln2 in two pieces.
__device__ __forceinline__ float softplus(float s) {
  const float e = exp_fast(s);
  const float u = 1.0f + e;
  const float err = e - (u - 1.0f);
  const float L = __builtin_amdgcn_logf(u);   // v_log_f32 = log2
  constexpr float c = 0.693147182464599609375f, cc = -1.904654323148236e-9f;   // ln 2 = c + cc
  const float h = L * c;
  float r = __builtin_fmaf(L, cc, __builtin_fmaf(L, c, -h));
  r = __builtin_fmaf(err, __builtin_amdgcn_rcpf(u), r);
  return s > 20.0f ? s : h + r;
}
// 1 - alpha of a sample: exp(-|delta| max(sigma, 0))  (nerf.py:283-285).  The clamp keeps the huge last interval (delta = 1e10) finite
// through exp_fast's correction term; exp(-1e4) is 0 in fp32 anyway.
__device__ __forceinline__ float transmittance(float delta, float sigma) {
  return exp_fast(fmaxf(-fabsf(delta) * fmaxf(sigma, 0.0f), -1.0e4f));
}
// d softplus / d s = sigmoid(s) = 1 / (1 + e^-s)
__device__ __forceinline__ float sigmoidf(float s) { return rcp_nr(1.0f + exp_fast(fminf(-s, 88.0f))); }

// models_bts.py:144-155 / 220-231.  (n,nv,3,4)@(n,1,4,P) then K@: sequential-k fused multiply-adds like a BLAS
// micro-kernel; divide and compares unfused.
template <bool WANT_DIST>
__device__ __forceinline__ Proj project(const Cam& c, float px, float py, float pz) {
  float cam[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float a = c.r[4 * i + 0] * px;
    a = __builtin_fmaf(c.r[4 * i + 1], py, a);
    a = __builtin_fmaf(c.r[4 * i + 2], pz, a);
    a = __builtin_fmaf(c.r[4 * i + 3], 1.0f, a);
    cam[i] = a;
  }
  float q[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float a = c.k[3 * i + 0] * cam[0];
    a = __builtin_fmaf(c.k[3 * i + 1], cam[1], a);
    a = __builtin_fmaf(c.k[3 * i + 2], cam[2], a);
    q[i] = a;
  }
  Proj p;
  p.z = q[2];
  const float zc = fmaxf(q[2], BTS_EPS);
  p.rz = rcp_nr(zc);
  p.x = div_by(q[0], zc, p.rz);
  p.y = div_by(q[1], zc, p.rz);
  p.invalid = (q[2] <= BTS_EPS) | (p.x < -1.0f) | (p.x > 1.0f) | (p.y < -1.0f) | (p.y > 1.0f);
  p.dist = WANT_DIST ? sqrtf(cam[0] * cam[0] + cam[1] * cam[1] + cam[2] * cam[2]) : 0.0f;
  return p;
}

// F.grid_sample(bilinear, border, align_corners=False) coordinates (ATen GridSampler.h:
// unnormalize ((x+1)*size-1)/2, clip to [0,size-1], floor, 4 weights).
struct Taps {
  int o00, o01, o10, o11;  // texel indices (y*W + x) of nw, ne, sw, se (clamped in-bounds; OOB taps have weight 0)
  float w00, w01, w10, w11;
};

// fs > 0: the map in memory is (H >> fs, W >> fs) and stands for its nearest-neighbour resize to H x W (BtsFieldCfg.feat_shift): the
// weights are those of the H x W map, the texel indices (and the x0 .. y1 handed back) those of the small one.  Two taps may then
// name the same texel with non-zero weights.
__device__ __forceinline__ Taps make_taps_xy(float x, float y, int H, int W, int& x0, int& y0, int& x1, int& y1, int fs = 0) {
  float ix = ((x + 1.0f) * (float)W - 1.0f) / 2.0f;
  float iy = ((y + 1.0f) * (float)H - 1.0f) / 2.0f;
  ix = fminf((float)(W - 1), fmaxf(ix, 0.0f));
  iy = fminf((float)(H - 1), fmaxf(iy, 0.0f));
  float fx = floorf(ix), fy = floorf(iy);
  float ex = fx + 1.0f, ey = fy + 1.0f;  // ix_se, iy_se
  Taps t;
  t.w00 = (ex - ix) * (ey - iy);
  t.w01 = (ix - fx) * (ey - iy);
  t.w10 = (ex - ix) * (iy - fy);
  t.w11 = (ix - fx) * (iy - fy);
  x0 = (int)fx, y0 = (int)fy;
  x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
  // NaN coordinates (never produced by finite inputs) would give x0 = INT_MIN: clamp for memory safety
  x0 = max(0, min(x0, W - 1));
  y0 = max(0, min(y0, H - 1));
  x0 >>= fs, x1 >>= fs, y0 >>= fs, y1 >>= fs, W >>= fs;
  t.o00 = y0 * W + x0;
  t.o01 = y0 * W + x1;
  t.o10 = y1 * W + x0;
  t.o11 = y1 * W + x1;
  return t;
}

__device__ __forceinline__ Taps make_taps(float x, float y, int H, int W, int fs = 0) {
  int x0, y0, x1, y1;
  return make_taps_xy(x, y, H, W, x0, y0, x1, y1, fs);
}

// ---- tiles of the projected map G / its gradient (one flag byte per tile: BtsRenderGrads.d_proj_tiles, bts_mark_sampled_tiles).
// A tile is 64 texels: 64 consecutive texels of the row-major map (rounds 4 - 5: always), or -- BtsFieldCfg.tile_blocks, where the map's
// height is a multiple of 4 and its width a multiple of 16: every shipped config at every scale -- a block of 4 rows x 16 texels,
// numbered row-major over the (H / 4) x (W / 16) blocks.  A training step's samples project onto streaks along the epipolar lines:
// at exp_kitti_360.yaml's batch they touch 14.6 % of the texels, 38.4 % of the 64 x 1 runs and 26.1 % of the 16 x 4 blocks
// (profiles/r04n/tile_stats.txt) -- a third less for both projection passes to read, contract and return to zero.  That pays where a
// block is four 4 KB pieces (channels-last F / dF, like G / dG); an NCHW map's 256-byte row pieces become 64-byte ones and the
// backward loses more than it saves (profiles/r06f): the caller picks.  The tile COUNT of a map is the same in both forms: ceil(H W / 64).
//   tw = tile_cols(Hm, Wm, blocks): blocks per map row in the 2-D form, 0 in the linear form
//   tile_of(y, x, Wm, tw):  the tile of texel (y, x)
//   tile_base(t, Wm, tw):   the first texel (y * Wm + x) of tile t;   tile_rs(Wm, tw): slot i of a tile is texel base + i + (i >> 4) * rs
__host__ __device__ inline int tile_cols(int Hm, int Wm, int blocks) { return (blocks && (Hm & 3) == 0 && (Wm & 15) == 0) ? (Wm >> 4) : 0; }
__device__ __forceinline__ unsigned tile_of(int y, int x, int Wm, int tw) {
  return tw ? (unsigned)((y >> 2) * tw + (x >> 4)) : (unsigned)(y * Wm + x) >> 6;
}
__device__ __forceinline__ int tile_base(int t, int Wm, int tw) {
  if (!tw) return t * 64;
  const int ty = t / tw;
  return ty * 4 * Wm + (t - ty * tw) * 16;
}
__device__ __forceinline__ int tile_rs(int Wm, int tw) { return tw ? Wm - 16 : 0; }

// depth code in [-1,1] (models_bts.py:157-171) of the projected point's depth (code_mode z) or distance (by_distance)
__device__ __forceinline__ float depth_code(const Proj& pe, bool by_distance, bool inv_z, float inv_dmax, float inv_range, float d_min, float range) {
  const float v = by_distance ? pe.dist : pe.z;
  float r;
  if (inv_z) {
    const float vc = fmaxf(v, BTS_EPS);
    const float rv = div_by(1.0f, vc, by_distance ? __builtin_amdgcn_rcpf(vc) : pe.rz);   // 1 / max(v, EPS)
    r = div_by(rv - inv_dmax, inv_range, rcp_nr(inv_range));
  } else {
    r = div_by(v - d_min, range, rcp_nr(range));
  }
  return __builtin_fmaf(2.0f, r, -1.0f);   // 2 r is exact: the same rounding as 2 * r - 1
}

// torch.linspace(start, end, steps)[i] as ATen's device kernel evaluates it (symmetric about the midpoint)
__device__ __forceinline__ float linspace_at(float start, float end, int steps, int i) {
  const float step = (end - start) / (float)(steps - 1);
  return i < steps / 2 ? start + step * (float)i : end - step * (float)(steps - i - 1);
}
// NeRFRenderer.sample_coarse (nerf.py:103-123): z_steps[k] = linspace(0, 1 - 1/K, K)[k] + u / K, then depth- or disparity-linear
// between near and far.  One routine for bts_sample_coarse and for the render kernels' in-kernel sampling: the two agree bit for bit.
__device__ __forceinline__ float coarse_base(int K, int k) { return K > 1 ? linspace_at(0.0f, 1.0f - 1.0f / (float)K, K, k) : 0.0f; }
__device__ __forceinline__ float coarse_depth(float u, float base, float step, float near, float far, bool lindisp) {
  const float sv = base + u * step;
  if (!lindisp) return near * (1.0f - sv) + far * sv;
  const float inv_near = div_by(1.0f, near, __builtin_amdgcn_rcpf(near)), inv_far = div_by(1.0f, far, __builtin_amdgcn_rcpf(far));
  const float den = inv_near * (1.0f - sv) + inv_far * sv;
  return div_by(1.0f, den, __builtin_amdgcn_rcpf(den));
}

// PE entry i of [x, y, zn | per octave k: sin(f_k x), sin(f_k y), sin(f_k zn), sin(f_k x + pi/2), ... ] (code.py:30-42);
// i == kPeDim is the constant 1 that multiplies the bias row.
template <int I>
__device__ __forceinline__ float pe_entry(const float (&v)[3], float freq_factor) {
  if constexpr (I < 3) {
    return v[I];
  } else if constexpr (I < kPeDim) {
    constexpr int j = I - 3;
    constexpr int oct = j / 6;
    constexpr int within = j % 6;
    constexpr int comp = within % 3;
    constexpr bool is_cos = within >= 3;
    const float f = freq_factor * (float)(1 << oct);  // exact power-of-two scaling, like freq_factor * 2.0**k in fp32
    float arg = v[comp] * f;
    if constexpr (is_cos) arg = arg + 1.57079637050628662109375f;  // fl32(pi/2) phase, addcmul(phase, x, f)
    return sinf(arg);
  } else if constexpr (I == kPeDim) {
    return 1.0f;
  } else {
    return 0.0f;
  }
}

// max(x, 0) as ONE instruction, v_med3_f32(x, 0, FLT_MAX) (= x clamped to [0, FLT_MAX]; hidden activations never reach 3.4e38).
// fmaxf on an MFMA result costs two: hipcc puts a canonicalising v_max(x, x) in front.  Not inline asm either: hipcc does not insert
// the MFMA-result -> VALU-read wait states around an asm statement.
__device__ __forceinline__ float relu1(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 3.4028234663852886e38f); }

// XCD-aware work-group remap: hardware places block b on XCD b % 8; give each XCD one contiguous range of tiles so
// that neighbouring rays (which share texels) hit the same L2.  Bijective for any nwg.
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = b & 7, idx = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Channel order of the projected feature map G in HBM.  A lane of half h holds accumulator rows ht*32 + 8q + 4h + e (q, e = 0..3)
// of a 32x32 MFMA tile; G stores those 16 values contiguously -- storage index ht*32 + 16h + 4q + e -- so that one lane reads ONE
// 64-byte piece per tap and hidden tile, and the two lane halves together consume exactly one 128-byte line.
__host__ __device__ constexpr int proj_storage_index(int hid) {
  const int ht = hid >> 5, r = hid & 31;
  const int q = r >> 3, h = (r >> 2) & 1, e = r & 3;
  return ht * 32 + h * 16 + q * 4 + e;
}
__host__ __device__ constexpr int proj_hidden_of_storage(int s) {
  const int ht = s >> 5, r = s & 31;
  const int h = r >> 4, q = (r >> 2) & 3, e = r & 3;
  return ht * 32 + 8 * q + 4 * h + e;
}

// packed MLP parameter offsets (see include/bts_render.h)
struct MlpLayout {
  int d_in, hd, nb;
  __host__ __device__ int w_in() const { return 0; }
  __host__ __device__ int b_in() const { return hd * d_in; }
  __host__ __device__ int blk(int i) const { return hd * d_in + hd + i * (2 * hd * hd + 2 * hd); }
  __host__ __device__ int blk_w0(int i) const { return blk(i); }
  __host__ __device__ int blk_b0(int i) const { return blk(i) + hd * hd; }
  __host__ __device__ int blk_w1(int i) const { return blk(i) + hd * hd + hd; }
  __host__ __device__ int blk_b1(int i) const { return blk(i) + 2 * hd * hd + hd; }
  __host__ __device__ int w_out() const { return blk(nb); }
  __host__ __device__ int b_out() const { return blk(nb) + hd; }
  __host__ __device__ int total() const { return blk(nb) + hd + 1; }
};

}  // namespace bts
