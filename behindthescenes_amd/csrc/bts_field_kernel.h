// Fused forward of the BehindTheScenes density-field renderer for gfx950 (MI355X) -- kernel template.
//
// One lane = one ray (or one query point).  A wave walks its 64 rays front-to-back, one sample per iteration:
//   project into the encoder view -> bilinear feature fetch -> positional encoding -> lin_in as
//   H^T[Hd x 64 pts] = W^T . X^T on v_mfma_f32_32x32x2_f32 (weights = A operand from LDS, the lane's own inputs = B operand
//   after one v_permlane32_swap per k-pair) -> optional ResnetBlockFC layers (C-layout -> B-layout is free through a k
//   permutation) -> lin_out as an in-lane dot product over the accumulator registers -> softplus -> colour taps of the nv
//   render views -> alpha compositing in registers.  sigma / alpha / T never leave the chip.
//
// Feature fetch, two variants (template PROJ):
//   PROJ = true  (default path): the caller passes G = F . w_in[:, :C]^T (bts_project_features).  Because bilinear
//                interpolation and lin_in are linear, lin_in(bilinear(F)) = bilinear(G) + w_in[:, C:] . PE + b_in: the four taps
//                of G are blended straight into the accumulator layout and only the 40 PE/bias inputs go through MFMA
//                (5 120 instead of 13 312 FLOP per point).  DECLARED ALGEBRAIC SHORTCUT (changes summation order only).
//   PROJ = false: raw channels-last F; 8 channels per chunk are blended per lane and pushed through MFMA (k = 104).
//
// Replaces: nerf.py:210-313 (composite), models_bts.py:138-338 (sample_features / sample_colors / forward),
//           resnetfc.py:132-184, code.py:30-42 of the reference.
#pragma once
#include "bts_common.h"

#include <cstdio>
#include <cstring>

namespace bts {

struct FwdParams {
  // field
  const float* feat;   // (n,H,W,C)   raw features (PROJ = false)
  const float* proj;   // (n,H,W,HD)  projected features G (PROJ = true)
  const float* K_enc;  // (n,3,3)
  const float* w2c_enc;
  const float* imgs;   // (n,nv,H,W,4)
  const float* K_r;
  const float* w2c_r;
  const float* empty_feature;
  const float* mlp;
  int n, H, W, nv;
  int fs;        // log2 of the feature map's downscale (BtsFieldCfg.feat_shift): G is (n, H >> fs, W >> fs, HD)
  int enc_view;  // render view whose camera is the encoder's (BtsFieldCfg.enc_render_view), -1: none (or fs > 0: other texel indices)
  int code_mode, inv_z, learn_empty, empty_empty;
  float freq_factor, d_min, d_max;
  float inv_dmax, inv_range, range;
  // render
  const float* rays;
  const float* z_samp;
  const float* jitter;   // z_samp == null: stratified jitter u (n*Bp, K), the sample depths are computed in the kernel (BtsRenderArgs.jitter)
  float* z_out;          // ... and written here when non-null (BtsRenderArgs.z_samp_out)
  int lindisp;
  int Bp, K, hard_cap, white_bkgd;
  float* rgb;
  float* depth;
  float* weights;
  float* alphas;
  float* invalid;
  float* rgb_samps;
  float* sigma_raw;
  float* trans;        // (n*Bp, K) transmittance in front of each sample (saved for the backward)
  float* invalid_wsum; // (n*Bp, nv) sum_k w_k invalid_k,v } per-ray reductions for the loss' invalid-ray policies (pipelined
  float* invalid_any;  // (n*Bp, nv) max_k invalid_k,v     } kernel only)
  const float* sigma_noise;  // (n*Bp, K) or null: added to the density before relu / alpha (nerf.py:279-280)
  // query
  const float* xyz;
  float* q_sigma;
  int only_density;
  int tiles_per_sample;
  // lane = sample render kernel
  unsigned long long* dbg;  // probe builds only: per-wave section cycle counters (bit 128 of ablate)
  int ablate;    // probe builds only (-DBTS_PROBE): bit mask of kernel sections to skip (tools/ablate_probe.py)
  int lpr;       // lanes per ray: 8, 16, 32 or 64 (>= min(K, 64)); 64 / lpr rays share one wave iteration
  int chunk_log2;  // pipelined kernel: ray groups per chunk of the XCD interleave (log2), see render_kernel_p
  long groups;   // number of ray groups (= n * Bp * lpr / 64)
};

#ifdef BTS_PROBE
#define BTS_ABL(bit) ((p.ablate & (bit)) != 0)
#else
#define BTS_ABL(bit) false
#endif

template <int C, int HD, int NB, bool PROJ>
struct Lds {
  static constexpr int CF = PROJ ? 0 : C;            // feature rows of lin_in evaluated per point (none with projected features)
  static constexpr int KIN = CF + kPeDim + 1;        // (features +) PE + bias row (even)
  static constexpr int W_IN = 0;                     // [KIN][HD]  k-major: Wl[k*HD + hid] = w_in[hid][k]
  static constexpr int BLK = W_IN + KIN * HD;        // per block: w0 [HD in][HD out], b0 [HD], w1 [HD][HD], b1 [HD]
  static constexpr int BLK_STRIDE = 2 * HD * HD + 2 * HD;
  static constexpr int W_OUT = BLK + NB * BLK_STRIDE;  // [HD]
  static constexpr int EMPTY = W_OUT + HD;             // [C] raw empty feature, or [HD] projected (w_in[:, :C] . empty) with PROJ
  static constexpr int TOTAL = EMPTY + (PROJ ? HD : C);
};

// Kernel-side order of the lin_in inputs (any permutation of k is free as long as A and B agree):
//   [0, C) features | C: x, C+1: y | C+2: depth code, C+3: constant 1 (bias row) | C+4+6*oct+{0,1,2}: sin(f x,y,z),
//   +{3,4,5}: sin(f . + pi/2).   Reference order (code.py:37-42): [features, x, y, z, oct0: sin(3) cos(3), oct1 ...].
template <int C>
__host__ __device__ constexpr int kernel_to_ref_input(int k) {
  if (k < C + 2) return k;
  if (k == C + 2) return C + 2;
  if (k == C + 3) return -1;
  return k - 1;
}

// stage the MLP into LDS in the k-major layouts the MFMA A operand wants
template <int C, int HD, int NB, bool PROJ>
__device__ __forceinline__ void stage_weights(float* lds, const float* __restrict__ mlp, const float* __restrict__ empty) {
  using L = Lds<C, HD, NB, PROJ>;
  constexpr int D_IN = C + kPeDim;
  const MlpLayout ml{D_IN, HD, NB};
  for (int i = threadIdx.x; i < L::KIN * HD; i += blockDim.x) {
    const int k = i / HD, hid = i % HD;
    const int src = kernel_to_ref_input<C>(k + (PROJ ? C : 0));  // -1: bias row
    lds[L::W_IN + i] = src >= 0 ? mlp[ml.w_in() + hid * D_IN + src] : mlp[ml.b_in() + hid];
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    float* base = lds + L::BLK + b * L::BLK_STRIDE;
    for (int i = threadIdx.x; i < HD * HD; i += blockDim.x) {
      const int k = i / HD, out = i % HD;
      base[i] = mlp[ml.blk_w0(b) + out * HD + k];
      base[HD * HD + HD + i] = mlp[ml.blk_w1(b) + out * HD + k];
    }
    for (int i = threadIdx.x; i < HD; i += blockDim.x) {
      base[HD * HD + i] = mlp[ml.blk_b0(b) + i];
      base[2 * HD * HD + HD + i] = mlp[ml.blk_b1(b) + i];
    }
  }
  for (int i = threadIdx.x; i < HD; i += blockDim.x) lds[L::W_OUT + i] = mlp[ml.w_out() + i];
  if constexpr (PROJ) {
    for (int hid = threadIdx.x; hid < HD; hid += blockDim.x) {
      float a = 0.0f;
      if (empty)
        for (int c = 0; c < C; ++c) a = __builtin_fmaf(mlp[ml.w_in() + hid * D_IN + c], empty[c], a);
      lds[L::EMPTY + hid] = a;
    }
  } else {
    for (int i = threadIdx.x; i < C; i += blockDim.x) lds[L::EMPTY + i] = empty ? empty[i] : 0.0f;
  }
}

// ---- projected features: gather G straight into the accumulator (C/D) layout --------------------------------------
// Lane l (half h = l >> 5) holds rows ht*32 + 8q + 4h + e (q, e = 0..3) of the 32x32 tile for point (l & 31) of point tile pt.
// G is stored with those 16 values contiguous (proj_storage_index), so per (pt, ht, tap) a lane reads ONE 64-byte piece = 4 float4,
// and lanes l / l+32 together consume one whole 128-byte line in the same instructions (no L1 re-fetch).  A stage = (pt, ht, tap
// pair) = 8 float4; the next stage's loads are issued before the current one is blended (ATen's nw, ne, sw, se order is kept).
struct GBuf {
  float4 v[2][4];  // [tap in pair][q]
};

template <int HD>
__device__ __forceinline__ void gload(GBuf& b, const float4* __restrict__ G, const int (&o)[4], int tp2, int idx4) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float4* row = G + (long)o[2 * tp2 + t] * (HD / 4) + idx4;
#pragma unroll
    for (int q = 0; q < 4; ++q) b.v[t][q] = row[q];
  }
}


// blend two taps into 16 accumulator rows with packed FMAs (v_pk_fma_f32); tap order nw, ne | sw, se as in ATen
template <bool FIRST>
__device__ __forceinline__ void gblend(f32x16& acc, const GBuf& b, float w0, float w1) {
  const f32x2 w0v = {w0, w0}, w1v = {w1, w1};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
      const float* a = reinterpret_cast<const float*>(&b.v[0][q]) + e;
      const float* c = reinterpret_cast<const float*>(&b.v[1][q]) + e;
      const f32x2 av = {a[0], a[1]}, cv = {c[0], c[1]};
      f32x2 s;
      if constexpr (FIRST) s = av * w0v;
      else s = __builtin_elementwise_fma(av, w0v, (f32x2){acc[4 * q + e], acc[4 * q + e + 1]});
      s = __builtin_elementwise_fma(cv, w1v, s);
      acc[4 * q + e] = s[0];
      acc[4 * q + e + 1] = s[1];
    }
  }
}

// S-th stage of the gather sequence over (pt, ht, tap pair); buffers alternate
template <int HD, int S>
__device__ __forceinline__ void gather_seq(f32x16 (&acc)[HD / 32][2], GBuf& cur, GBuf& nxt, const float4* __restrict__ G,
                                           const int (&o)[2][4], const float (&w)[2][4], int h) {
  constexpr int HT = HD / 32;
  constexpr int NS = 2 * HT * 2;
  if constexpr (S < NS) {
    constexpr int pt = S / (2 * HT), ht = (S / 2) % HT, tp2 = S % 2;
    if constexpr (S + 1 < NS) {
      constexpr int pt1 = (S + 1) / (2 * HT), ht1 = ((S + 1) / 2) % HT, tp1 = (S + 1) % 2;
      gload<HD>(nxt, G, o[pt1], tp1, ht1 * 8 + 4 * h);
    }
    if constexpr (tp2 == 0) gblend<true>(acc[ht][pt], cur, w[pt][0], w[pt][1]);
    else gblend<false>(acc[ht][pt], cur, w[pt][2], w[pt][3]);
    gather_seq<HD, S + 1>(acc, nxt, cur, G, o, w, h);
  }
}

// learn_empty (models_bts.py:181-182): points outside the encoder frustum take the (projected) empty feature instead
template <int HD>
__device__ __forceinline__ void apply_empty(f32x16 (&acc)[HD / 32][2], const bool (&emp)[2], const float* lds_empty, int h) {
#pragma unroll
  for (int ht = 0; ht < HD / 32; ++ht)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float ev = lds_empty[ht * 32 + mfma_row(q, 0) + 4 * h];
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) acc[ht][pt][q] = emp[pt] ? ev : acc[ht][pt][q];
    }
}

// value of this lane's ray (v) -> {value of ray (l & 31), value of ray 32 + (l & 31)} on every lane
__device__ __forceinline__ void bcast_tiles(unsigned v, unsigned& t0, unsigned& t1) {
  auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  t0 = r[0];
  t1 = r[1];
}

// one k-pair of lin_in: inputs (a, b) = the lane's x[2s], x[2s+1]
template <int HD>
__device__ __forceinline__ void kstep(f32x16 (&acc)[HD / 32][2], const float* wl, int lane_off, float a, float b, bool nomfma = false) {
  swap32(a, b);
#pragma unroll
  for (int ht = 0; ht < HD / 32; ++ht) {
    const float w = wl[lane_off + ht * 32];
    if (nomfma) {  // probe builds only
      acc[ht][0][0] += w * a, acc[ht][1][0] += w * b;
      continue;
    }
    acc[ht][0] = mfma(w, a, acc[ht][0]);
    acc[ht][1] = mfma(w, b, acc[ht][1]);
  }
}

// First k-pair of a FRESH accumulator set: acc = W . x (srcC = inline 0, saves zero-filling 64 registers per sample)
template <int HD>
__device__ __forceinline__ void kstep_first(f32x16 (&acc)[HD / 32][2], const float* wl, int lane_off, float a, float b) {
  swap32(a, b);
  f32x16 zero;
#pragma unroll
  for (int q = 0; q < 16; ++q) zero[q] = 0.0f;
#pragma unroll
  for (int ht = 0; ht < HD / 32; ++ht) {
    const float w = wl[lane_off + ht * 32];
    acc[ht][0] = mfma(w, a, zero);
    acc[ht][1] = mfma(w, b, zero);
  }
}

// out[ot][pt] += W^T(k-major, [HD in][HD out]) . relu(in)   (one ResnetBlockFC linear; C-layout of `in` feeds B directly).
// RELU_IN = false with w = the ROW-major weight ([out][in] = nn.Linear.weight) gives the backward product W^T . g.
template <int HD, bool RELU_IN = true>
__device__ __forceinline__ void hidden_layer(f32x16 (&out)[HD / 32][2], const f32x16 (&in)[HD / 32][2], const float* w, int lane) {
  const int h = lane >> 5, col = lane & 31;
#pragma unroll
  for (int it = 0; it < HD / 32; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kin = it * 32 + mfma_row(r, 0) + 4 * h;
      const float b0 = RELU_IN ? fmaxf(in[it][0][r], 0.0f) : in[it][0][r];
      const float b1 = RELU_IN ? fmaxf(in[it][1][r], 0.0f) : in[it][1][r];
#pragma unroll
      for (int ot = 0; ot < HD / 32; ++ot) {
        const float a = w[kin * HD + ot * 32 + col];
        out[ot][0] = mfma(a, b0, out[ot][0]);
        out[ot][1] = mfma(a, b1, out[ot][1]);
      }
    }
  }
}

__device__ __forceinline__ void load_chunk(float4 (&buf)[4][2], const float4* t00, const float4* t01, const float4* t10,
                                           const float4* t11, int c) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    buf[0][j] = t00[2 * c + j], buf[1][j] = t01[2 * c + j];
    buf[2][j] = t10[2 * c + j], buf[3][j] = t11[2 * c + j];
  }
}

// bilinear blend of one 8-channel chunk (ATen accumulates nw, ne, sw, se in this order) and its 4 k-pairs of lin_in
template <int HD>
__device__ __forceinline__ void feature_chunk(f32x16 (&acc)[HD / 32][2], const float4 (&buf)[4][2], const Taps& tp, bool learn_empty,
                                              bool use_empty, const float* empty, const float* wl) {
  float f[8];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float* a = reinterpret_cast<const float*>(&buf[0][j]);
    const float* b = reinterpret_cast<const float*>(&buf[1][j]);
    const float* c = reinterpret_cast<const float*>(&buf[2][j]);
    const float* d = reinterpret_cast<const float*>(&buf[3][j]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float s = a[e] * tp.w00;
      s = s + b[e] * tp.w01;
      s = s + c[e] * tp.w10;
      s = s + d[e] * tp.w11;
      f[4 * j + e] = s;
    }
  }
  if (learn_empty) {
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = use_empty ? empty[e] : f[e];
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) kstep<HD>(acc, wl + 2 * s * HD, 0, f[2 * s], f[2 * s + 1]);
}

// sin and cos of one argument: 3-term FMA Cody-Waite reduction by pi/2 + minimax polynomials on [-pi/4, pi/4].
// Max abs error 9.5e-8 for |arg| <= 1e5 (validated against fp64 on 4e7 PE arguments; libm sinf: 3.3e-8) at ~1/3 of ocml's
// instruction count.  Larger arguments (points within 1e-3 of the encoder's camera plane) take the ocml path.
__device__ __forceinline__ void sincos_small(float arg, float& s, float& c) {
  const float j = rintf(arg * 0.63661977236758134308f);
  float r = __builtin_fmaf(-j, 1.57079625129699707031f, arg);
  r = __builtin_fmaf(-j, 7.54978941586159635335e-8f, r);  // third term (5.4e-15 j <= 3.5e-10 for |arg| <= 1e5) dropped
  const float r2 = r * r;
  const float sp = __builtin_fmaf(r2, __builtin_fmaf(r2, __builtin_fmaf(r2, 2.6083159809786593541503e-06f, -1.981069071916863322258e-04f),
                                                     8.333307858556509017944e-03f), -1.666666597127914428711e-01f);
  const float sn = __builtin_fmaf(r * r2, sp, r);
  const float cp = __builtin_fmaf(r2, __builtin_fmaf(r2, __builtin_fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f),
                                                     4.166664568298827e-2f), -0.5f);
  const float cs = __builtin_fmaf(r2, cp, 1.0f);
  const int q = (int)j;
  const float ss = (q & 1) ? cs : sn;
  const float cc = (q & 1) ? sn : cs;
  s = (q & 2) ? -ss : ss;
  c = ((q + 1) & 2) ? -cc : cc;
}

// The same for TWO arguments at once (x and y of a point; or one coordinate at two octaves): the argument reduction and both
// polynomials are packed FP32 instructions (v_pk_mul_f32 / v_pk_fma_f32: two IEEE operations per lane and instruction, the rate the
// 157 TF fp32 vector peak is quoted at), only rounding, conversion and the quadrant selects stay per element.  Element by element
// the operation sequence is sincos_small's: the results are bit-identical to it.
__device__ __forceinline__ void sincos_small2(f32x2 arg, f32x2& s, f32x2& c) {
  const f32x2 t = arg * (f32x2){0.63661977236758134308f, 0.63661977236758134308f};
  const f32x2 j = {rintf(t[0]), rintf(t[1])};
  f32x2 r = __builtin_elementwise_fma(-j, (f32x2){1.57079625129699707031f, 1.57079625129699707031f}, arg);
  r = __builtin_elementwise_fma(-j, (f32x2){7.54978941586159635335e-8f, 7.54978941586159635335e-8f}, r);
  const f32x2 r2 = r * r;
  auto k2 = [](float v) { return (f32x2){v, v}; };
  const f32x2 sp = __builtin_elementwise_fma(r2, __builtin_elementwise_fma(r2, __builtin_elementwise_fma(r2, k2(2.6083159809786593541503e-06f), k2(-1.981069071916863322258e-04f)),
                                                                            k2(8.333307858556509017944e-03f)), k2(-1.666666597127914428711e-01f));
  const f32x2 sn = __builtin_elementwise_fma(r * r2, sp, r);
  const f32x2 cp = __builtin_elementwise_fma(r2, __builtin_elementwise_fma(r2, __builtin_elementwise_fma(r2, k2(2.443315711809948e-5f), k2(-1.388731625493765e-3f)),
                                                                            k2(4.166664568298827e-2f)), k2(-0.5f));
  const f32x2 cs = __builtin_elementwise_fma(r2, cp, k2(1.0f));
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int q = (int)j[e];
    const float ss = (q & 1) ? cs[e] : sn[e];
    const float cc = (q & 1) ? sn[e] : cs[e];
    s[e] = (q & 2) ? -ss : ss;
    c[e] = ((q + 1) & 2) ? -cc : cc;
  }
}

// one PE octave: sin(f x), sin(f y), sin(f code), then the same with the fl32(pi/2) phase (code.py:25-28, 38).
// The reference's "cos" entry is sin(fl(arg + P)), P = fl32(pi/2): with the exact rounding error e of that addition (TwoSum),
// fl(arg + P) = arg + pi/2 + d, d = (P - pi/2) - e, so the entry equals cos(arg + d) = cos(arg) - d sin(arg) + O(d^2), |d| < 4e-6:
// one sincos gives both entries with the reference's argument rounding reproduced (max deviation from it 1.2e-7).
// raw sine / cosine of the three encoding arguments of one octave; x and y travel as a packed pair (see sincos_small2), the depth
// code on its own
struct SinCos3 {
  f32x2 sxy, cxy;
  float sz, cz;
};
__device__ __forceinline__ void pe_direct(SinCos3& r, const float (&v)[3], float f) {
  sincos_small2((f32x2){v[0], v[1]} * (f32x2){f, f}, r.sxy, r.cxy);
  sincos_small(v[2] * f, r.sz, r.cz);
}
// next octave by angle doubling: fl(v * 2f) = 2 fl(v * f) exactly, so sin / cos of the doubled ARGUMENT are 2sc and 1 - 2s^2 of the
// previous octave's; one doubling costs ~1e-7 extra absolute error (max 2.7e-7 vs 1.2e-7 direct, rms 4e-8 vs 2e-8 on PE
// arguments), which is why only every other octave is derived this way.
__device__ __forceinline__ void pe_double(SinCos3& r, const SinCos3& q) {
  const f32x2 t = q.sxy + q.sxy;
  r.sxy = t * q.cxy;
  r.cxy = __builtin_elementwise_fma(-t, q.sxy, (f32x2){1.0f, 1.0f});
  const float tz = q.sz + q.sz;
  r.sz = tz * q.cz;
  r.cz = __builtin_fmaf(-tz, q.sz, 1.0f);
}
// the six encoding entries of an octave from its raw sines / cosines: sin(arg), then the reference's "cos" = sin(fl(arg + P))
__device__ __forceinline__ void pe_entries(float (&o)[6], const SinCos3& r, const float (&v)[3], float f) {
  constexpr float P = 1.57079637050628662109375f;
  {
    const f32x2 Pv = {P, P};
    const f32x2 arg = (f32x2){v[0], v[1]} * (f32x2){f, f};
    const f32x2 sm = arg + Pv;
    const f32x2 bb = sm - arg;
    const f32x2 err = (arg - (sm - bb)) + (Pv - bb);   // arg + P = sm + err exactly
    const f32x2 d = (f32x2){4.371139000186241e-08f, 4.371139000186241e-08f} - err;     // (P - pi/2) - err
    const f32x2 ec = __builtin_elementwise_fma(-d, r.sxy, r.cxy);
    o[0] = r.sxy[0], o[1] = r.sxy[1], o[3] = ec[0], o[4] = ec[1];
  }
  {
    const float arg = v[2] * f;
    const float sm = arg + P;
    const float bb = sm - arg;
    const float err = (arg - (sm - bb)) + (P - bb);
    const float d = 4.371139000186241e-08f - err;
    o[2] = r.sz;
    o[5] = __builtin_fmaf(-d, r.sz, r.cz);
  }
}
// scalar form of pe_entries: one argument's pair (sin, reference "cos" = sin(fl(arg + P)))
__device__ __forceinline__ void pe_entry1(float arg, float s, float c, float& es, float& ec) {
  constexpr float P = 1.57079637050628662109375f;
  const float sm = arg + P;
  const float bb = sm - arg;
  const float err = (arg - (sm - bb)) + (P - bb);   // arg + P = sm + err exactly
  const float d = 4.371139000186241e-08f - err;     // (P - pi/2) - err
  es = s;
  ec = __builtin_fmaf(-d, s, c);
}
// pe_octave_fast: branch-free (valid while every |argument| <= 1e5); pe_octave_exact: libm range reduction for any argument.
__device__ __forceinline__ void pe_octave_fast(float (&o)[6], const float (&v)[3], float f) {
  SinCos3 r;
  pe_direct(r, v, f);
  pe_entries(o, r, v, f);
}
__device__ __forceinline__ void pe_octave_exact(float (&o)[6], const float (&v)[3], float f) {
  constexpr float P = 1.57079637050628662109375f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float arg = v[i] * f;
    o[i] = sinf(arg);
    o[3 + i] = sinf(arg + P);
  }
}
// true when some octave argument of this point leaves the fast path's range (points within ~1e-3 of the encoder's camera plane)
__device__ __forceinline__ bool pe_needs_exact(const float (&v)[3], float freq_factor) {
  const float fmax_ = freq_factor * (float)(1 << (kNumFreqs - 1));
  return !(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fabsf(v[2])) * fmax_ <= 1.0e5f);
}
__device__ __forceinline__ void pe_octave(float (&o)[6], const float (&v)[3], float f) {
  bool big = false;
#pragma unroll
  for (int i = 0; i < 3; ++i) big |= !(fabsf(v[i] * f) <= 1.0e5f);
  if (__builtin_expect(__any(big), 0)) pe_octave_exact(o, v, f);
  else pe_octave_fast(o, v, f);
}

// Everything between a world point and its pre-softplus density: projection into the encoder view, bilinear feature fetch,
// positional encoding, lin_in (+ blocks) on MFMA, lin_out.  One lane = one point; all 64 lanes must be active (MFMA).
template <int C, int HD, int NB, bool PROJ>
__device__ __forceinline__ float eval_point(const FwdParams& p, const float* lds, const Cam& enc, const float4* __restrict__ featp,
                                            int lane, float b_out, float px, float py, float pz, Proj& pe) {
  using L = Lds<C, HD, NB, PROJ>;
  constexpr int HT = HD / 32;
  const int H = p.H, W = p.W;
  const int lane_off = (lane >> 5) * HD + (lane & 31);
  // ---------------- encoder view: projection, taps, depth code
  pe = p.code_mode == 1 ? project<true>(enc, px, py, pz) : project<false>(enc, px, py, pz);
  const Taps tp = make_taps(pe.x, pe.y, H, W, p.fs);
  float v3[3];
  v3[0] = pe.x, v3[1] = pe.y;
  v3[2] = depth_code(pe, p.code_mode == 1, p.inv_z != 0, p.inv_dmax, p.inv_range, p.d_min, p.range);
  const bool use_empty = (p.learn_empty != 0) & pe.invalid;

  f32x16 acc[HT][2];
#pragma unroll
  for (int ht = 0; ht < HT; ++ht)
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) acc[ht][pt] = zero_acc();

  const float* wl = lds + L::W_IN + lane_off;
  if constexpr (PROJ) {
    // ---------------- projected features: bilinear blend of G rows directly into the accumulators
    int o[2][4];
    float wq[2][4];
    bool emp[2];
    unsigned t0, t1;
    bcast_tiles((unsigned)tp.o00, t0, t1), o[0][0] = (int)t0, o[1][0] = (int)t1;
    bcast_tiles((unsigned)tp.o01, t0, t1), o[0][1] = (int)t0, o[1][1] = (int)t1;
    bcast_tiles((unsigned)tp.o10, t0, t1), o[0][2] = (int)t0, o[1][2] = (int)t1;
    bcast_tiles((unsigned)tp.o11, t0, t1), o[0][3] = (int)t0, o[1][3] = (int)t1;
    bcast_tiles(__float_as_uint(tp.w00), t0, t1), wq[0][0] = __uint_as_float(t0), wq[1][0] = __uint_as_float(t1);
    bcast_tiles(__float_as_uint(tp.w01), t0, t1), wq[0][1] = __uint_as_float(t0), wq[1][1] = __uint_as_float(t1);
    bcast_tiles(__float_as_uint(tp.w10), t0, t1), wq[0][2] = __uint_as_float(t0), wq[1][2] = __uint_as_float(t1);
    bcast_tiles(__float_as_uint(tp.w11), t0, t1), wq[0][3] = __uint_as_float(t0), wq[1][3] = __uint_as_float(t1);
    bcast_tiles(use_empty ? 1u : 0u, t0, t1), emp[0] = t0 != 0, emp[1] = t1 != 0;
    if (!BTS_ABL(1)) {
    GBuf ga, gb;
    gload<HD>(ga, featp, o[0], 0, 4 * (lane >> 5));
    gather_seq<HD, 0>(acc, ga, gb, featp, o, wq, lane >> 5);
    } else {
      acc[0][0][0] = wq[0][0] + wq[0][1] + wq[0][2] + wq[0][3] + (float)(o[0][0] + o[0][1] + o[0][2] + o[0][3]);
      acc[0][1][0] = wq[1][0] + wq[1][1] + wq[1][2] + wq[1][3] + (float)(o[1][0] + o[1][1] + o[1][2] + o[1][3]);
    }
    if (p.learn_empty && __any(use_empty)) apply_empty<HD>(acc, emp, lds + L::EMPTY, lane >> 5);
  } else {
  // ---------------- features: 8 channels per chunk; rolled ping-pong loop, the next chunk's 8 float4 loads are in
  // flight while this chunk's 16 MFMAs run (a fully unrolled loop lets the scheduler hoist all 64 loads and spill)
  const float4* t00 = featp + (long)tp.o00 * (C / 4);
  const float4* t01 = featp + (long)tp.o01 * (C / 4);
  const float4* t10 = featp + (long)tp.o10 * (C / 4);
  const float4* t11 = featp + (long)tp.o11 * (C / 4);
  float4 bufA[4][2], bufB[4][2];
  load_chunk(bufA, t00, t01, t10, t11, 0);
  const float* empty = lds + L::EMPTY;
#pragma unroll 1
  for (int c = 0; c < C / 8; c += 2) {
    load_chunk(bufB, t00, t01, t10, t11, c + 1);
    feature_chunk<HD>(acc, bufA, tp, p.learn_empty != 0, use_empty, empty + c * 8, wl + c * 8 * HD);
    if (c + 2 < C / 8) load_chunk(bufA, t00, t01, t10, t11, c + 2);
    feature_chunk<HD>(acc, bufB, tp, p.learn_empty != 0, use_empty, empty + (c + 1) * 8, wl + (c + 1) * 8 * HD);
  }
    wl += C * HD;
  }
  // ---------------- positional encoding (+ bias row): [x, y] [code, 1] then 3 k-pairs per octave, sines of the next
  // octave computed while the current octave's MFMAs run
  const bool nomfma = BTS_ABL(4);
  kstep<HD>(acc, wl, 0, v3[0], v3[1], nomfma);
  kstep<HD>(acc, wl + 2 * HD, 0, v3[2], 1.0f, nomfma);
  wl += 4 * HD;
  float sc[6], sn[6];
  if (BTS_ABL(2)) {
#pragma unroll
    for (int i = 0; i < 6; ++i) sc[i] = v3[i % 3] * p.freq_factor;
  } else {
    pe_octave(sc, v3, p.freq_factor);
  }
  float ff = p.freq_factor;
#pragma unroll 1
  for (int oct = 0; oct < kNumFreqs; ++oct) {
    ff = ff * 2.0f;
    if (oct + 1 < kNumFreqs) {
      if (BTS_ABL(2)) {
#pragma unroll
        for (int i = 0; i < 6; ++i) sn[i] = v3[i % 3] * ff;
      } else {
        pe_octave(sn, v3, ff);
      }
    }
    kstep<HD>(acc, wl, 0, sc[0], sc[1], nomfma);
    kstep<HD>(acc, wl + 2 * HD, 0, sc[2], sc[3], nomfma);
    kstep<HD>(acc, wl + 4 * HD, 0, sc[4], sc[5], nomfma);
    wl += 6 * HD;
#pragma unroll
    for (int i = 0; i < 6; ++i) sc[i] = sn[i];
  }

  // ---------------- ResnetBlockFC layers: h = h + fc_1(relu(fc_0(relu(h))))   (resnetfc.py:53-62)
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const float* base = lds + L::BLK + b * L::BLK_STRIDE;
    f32x16 net[HT][2];
#pragma unroll
    for (int ot = 0; ot < HT; ++ot)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float bias = base[HD * HD + ot * 32 + mfma_row(q, 0) + 4 * (lane >> 5)];
        net[ot][0][q] = bias, net[ot][1][q] = bias;
      }
    hidden_layer<HD>(net, acc, base, lane);
#pragma unroll
    for (int ot = 0; ot < HT; ++ot)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float bias = base[2 * HD * HD + HD + ot * 32 + mfma_row(q, 0) + 4 * (lane >> 5)];
        acc[ot][0][q] += bias, acc[ot][1][q] += bias;
      }
    hidden_layer<HD>(acc, net, base + HD * HD + HD, lane);
  }

  // ---------------- lin_out: in-lane dot over the hidden rows this lane holds, then fold the two lane halves
  float p0 = 0.0f, p1 = 0.0f;
  if (BTS_ABL(32)) {
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int ht = 0; ht < HT; ++ht) s0 += acc[ht][0][0] + acc[ht][0][5] + acc[ht][0][15], s1 += acc[ht][1][0] + acc[ht][1][5] + acc[ht][1][15];
    swap32(s0, s1);
    return (s0 + s1) + b_out;
  }
#pragma unroll
  for (int ht = 0; ht < HT; ++ht)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float w2 = lds[L::W_OUT + ht * 32 + mfma_row(q, 0) + 4 * (lane >> 5)];
      p0 = __builtin_fmaf(fmaxf(acc[ht][0][q], 0.0f), w2, p0);
      p1 = __builtin_fmaf(fmaxf(acc[ht][1][q], 0.0f), w2, p1);
    }
  swap32(p0, p1);  // p0 = {tile0.lo, tile1.lo}, p1 = {tile0.hi, tile1.hi}: lane l now holds both halves of ITS ray
  return (p0 + p1) + b_out;
}

template <int C, int HD, int NB, int NVMAX, bool QUERY, bool PROJ>
__global__ __launch_bounds__(256, 2) void field_kernel(const FwdParams p) {
  using L = Lds<C, HD, NB, PROJ>;
  constexpr int HT = HD / 32;
  __shared__ float lds[L::TOTAL];
  stage_weights<C, HD, NB, PROJ>(lds, p.mlp, p.empty_feature);
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int sample = wg / p.tiles_per_sample;  // wave-uniform batch element
  const int tile = wg - sample * p.tiles_per_sample;
  const int Bp = p.Bp;
  const int r_raw = tile * 256 + wave * 64 + lane;
  const bool active = r_raw < Bp;
  const int r = active ? r_raw : Bp - 1;
  const long ray = (long)sample * Bp + r;
  const int lane_off = (lane >> 5) * HD + (lane & 31);
  const int H = p.H, W = p.W, nv = p.nv;

  // wave-uniform cameras -> scalar registers
  const Cam enc = load_cam(p.w2c_enc + sample * 16, p.K_enc + sample * 9);
  const float4* __restrict__ featp =
      PROJ ? reinterpret_cast<const float4*>(p.proj) + (long)sample * H * W * (HD / 4)
           : reinterpret_cast<const float4*>(p.feat) + (long)sample * H * W * (C / 4);

  const float b_out = p.mlp[MlpLayout{C + kPeDim, HD, NB}.b_out()];

  float ox, oy, oz, dx, dy, dz;
  const float* zrow = nullptr;
  int K = 1;
  if constexpr (QUERY) {
    ox = p.xyz[ray * 3 + 0], oy = p.xyz[ray * 3 + 1], oz = p.xyz[ray * 3 + 2];
    dx = dy = dz = 0.0f;
  } else {
    const float4 r0 = reinterpret_cast<const float4*>(p.rays)[ray * 2];
    const float4 r1 = reinterpret_cast<const float4*>(p.rays)[ray * 2 + 1];
    ox = r0.x, oy = r0.y, oz = r0.z, dx = r0.w, dy = r1.x, dz = r1.y;
    K = p.K;
    zrow = p.z_samp + ray * K;
  }

  float T = 1.0f, depth = 0.0f, wsum = 0.0f;
  float rgb_acc[NVMAX * 3];
#pragma unroll
  for (int i = 0; i < NVMAX * 3; ++i) rgb_acc[i] = 0.0f;
  float z_next = QUERY ? 0.0f : zrow[0];

  for (int k = 0; k < K; ++k) {
    const float z = z_next;
    if constexpr (!QUERY) z_next = (k + 1 < K) ? zrow[k + 1] : 0.0f;
    // nerf.py:231  points = o + z * d   (mul, then add)
    const float px = QUERY ? ox : ox + z * dx;
    const float py = QUERY ? oy : oy + z * dy;
    const float pz = QUERY ? oz : oz + z * dz;

    Proj pe;
    const float s_raw = eval_point<C, HD, NB, PROJ>(p, lds, enc, featp, lane, b_out, px, py, pz, pe);
    float sigma = softplus(s_raw);
    if (p.empty_empty) sigma = pe.invalid ? 0.0f : sigma;
    if constexpr (!QUERY) {
      if (p.sigma_noise) sigma += p.sigma_noise[ray * K + k];
    }

    // ---------------- colour taps (models_bts.py:218-264)
    float col[NVMAX * 3];
    bool inv[NVMAX];
#pragma unroll
    for (int j = 0; j < NVMAX; ++j) {
      col[3 * j] = col[3 * j + 1] = col[3 * j + 2] = 0.0f;
      inv[j] = pe.invalid;
      if (j < nv) {
        const Cam cj = load_cam(p.w2c_r + ((long)sample * nv + j) * 16, p.K_r + ((long)sample * nv + j) * 9);
        const Proj pc = project<false>(cj, px, py, pz);
        const Taps tc = make_taps(pc.x, pc.y, H, W);
        const float4* img = reinterpret_cast<const float4*>(p.imgs) + ((long)sample * nv + j) * H * W;
        const float4 a = img[tc.o00], b = img[tc.o01], cc = img[tc.o10], d = img[tc.o11];
        col[3 * j + 0] = ((a.x * tc.w00 + b.x * tc.w01) + cc.x * tc.w10) + d.x * tc.w11;
        col[3 * j + 1] = ((a.y * tc.w00 + b.y * tc.w01) + cc.y * tc.w10) + d.y * tc.w11;
        col[3 * j + 2] = ((a.z * tc.w00 + b.z * tc.w01) + cc.z * tc.w10) + d.z * tc.w11;
        inv[j] = pc.invalid | pe.invalid;
      }
    }

    if constexpr (QUERY) {
      if (active) {
        p.q_sigma[ray] = sigma;
        if (p.only_density) {
          if (p.invalid) p.invalid[ray] = pe.invalid ? 1.0f : 0.0f;
        } else {
#pragma unroll
          for (int j = 0; j < NVMAX; ++j)
            if (j < nv) {
              if (p.invalid) p.invalid[ray * nv + j] = inv[j] ? 1.0f : 0.0f;
              p.rgb[(ray * nv + j) * 3 + 0] = col[3 * j + 0];
              p.rgb[(ray * nv + j) * 3 + 1] = col[3 * j + 1];
              p.rgb[(ray * nv + j) * 3 + 2] = col[3 * j + 2];
            }
        }
      }
    } else {
      // ---------------- alpha compositing (nerf.py:225-299)
      const float delta = (k + 1 < K) ? (z_next - z) : 1e10f;
      float alpha = 1.0f - transmittance(delta, sigma);
      if (p.hard_cap && k == K - 1) alpha = 1.0f;
      const float wgt = alpha * T;
      const float T_before = T;
      T = T * ((1.0f - alpha) + 1e-10f);
      depth = depth + wgt * z;
      wsum = wsum + wgt;
#pragma unroll
      for (int i = 0; i < NVMAX * 3; ++i) rgb_acc[i] = rgb_acc[i] + wgt * col[i];
      if (active) {
        const long pk = ray * K + k;
        if (p.weights) p.weights[pk] = wgt;
        if (p.alphas) p.alphas[pk] = alpha;
        if (p.sigma_raw) p.sigma_raw[pk] = s_raw;
        if (p.trans) p.trans[pk] = T_before;
        if (p.invalid) {
#pragma unroll
          for (int j = 0; j < NVMAX; ++j)
            if (j < nv) p.invalid[pk * nv + j] = inv[j] ? 1.0f : 0.0f;
        }
        if (p.rgb_samps) {
#pragma unroll
          for (int j = 0; j < NVMAX; ++j)
            if (j < nv) {
              p.rgb_samps[(pk * nv + j) * 3 + 0] = col[3 * j + 0];
              p.rgb_samps[(pk * nv + j) * 3 + 1] = col[3 * j + 1];
              p.rgb_samps[(pk * nv + j) * 3 + 2] = col[3 * j + 2];
            }
        }
      }
    }
  }

  if constexpr (!QUERY) {
    if (active) {
      p.depth[ray] = depth;
#pragma unroll
      for (int j = 0; j < NVMAX; ++j)
        if (j < nv) {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            float v = rgb_acc[3 * j + c];
            if (p.white_bkgd) v = (v + 1.0f) - wsum;  // nerf.py:301-304
            p.rgb[(ray * nv + j) * 3 + c] = v;
          }
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The render kernel: lane = SAMPLE.  A wave takes one ray (or 64/lpr short rays) per iteration and evaluates its K samples at
// once: z / weights / alphas / invalid rows are read and written as contiguous rows, all samples of a ray that originates at the
// encoder camera hit the SAME four texels (one broadcast load instead of 64 gathers), and the waves resident at any moment work
// on consecutive rays, so their texel footprints overlap in L1/L2 instead of evicting each other (with lane = ray every sample
// step of every resident wave re-touched ~1 KB per ray: 19 % L2 hit rate, >100x the compulsory HBM/MALL traffic).
// Alpha compositing is a segmented prefix product / sum over the lanes of a ray.
// ---------------------------------------------------------------------------------------------------------------
// One ray group (lpr lanes per ray, all chunks of K) rendered front to back -- the body of render_kernel, also used (out of line) by
// the pipelined kernel to re-render the rare rays whose encoding needs libm range reduction (eval_point handles that internally).
template <int C, int HD, int NB, int NVMAX, bool PROJ>
__device__ __forceinline__ void render_group(const FwdParams& p, const float* lds, long g, int lane, float b_out) {
  const int lpr = p.lpr, R = 64 / lpr;
  const int kl = lane & (lpr - 1);
  const int Bp = p.Bp, K = p.K, H = p.H, W = p.W, nv = p.nv;
  {
    const long ray = g * R + lane / lpr;
    const int sample = __builtin_amdgcn_readfirstlane((int)((g * R) / Bp));  // all rays of a group belong to one batch element
    const Cam enc = load_cam(p.w2c_enc + sample * 16, p.K_enc + sample * 9);
    const float4* __restrict__ featp =
        PROJ ? reinterpret_cast<const float4*>(p.proj) + (long)sample * H * W * (HD / 4)
             : reinterpret_cast<const float4*>(p.feat) + (long)sample * H * W * (C / 4);
    const float4 r0 = reinterpret_cast<const float4*>(p.rays)[ray * 2];
    const float4 r1 = reinterpret_cast<const float4*>(p.rays)[ray * 2 + 1];
    const float ox = r0.x, oy = r0.y, oz = r0.z, dx = r0.w, dy = r1.x, dz = r1.y;
    const float* zrow = p.z_samp + ray * K;

    float T_carry = 1.0f, depth_part = 0.0f, w_part = 0.0f;
    float rgb_part[NVMAX * 3];
#pragma unroll
    for (int i = 0; i < NVMAX * 3; ++i) rgb_part[i] = 0.0f;

    for (int kc = 0; kc < K; kc += 64) {
      const int k = kc + kl;
      const bool valid = k < K;
      const int kk = valid ? k : K - 1;
      const float z = zrow[kk];
      const float z_nx = zrow[min(kk + 1, K - 1)];
      // nerf.py:231  points = o + z * d   (mul, then add)
      const float px = ox + z * dx, py = oy + z * dy, pz = oz + z * dz;
      Proj pe;
      const float s_raw = eval_point<C, HD, NB, PROJ>(p, lds, enc, featp, lane, b_out, px, py, pz, pe);
      float sigma = softplus(s_raw);
      if (p.empty_empty) sigma = pe.invalid ? 0.0f : sigma;
      if (p.sigma_noise) sigma += p.sigma_noise[ray * K + kk];

      // ---------------- colour taps (models_bts.py:218-264)
      float col[NVMAX * 3];
      bool inv[NVMAX];
#pragma unroll
      for (int j = 0; j < NVMAX; ++j) {
        col[3 * j] = col[3 * j + 1] = col[3 * j + 2] = 0.0f;
        inv[j] = pe.invalid;
        if (j < nv && !BTS_ABL(8)) {
          const Cam cj = load_cam(p.w2c_r + ((long)sample * nv + j) * 16, p.K_r + ((long)sample * nv + j) * 9);
          const Proj pc = project<false>(cj, px, py, pz);
          const Taps tc = make_taps(pc.x, pc.y, H, W);
          const float4* img = reinterpret_cast<const float4*>(p.imgs) + ((long)sample * nv + j) * H * W;
          const float4 a = img[tc.o00], b = img[tc.o01], cc = img[tc.o10], d = img[tc.o11];
          col[3 * j + 0] = ((a.x * tc.w00 + b.x * tc.w01) + cc.x * tc.w10) + d.x * tc.w11;
          col[3 * j + 1] = ((a.y * tc.w00 + b.y * tc.w01) + cc.y * tc.w10) + d.y * tc.w11;
          col[3 * j + 2] = ((a.z * tc.w00 + b.z * tc.w01) + cc.z * tc.w10) + d.z * tc.w11;
          inv[j] = pc.invalid | pe.invalid;
        }
      }

      // ---------------- alpha compositing (nerf.py:225-299) as a segmented scan over the lanes of each ray
      const float delta = (k + 1 < K) ? (z_nx - z) : 1e10f;
      float alpha = 1.0f - transmittance(delta, sigma);
      if (p.hard_cap && k == K - 1) alpha = 1.0f;
      const float t = valid ? (1.0f - alpha) + 1e-10f : 1.0f;
      float incl = t;  // inclusive prefix product of t over the ray's lanes
      for (int d = 1; d < lpr; d <<= 1) {
        const float y = __shfl_up(incl, d, lpr);
        if (kl >= d) incl *= y;
      }
      float excl = __shfl_up(incl, 1, lpr);
      if (kl == 0) excl = 1.0f;
      const float T = T_carry * excl;
      T_carry = T_carry * __shfl(incl, lpr - 1, lpr);
      const float wgt = valid ? alpha * T : 0.0f;
      depth_part = depth_part + wgt * z;
      w_part = w_part + wgt;
#pragma unroll
      for (int i = 0; i < NVMAX * 3; ++i) rgb_part[i] = rgb_part[i] + wgt * col[i];
      if (valid && !BTS_ABL(16)) {
        const long pk = ray * K + k;
        if (p.weights) p.weights[pk] = wgt;
        if (p.alphas) p.alphas[pk] = alpha;
        if (p.sigma_raw) p.sigma_raw[pk] = s_raw;
        if (p.trans) p.trans[pk] = T;
        if (p.invalid) {
#pragma unroll
          for (int j = 0; j < NVMAX; ++j)
            if (j < nv) p.invalid[pk * nv + j] = inv[j] ? 1.0f : 0.0f;
        }
        if (p.rgb_samps) {
#pragma unroll
          for (int j = 0; j < NVMAX; ++j)
            if (j < nv) {
              p.rgb_samps[(pk * nv + j) * 3 + 0] = col[3 * j + 0];
              p.rgb_samps[(pk * nv + j) * 3 + 1] = col[3 * j + 1];
              p.rgb_samps[(pk * nv + j) * 3 + 2] = col[3 * j + 2];
            }
        }
      }
    }
    // ---------------- per-ray sums over the lanes of the ray
    for (int d = lpr >> 1; d >= 1; d >>= 1) {
      depth_part += __shfl_xor(depth_part, d, lpr);
      w_part += __shfl_xor(w_part, d, lpr);
#pragma unroll
      for (int i = 0; i < NVMAX * 3; ++i)
        if (i < nv * 3) rgb_part[i] += __shfl_xor(rgb_part[i], d, lpr);
    }
    if (kl == 0) {
      p.depth[ray] = depth_part;
#pragma unroll
      for (int i = 0; i < NVMAX * 3; ++i)
        if (i < nv * 3) p.rgb[ray * nv * 3 + i] = p.white_bkgd ? (rgb_part[i] + 1.0f) - w_part : rgb_part[i];  // nerf.py:301-304
    }
  }
}

template <int C, int HD, int NB, int NVMAX, bool PROJ>
__global__ __launch_bounds__(256, 2) void render_kernel(const FwdParams p) {
  using L = Lds<C, HD, NB, PROJ>;
  __shared__ float lds[L::TOTAL];
  stage_weights<C, HD, NB, PROJ>(lds, p.mlp, p.empty_feature);
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nwg = gridDim.x;  // multiple of 8
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int wg_per_xcd = nwg >> 3;
  const int xcd = wg / wg_per_xcd;
  const int lw = (wg - xcd * wg_per_xcd) * 4 + wave;  // wave index inside its XCD
  const int waves_per_xcd = wg_per_xcd * 4;
  const long gx = (p.groups + 7) >> 3;
  const long g_end = min(p.groups, (xcd + 1) * gx);
  const float b_out = p.mlp[MlpLayout{C + kPeDim, HD, NB}.b_out()];
  for (long g = xcd * gx + lw; g < g_end; g += waves_per_xcd) render_group<C, HD, NB, NVMAX, PROJ>(p, lds, g, lane, b_out);
}

// ---------------------------------------------------------------------------------------------------------------
// launch dispatch (instantiated per translation unit for one value of PROJ)
// ---------------------------------------------------------------------------------------------------------------
void set_error(const char* fmt, const char* a = "", long b = 0, long c = 0, long d = 0);

template <int C, int HD, int NB, bool QUERY, bool PROJ>
static int launch_nv(const FwdParams& p, int grid, hipStream_t s) {
  if (p.nv <= 1) field_kernel<C, HD, NB, 1, QUERY, PROJ><<<grid, 256, 0, s>>>(p);
  else if (p.nv <= 2) field_kernel<C, HD, NB, 2, QUERY, PROJ><<<grid, 256, 0, s>>>(p);
  else if (p.nv <= 4) field_kernel<C, HD, NB, 4, QUERY, PROJ><<<grid, 256, 0, s>>>(p);
  else field_kernel<C, HD, NB, 8, QUERY, PROJ><<<grid, 256, 0, s>>>(p);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: kernel launch failed (%ld)", hipGetErrorString(e), (long)e);
    return BTS_E_LAUNCH;
  }
  return BTS_OK;
}

template <int C, int HD, int NB, bool PROJ>
static int launch_render_nv(const FwdParams& p, int grid, hipStream_t s) {
  if (p.nv <= 1) render_kernel<C, HD, NB, 1, PROJ><<<grid, 256, 0, s>>>(p);
  else if (p.nv <= 2) render_kernel<C, HD, NB, 2, PROJ><<<grid, 256, 0, s>>>(p);
  else if (p.nv <= 4) render_kernel<C, HD, NB, 4, PROJ><<<grid, 256, 0, s>>>(p);
  else render_kernel<C, HD, NB, 8, PROJ><<<grid, 256, 0, s>>>(p);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: kernel launch failed (%ld)", hipGetErrorString(e), (long)e);
    return BTS_E_LAUNCH;
  }
  return BTS_OK;
}

template <bool PROJ>
int launch_render(const FwdParams& p, int C, int HD, int NB, int grid, hipStream_t s) {
  if (C == 64 && HD == 64 && NB == 0) return launch_render_nv<64, 64, 0, PROJ>(p, grid, s);
  if (C == 32 && HD == 32 && NB == 1) return launch_render_nv<32, 32, 1, PROJ>(p, grid, s);
  if (C == 32 && HD == 32 && NB == 0) return launch_render_nv<32, 32, 0, PROJ>(p, grid, s);
  set_error("%s: unsupported MLP shape C=%ld d_hidden=%ld n_blocks=%ld", "bts", C, HD, NB);
  return BTS_E_UNSUPPORTED;
}

template <bool QUERY, bool PROJ>
int launch_field(const FwdParams& p, int C, int HD, int NB, int grid, hipStream_t s) {
  if (C == 64 && HD == 64 && NB == 0) return launch_nv<64, 64, 0, QUERY, PROJ>(p, grid, s);
  if (C == 32 && HD == 32 && NB == 1) return launch_nv<32, 32, 1, QUERY, PROJ>(p, grid, s);
  if (C == 32 && HD == 32 && NB == 0) return launch_nv<32, 32, 0, QUERY, PROJ>(p, grid, s);
  set_error("%s: unsupported MLP shape C=%ld d_hidden=%ld n_blocks=%ld", "bts", C, HD, NB);
  return BTS_E_UNSUPPORTED;
}

}  // namespace bts
