// Backward of the fused renderer, lane = SAMPLE form, general case: ResnetBlockFC layers (the RE10K model) and / or more than 64
// samples per ray.  Replaces the round-1 lane = ray kernel (bts_bwd.hip, one wave per SIMD, 0.08 of peak) for every shape the
// gate-bit passes of bts_bwd_rows.hip do not cover.
//
// The factorisation of bts_bwd_rows.hip survives the blocks: with the relu gates m0 (lin_in output h0),
// mn (fc_0 output n) and m1 (block output h1) the gradient of EVERY layer is g_s (the gradient at the pre-softplus density, one
// float per sample) times a vector that depends on the gates only:
//     v1 = m1 . w_out                       g_h1 = g_s v1            (lin_out, resnetfc.py:183)
//     vn = mn . (W1^T v1)                   g_n  = g_s vn            (fc_1,    resnetfc.py:53-62)
//     v0 = v1 + m0 . (W0^T vn)              g_h0 = g_s v0            (fc_0 + the residual path)
//   pass A  rowsb_kernel       one ray per wave iteration (64 samples at a time, chunks of a long ray back to front with the suffix
//                              sum carried), lane = sample: the FORWARD's pipeline (f16-split lin_in, gather through LDS, f16 block
//                              layers) recomputes h0, n, h1 bit-identically, the compositing gradient is a suffix scan across the
//                              lanes, the two transposed mat-vecs run on the matrix pipe in the C layout (the accumulator layout of
//                              one layer IS the B operand of the next), and u0 = g_s v0 goes to the workspace as one 4 d_hidden
//                              byte row per sample (storage order of G).  The fc_0 / fc_1 weight gradients need the activations
//                              themselves: dW1 = sum g_h1 (x) relu(n), dW0 = sum g_n (x) relu(h0) are contracted right here --
//                              both operands of a point tile transposed through the wave's (idle) gather ring, fp32 MFMA, two
//                              persistent 32x32 accumulators per wave.  dw_out, db_out, db_0, db_1 ride along.
//   pass B  scatter_kernel     (bts_bwd_rows.hip, ROWS form) dG += w_tap u0: the sliding LDS texel window, rows instead of gate bits
//   pass C  dwpe_rows_kernel   (bts_bwd_rows.hip) dW_pe, db_in = sum u0 (x) [pe, 1]: u0 rows straight from the workspace as the A
//                              operand (lane = channel reads its channel of sample k: no transposition), encoding recomputed.
// What torch.autograd would do for nerf.py:283-299 + models_bts.py:266-338 + resnetfc.py:53-62, 132-184 of the reference.
#define BTS_NO_LAUNCH_GLUE
#undef BTS_GATHER_REGS   // (the register-gather A/B build, variants/libbts_gatherregs.so, concerns the render kernels: the row passes of the backward exist in the LDS-gather form only)
#include "bts_render_kernel.h"
#include "bts_bwd.h"
#include <cstdlib>
#include <type_traits>

namespace bts {

#ifndef BTS_GATHER_LDS
#error "bts_bwd_blocks.hip is written against the LDS gather (the wave's gather ring doubles as its transposition tiles)"
#endif

// Weights of this pass in LDS next to the forward's f16 operands (LdsH): the encoding rows of lin_in in fp32 (k-major: the cold path's
// A operand), w_out, the projected empty feature, and per block fc_1 / fc_0 TRANSPOSED as split-precision f16 A operands (the layout of
// LdsH::W_BLK with the roles of input and output swapped) for the products W^T g of the backward.
template <int HD, int NB>
struct LdsB {
  static constexpr int PE_ROWS = kPeDim + 1;
  static constexpr int W_IN = 0;                       // [40][HD] k-major, kernel input order (kernel_to_ref_input)
  static constexpr int W_OUT = W_IN + PE_ROWS * HD;    // [HD]
  static constexpr int EMPTY = W_OUT + HD;             // [HD] projected empty feature (unscaled)
  static constexpr int VSCALE = EMPTY + HD;            // [0] s_v = a power of two with max |w_out| s_v in [8, 16), [1] 1 / s_v, [2] scratch
  static constexpr int BLK = VSCALE + 4;               // per block: layer 0 = fc_0^T, layer 1 = fc_1^T, each [term hi/lo][k-slice 2][64 lanes][8 halves] x 2^S
  static constexpr int BLK_TERM_STRIDE = 2 * 64 * 4;   // floats
  static constexpr int BLK_LAYER_STRIDE = 2 * BLK_TERM_STRIDE;
  static constexpr int BLK_STRIDE = 2 * BLK_LAYER_STRIDE;
  static constexpr int TOTAL = BLK + NB * BLK_STRIDE;
};

// `scale` = 2^S of the forward's f16 operands (LdsH::SCALE): the transposed copies carry the same factor
template <int C, int HD, int NB>
__device__ __forceinline__ void stage_weights_b(float* lb, const float* __restrict__ mlp, const float* __restrict__ empty, float scale) {
  using L = LdsB<HD, NB>;
  constexpr int D_IN = C + kPeDim;
  const MlpLayout ml{D_IN, HD, NB};
  for (int i = threadIdx.x; i < L::PE_ROWS * HD; i += blockDim.x) {
    const int k = i / HD, hid = i % HD;
    const int src = kernel_to_ref_input<C>(k + C);   // -1: bias row
    lb[L::W_IN + i] = src >= 0 ? mlp[ml.w_in() + hid * D_IN + src] : mlp[ml.b_in() + hid];
  }
  for (int i = threadIdx.x; i < HD; i += blockDim.x) lb[L::W_OUT + i] = mlp[ml.w_out() + i];
  for (int hid = threadIdx.x; hid < HD; hid += blockDim.x) {
    float a = 0.0f;
    if (empty)
      for (int c = 0; c < C; ++c) a = __builtin_fmaf(mlp[ml.w_in() + hid * D_IN + c], empty[c], a);
    lb[L::EMPTY + hid] = a;
  }
  if (threadIdx.x == 0) {
    float m = 0.0f;
    for (int i = 0; i < HD; ++i) m = fmaxf(m, fabsf(mlp[ml.w_out() + i]));
    int ex = 0;
    if (m > 0.0f && m < 3.0e38f) frexpf(m, &ex);      // m = f 2^ex, f in [0.5, 1)
    const int sv = max(-60, min(60, 4 - ex));
    lb[L::VSCALE] = ldexpf(1.0f, sv), lb[L::VSCALE + 1] = ldexpf(1.0f, -sv);
  }
  if constexpr (NB > 0) {
    _Float16* wb = reinterpret_cast<_Float16*>(lb + L::BLK);
    for (int i = threadIdx.x; i < NB * 2 * 2 * 64 * 8; i += blockDim.x) {
      const int e = i & 7, lane = (i >> 3) & 63, sl = (i >> 9) & 1, layer = i >> 10;   // layer = 2 * block + {0: fc_0^T, 1: fc_1^T}
      const int kout = mfma_row(8 * sl + e, lane >> 5), in = lane & 31;                  // contraction over the layer's OUTPUT channel
      float w = mlp[((layer & 1) ? ml.blk_w1(layer >> 1) : ml.blk_w0(layer >> 1)) + kout * HD + in] * scale;
      asm("" : "+v"(w));
      const _Float16 hi = (_Float16)w;
      _Float16* dst = wb + layer * L::BLK_LAYER_STRIDE * 2 + (sl * 64 + lane) * 8 + e;
      dst[0] = hi;
      dst[L::BLK_TERM_STRIDE * 2] = (_Float16)(w - (float)hi);
    }
  }
}

// hidden_layer_h (bts_render_kernel.h) for ONE point tile: out += (W 2^S) . relu(in) 2^-S.  The tiles are independent columns of the
// product and the three split terms enter in hidden_layer_h's order: bit-identical to the two-tile form.
__device__ __forceinline__ void hidden_layer_h1(f32x16& out, const f32x16& in, const float* wl /* lane-resolved, this layer */, int term_stride,
                                                float inv_scale) {
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    const h8 ah = *reinterpret_cast<const h8*>(wl + sl * 256);
    const h8 al = *reinterpret_cast<const h8*>(wl + term_stride + sl * 256);
    float vc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float v = __builtin_amdgcn_fmed3f(in[8 * sl + i], 0.0f, 3.4028234663852886e38f) * inv_scale;
      vc[i] = fminf(v, 6.0e4f);
      asm("" : "+v"(vc[i]));   // opaque fp32 value: both conversions of the split must see the same rounding (see f16_region)
    }
    const float neg1 = opaque_neg1();
    unsigned uh[4], ul[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split2_f16(vc[2 * j], vc[2 * j + 1], neg1, uh[j], ul[j]);
    const h8 bh = __builtin_bit_cast(h8, (u32x4){uh[0], uh[1], uh[2], uh[3]});
    const h8 bl = __builtin_bit_cast(h8, (u32x4){ul[0], ul[1], ul[2], ul[3]});
    out = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, out, 0, 0, 0);
    out = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, out, 0, 0, 0);
    out = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, out, 0, 0, 0);
  }
}

// out += (W^T 2^S) . (in in_mul) for ONE point tile of one transposed ResnetBlockFC linear of width 32 on the f16 pipe (split precision,
// the backward twin of hidden_layer_h): the lane's own 16 values of the tile in the C layout are the B operand of two 16-row k-slices.
__device__ __forceinline__ void hidden_layer_ht(f32x16& out, const f32x16& in, const float* wl /* lane-resolved, this layer */, int term_stride,
                                                float in_mul) {
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    const h8 ah = *reinterpret_cast<const h8*>(wl + sl * 256);
    const h8 al = *reinterpret_cast<const h8*>(wl + term_stride + sl * 256);
    float vc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      vc[i] = __builtin_amdgcn_fmed3f(in[8 * sl + i] * in_mul, -6.0e4f, 6.0e4f);
      asm("" : "+v"(vc[i]));   // opaque fp32 value: both conversions of the split must see the same rounding (see f16_region)
    }
    const float neg1 = opaque_neg1();
    unsigned uh[4], ul[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split2_f16(vc[2 * j], vc[2 * j + 1], neg1, uh[j], ul[j]);
    const h8 bh = __builtin_bit_cast(h8, (u32x4){uh[0], uh[1], uh[2], uh[3]});
    const h8 bl = __builtin_bit_cast(h8, (u32x4){ul[0], ul[1], ul[2], ul[3]});
    out = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, out, 0, 0, 0);
    out = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, out, 0, 0, 0);
    out = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, out, 0, 0, 0);
  }
}

__device__ __forceinline__ void wave_lds_fence() {   // lanes exchange data through LDS without a barrier: pin the order for the compiler
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sum over the 32 lanes of each wave half of 16 registers at once (see bts_bwd_rows.hip): afterwards both lanes of the pair
// (col, col ^ 1) hold the total of register col >> 1
__device__ __forceinline__ float half_reduce16(float (&w)[16], int col) {
  int d = 16;
#pragma unroll
  for (int n = 16; n > 1; n >>= 1) {
    const bool up = (col & d) != 0;
#pragma unroll
    for (int j = 0; j < n / 2; ++j) {
      float lo = w[j], hi = w[n / 2 + j];
      asm("" : "+v"(lo), "+v"(hi));   // opaque values: keep the selects on values, not on a dynamically indexed array
      const float keep = up ? hi : lo;
      const float give = up ? lo : hi;
      w[j] = keep + __shfl_xor(give, d, 64);
    }
    d >>= 1;
  }
  return w[0] + __shfl_xor(w[0], 1, 64);
}

// Cold path (see eval_point_exact in bts_render_kernel.h): some sample's encoding argument leaves the fast sincos range.  lin_in's
// output h0 of the wave's 64 samples with libm sines and fp32-input MFMAs, as the forward evaluated it, written as
// tile[point of tile PT][hidden] (leading dimension HD + 1) -- one point tile per call: the tile lives in the wave's gather ring.
// Out of line: inlined, its 36 libm sines cost the hot path > 200 spilled VGPRs.
template <int C, int HD, int NB>
__device__ __attribute__((noinline)) void lin_in_exact(const float* lb, const float4* G, const float* w2c, const float* Kc, int H, int W,
                                                       int fs, int code_mode, int inv_z, float inv_dmax, float inv_range, float d_min, float range,
                                                       float freq_factor, int learn_empty, float px, float py, float pz, float* tile, int pt_sel) {
  using L = LdsB<HD, NB>;
  constexpr int HT = HD / 32;
  const int lane = threadIdx.x & 63, h = lane >> 5, col = lane & 31;
  const Cam enc = load_cam(w2c, Kc);
  const Proj pe = code_mode == 1 ? project<true>(enc, px, py, pz) : project<false>(enc, px, py, pz);
  const Taps tp = make_taps(pe.x, pe.y, H, W, fs);
  float v3[3];
  v3[0] = pe.x, v3[1] = pe.y;
  v3[2] = depth_code(pe, code_mode == 1, inv_z != 0, inv_dmax, inv_range, d_min, range);
  const bool use_empty = (learn_empty != 0) & pe.invalid;
  f32x16 acc[HT][2];
#pragma unroll
  for (int ht = 0; ht < HT; ++ht)
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) acc[ht][pt] = zero_acc();
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
  {
    GBuf ga, gb;
    gload<HD>(ga, G, o[0], 0, 4 * h);
    gather_seq<HD, 0>(acc, ga, gb, G, o, wq, h);
  }
  if (learn_empty && __any(use_empty)) apply_empty<HD>(acc, emp, lb + L::EMPTY, h);
  const float* wl = lb + L::W_IN + h * HD + col;
  kstep<HD>(acc, wl, 0, v3[0], v3[1]);
  kstep<HD>(acc, wl + 2 * HD, 0, v3[2], 1.0f);
  wl += 4 * HD;
  float ff = freq_factor;
#pragma unroll 1
  for (int oct = 0; oct < kNumFreqs; ++oct) {
    float sc[6];
    pe_octave(sc, v3, ff);
    kstep<HD>(acc, wl, 0, sc[0], sc[1]);
    kstep<HD>(acc, wl + 2 * HD, 0, sc[2], sc[3]);
    kstep<HD>(acc, wl + 4 * HD, 0, sc[4], sc[5]);
    wl += 6 * HD;
    ff = ff * 2.0f;
  }
#pragma unroll
  for (int ht = 0; ht < HT; ++ht)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float v = pt_sel ? acc[ht][1][q] : acc[ht][0][q];
      tile[col * (HD + 1) + ht * 32 + mfma_row(q, h)] = v;
    }
}

struct RowsbOut {
  float* u0_ws;   // (n*Bp, K, HD) g_h0 = gradient at lin_in's output, channels in the storage order of G
#ifdef BTS_TICKS
  unsigned long long* ticks;   // diagnostic build: [waves][16] cycles per section of the iteration (tools/rowsb_ticks.py)
#endif
};
// diagnostic build (-DBTS_TICKS, behindthescenes_amd/variants): s_memtime at the section borders of rowsb_kernel.  Reading the counter
// drains lgkmcnt, so the sections are somewhat longer than in the product; their shares are what the numbers are for.
#ifdef BTS_TICKS
#define RB_TICK(i)                                                   \
  {                                                                  \
    const unsigned long long t_now = __builtin_readcyclecounter();   \
    t_acc[i] += t_now - t_last;                                      \
    t_last = t_now;                                                  \
  }
#else
#define RB_TICK(i)
#endif

// ---------------------------------------------------------------------------------------------------------------
// pass A
// ---------------------------------------------------------------------------------------------------------------
// PK: the 48-lane mode of render_kernel_p for the backward (32 < K <= 48 and the rays of a batch element a multiple of four --
// exp_re10k.yaml's n_coarse = 48): a group is FOUR rays taken in THREE iterations, walked back to front like the chunks of a long ray --
// lanes 0-47 one whole ray (ray it of the group) per iteration, lanes 48-63 the 16-sample row `it` of the fourth -- so that no
// instruction runs with a quarter of its lanes idle.  Everything per sample is lane-local; what belongs to a RAY (origin, direction,
// upstream gradients: wave-uniform scalars in the one-ray mode) becomes a per-lane select between the iteration's two rays, the suffix
// sum of the compositing gradient is segmented (lanes 0-47 | 48-63), and the fourth ray's carries over its three rows.
template <int C, int HD, int NB, int NVMAX, bool PK = false>
__global__ __launch_bounds__(256, 2) void rowsb_kernel(const BwdParams bp, const RowsbOut ro) {
  static_assert(NB == 0 || HD == 32, "ResnetBlockFC layers are laid out for d_hidden = 32 (the RE10K model)");
  const FwdParams& p = bp.f;
  using L = LdsB<HD, NB>;
  using LH = LdsH<C, HD, NB>;
  constexpr int HT = HD / 32;
  constexpr int NS = 4 * HT;
  constexpr int LB_PAD = (L::TOTAL + 3) & ~3;
  __shared__ __attribute__((aligned(16))) float lds[LB_PAD + LH::TOTAL + 4];
  float* const lh = lds + LB_PAD;   // 16-byte aligned: the f16 A operands are read as ds_read_b128
  for (int hid = threadIdx.x; hid < HD; hid += blockDim.x) {   // the projected empty feature first: stage_weights_h scales it
    float a = 0.0f;
    if (p.empty_feature)
      for (int c = 0; c < C; ++c) a = __builtin_fmaf(p.mlp[hid * (C + kPeDim) + c], p.empty_feature[c], a);
    lds[L::EMPTY + hid] = a;
  }
  __syncthreads();
  stage_weights_h<C, HD, NB>(lh, lds + L::EMPTY, p.mlp);
  __syncthreads();
  const float scale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lh[LH::SCALE])));
  const float inv_scale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lh[LH::SCALE + 1])));
  stage_weights_b<C, HD, NB>(lds, p.mlp, p.empty_feature, scale);
  __syncthreads();
  const float s_v = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lds[L::VSCALE])));
  const float inv_s_v = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lds[L::VSCALE + 1])));

  const int lane = threadIdx.x & 63;
  const int h0 = lane >> 5;
  const int col = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  extern __shared__ __attribute__((aligned(128))) char gather_lds[];   // per wave: ring of 3 x 4 KB + 768 B tap table (render_kernel_p)
  GatherLds gl;
  {
    char* base = gather_lds + wave * kGatherLdsPerWave;
    gl.ring = base;
    gl.ring_m0 = (unsigned)(unsigned long)base;
    gl.tab = reinterpret_cast<unsigned*>(base + 3 * 4096);
    gl.m = lane >> 3;
    gl.piece16 = 16u * (unsigned)(((lane & 7) + (lane >> 4)) & 7);
    gl.piece16x = gl.piece16 ^ 64u;
#pragma unroll
    for (int q = 0; q < 4; ++q) gl.rd[q] = (unsigned)(col * 128 + ((4 * h0 + q - (col >> 1)) & 7) * 16);
  }
  // the ring is idle between the last gather block of a ray and the first of the next: its memory serves as the transposition
  // tiles of the weight-gradient contraction ([32 points][33] x 2) and as the cold path's h0 tile ([32 points][HD + 1])
  float* const tile_a = reinterpret_cast<float*>(gather_lds + wave * kGatherLdsPerWave);
  float* const tile_b = tile_a + 32 * 33;
  static_assert(2 * 32 * 33 * 4 <= 3 * 4096 && 32 * (HD + 1) * 4 <= 3 * 4096, "tiles must fit the wave's gather ring");

  const int nwg = gridDim.x;  // multiple of 8
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int wg_per_xcd = nwg >> 3;
  const int xcd = wg / wg_per_xcd;
  const int lw = (wg - xcd * wg_per_xcd) * 4 + wave;
  const int waves_per_xcd = wg_per_xcd * 4;
  // chunk-interleaved ray distribution over the XCDs, as render_kernel_p (one group = one ray here)
  const int CHL = p.chunk_log2;
  const int n_groups = (int)p.groups;
  const int n_chunks = (n_groups + (1 << CHL) - 1) >> CHL;
  auto group_of = [&](int idx) -> int {
    const int c = ((idx >> CHL) << 3) + xcd;
    const int gg = (c << CHL) + (idx & ((1 << CHL) - 1));
    return (c < n_chunks && gg < n_groups) ? gg : -1;
  };
  const int Bp = p.Bp, K = p.K;
  const int kc_last = PK ? 128 : ((K - 1) >> 6) << 6;   // first sample of the last 64-sample chunk of a ray: chunks are walked back to front
                                                        // (PK: kc >> 6 is the iteration of the group, 2 .. 0)
  const bool mainl = !PK || lane < 48;
  // this lane's sample in iteration kc: its index in the ray (may be >= K: an idle lane) ...
  auto k_of = [&](int kc) -> int { return PK ? (mainl ? lane : 16 * (kc >> 6) + lane - 48) : kc + lane; };
  // ... and the row of (ray, sample) tables it indexes, relative to the group's first ray
  auto row_of = [&](int kc, int kk) -> unsigned { return PK ? (unsigned)((mainl ? (kc >> 6) : 3) * K + kk) : (unsigned)kk; };

  // persistent per-wave gradient state
  float dw_acc[HT], db_acc = 0.0f;   // this lane's share of dw_out (see dw_out below) and of db_out
#pragma unroll
  for (int ht = 0; ht < HT; ++ht) dw_acc[ht] = 0.0f;
  f32x16 dwb[NB > 0 ? NB : 1][2];    // per block: dW0 [out][in], dW1 [out][in] as 32x32 accumulator tiles
  float dbb[NB > 0 ? NB : 1][2];     // per block: this lane's share of db0 / db1 (channel col, samples of its lane half)
#pragma unroll
  for (int b = 0; b < (NB > 0 ? NB : 1); ++b)
#pragma unroll
    for (int j = 0; j < 2; ++j) dwb[b][j] = zero_acc(), dbb[b][j] = 0.0f;

  int sample_end = Bp;
  int sample = 0;
  int idx = lw;
  int g = group_of(idx);
  float z_pre = 0.0f, zn_pre = 0.0f, s_pre = 0.0f, t_pre = 0.0f;
  // (per-sample tensors are addressed as a wave-uniform row base -- the ray's first sample, an SGPR pair -- plus a 32-bit lane offset:
  // 64-bit per-lane addresses computed ahead of their loads were what this kernel spilled, and a reload from scratch waits with
  // vmcnt(0): the four prefetch loads below ran as four serial memory round trips per iteration)
  // (`f`: the parameter block as laundered for this iteration -- through the by-value copy `p` the compiler hoists the lane parts of the
  // four addresses out of the loop, keeps them as 64-bit VGPR pairs and spills exactly those)
  auto fetch_state = [&](const FwdParams __attribute__((address_space(4)))* f, int gg, int kc) {   // the per-sample state of chunk kc of ray (group) gg
    const int k = k_of(kc);
    const int kk = k < K ? k : K - 1;
    const long row = (long)gg * (PK ? 4 : 1) * K;   // uniform
    const unsigned r = row_of(kc, kk);
    z_pre = at32(f->z_samp + row, r), zn_pre = at32(f->z_samp + row, kk + 1 < K ? r + 1u : r);
    s_pre = at32(f->sigma_raw + row, r), t_pre = at32(f->trans + row, r);
  };
  if (g >= 0) fetch_state(&kernarg_view<BwdParams>()->f, g, kc_last);
  // the rays' wave-uniform inputs one iteration ahead (bts_bwd.h: fetch_ray_record): `rec` the group's ray (PK: its fourth ray), `recm`
  // (PK) the ray lanes 0-47 take in the next iteration
  const int nv3 = p.nv * 3;
  float rec = 0.0f, recm = 0.0f;
  if (g >= 0) {
    rec = fetch_ray_record(kernarg_view<BwdParams>(), PK ? (long)g * 4 + 3 : (long)g, nv3, lane);
    if (PK) recm = fetch_ray_record(kernarg_view<BwdParams>(), (long)g * 4 + 2, nv3, lane);
  }
#ifdef BTS_TICKS
  unsigned long long t_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last = __builtin_readcyclecounter();
  const unsigned long long t_begin = t_last;
  unsigned n_iter = 0;
#endif

  for (; g >= 0; idx += waves_per_xcd, g = group_of(idx)) {
    auto qb = kernarg_view<BwdParams>();   // this iteration's parameters, re-read where they are used (bts_common.h: kernarg_view)
    asm volatile("" : "+s"(qb));
    IterHead ih(qb);   // (batching these scalar loads -- IterHeadT<true> -- changes nothing here either: 0.640 vs 0.634 ms, profiles/r03_experiments/r03s)
    const int H = ih.H, W = ih.W, nv = ih.nv, fs = ih.fs;
    const long ray0 = PK ? (long)g * 4 : (long)g;   // the (first) ray of the group
    while (ray0 >= sample_end) ++sample, sample_end += Bp;
    const Cam enc = load_cam(ih.w2c_enc + sample * 16, ih.K_enc + sample * 9);
    const float4* __restrict__ G = reinterpret_cast<const float4*>(ih.proj) + (long)sample * (H >> fs) * (W >> fs) * (HD / 4);
    // the ray's scalars (PK: the fourth ray's, selected into lanes 48-63 of every iteration below), fetched during the previous group;
    // the next group's go out now
    const RayIn<NVMAX * 3> rin = unpack_ray_record<NVMAX * 3>(qb, rec, nv3);
    const int g_nxg = group_of(idx + waves_per_xcd);
    if (g_nxg >= 0) rec = fetch_ray_record(qb, PK ? (long)g_nxg * 4 + 3 : (long)g_nxg, nv3, lane);
    float S_carry = 0.0f;   // sum over the samples of the chunks behind this one of g_w w (PK: of the fourth ray's rows behind this one)
    RB_TICK(10)   // (part of 0) the ray's scalars are in

    for (int kc = kc_last; kc >= 0; kc -= 64) {
      const int k = k_of(kc);
      const bool valid = k < K;
      const int kk = valid ? k : K - 1;
      const bool last = k == K - 1;
      const long rowk = ray0 * K;         // uniform: the first sample of the group's first ray
      const unsigned kku = row_of(kc, kk);
      const float z = z_pre, z_nx = zn_pre, s_raw = s_pre, T = t_pre;
      float ox, oy, oz, dx, dy, dz, g_rgb[NVMAX * 3], g_bkgd, g_depth;
      if constexpr (PK) {
        const RayIn<NVMAX * 3> rm = unpack_ray_record<NVMAX * 3>(qb, recm, nv3);
        const long r_nx = kc > 0 ? ray0 + (kc >> 6) - 1 : (g_nxg >= 0 ? (long)g_nxg * 4 + 2 : -1);
        if (r_nx >= 0) recm = fetch_ray_record(qb, r_nx, nv3, lane);
        ox = mainl ? rm.o[0] : rin.o[0], oy = mainl ? rm.o[1] : rin.o[1], oz = mainl ? rm.o[2] : rin.o[2];
        dx = mainl ? rm.d[0] : rin.d[0], dy = mainl ? rm.d[1] : rin.d[1], dz = mainl ? rm.d[2] : rin.d[2];
#pragma unroll
        for (int i = 0; i < NVMAX * 3; ++i) g_rgb[i] = mainl ? rm.g_rgb[i] : rin.g_rgb[i];
        g_bkgd = mainl ? rm.g_bkgd : rin.g_bkgd, g_depth = mainl ? rm.g_depth : rin.g_depth;
      } else {
        ox = rin.o[0], oy = rin.o[1], oz = rin.o[2], dx = rin.d[0], dy = rin.d[1], dz = rin.d[2];
#pragma unroll
        for (int i = 0; i < NVMAX * 3; ++i) g_rgb[i] = rin.g_rgb[i];
        g_bkgd = rin.g_bkgd, g_depth = rin.g_depth;
      }
      {  // the state of the chunk evaluated next lands while this one is evaluated (ONE call site: two get tail-merged into a block
         // that rebuilds 64-bit lane addresses from spilled parts)
        const int g_nx = kc > 0 ? g : group_of(idx + waves_per_xcd);
        const int kc_nx = kc > 0 ? kc - 64 : kc_last;
        if (g_nx >= 0) fetch_state(&qb->f, g_nx, kc_nx);
      }
      int h = h0;
      asm volatile("" : "+v"(h));   // keep the weight reads inside the persistent loop (see render_kernel_p)
      const float px = ox + z * dx, py = oy + z * dy, pz = oz + z * dz;

      // ---------------- the forward's per-sample colours (and the optional per-sample upstream gradients): loads issued here, used behind
      // the geometry and the first gather blocks -- read where they are needed, each is a full memory round trip with the wave idle.
      // (Issuing them one chunk ahead, in front of the previous chunk's row stores -- vmcnt retires in order, a load behind the stores
      // is back when they are -- was tried: the nine registers it keeps alive cost more in spills than the wait, 0.676 vs 0.629 ms.)
      float cs_v[NVMAX * 3];
#ifdef BTS_ABL_B1   // timing ablation: no per-sample colour loads
      const bool have_cs = false;
#else
      const bool have_cs = qb->f.rgb_samps != nullptr;
#endif
#pragma unroll
      for (int i = 0; i < NVMAX * 3; ++i) cs_v[i] = 0.0f;
      if (have_cs) {
        const float* cs = qb->f.rgb_samps + rowk * (long)(nv * 3);   // uniform
        const unsigned co = kku * (unsigned)(nv * 3);
#pragma unroll
        for (int i = 0; i < NVMAX * 3; ++i) cs_v[i] = at32(cs, co + (unsigned)min(i, nv * 3 - 1));   // entries beyond nv meet g_rgb = 0
      }
      const float gw_k = qb->g_weights ? at32(qb->g_weights + rowk, kku) : 0.0f;
      const float ga_k = qb->g_alphas ? at32(qb->g_alphas + rowk, kku) : 0.0f;

      RB_TICK(11)   // (part of 0) per-sample loads issued
      // ---------------- encoder view
      const Proj pe = ih.code_mode == 1 ? project<true>(enc, px, py, pz) : project<false>(enc, px, py, pz);
      Taps tp = make_taps(pe.x, pe.y, H, W, fs);
      float v3[3];
      v3[0] = pe.x, v3[1] = pe.y;
      v3[2] = depth_code(pe, ih.code_mode == 1, ih.inv_z != 0, ih.inv_dmax, ih.inv_range, ih.d_min, ih.range);
      const bool use_empty = (ih.learn_empty != 0) & pe.invalid;
      if (use_empty) tp.w00 = tp.w01 = tp.w10 = tp.w11 = 0.0f;
      tp.w00 *= scale, tp.w01 *= scale, tp.w10 *= scale, tp.w11 *= scale;

      float wq[2][4];
      bool emp[2];
      {
        unsigned t0, t1;
        bcast_tiles(__float_as_uint(tp.w00), t0, t1), wq[0][0] = __uint_as_float(t0), wq[1][0] = __uint_as_float(t1);
        bcast_tiles(__float_as_uint(tp.w01), t0, t1), wq[0][1] = __uint_as_float(t0), wq[1][1] = __uint_as_float(t1);
        bcast_tiles(__float_as_uint(tp.w10), t0, t1), wq[0][2] = __uint_as_float(t0), wq[1][2] = __uint_as_float(t1);
        bcast_tiles(__float_as_uint(tp.w11), t0, t1), wq[0][3] = __uint_as_float(t0), wq[1][3] = __uint_as_float(t1);
        bcast_tiles(use_empty ? 1u : 0u, t0, t1), emp[0] = t0 != 0, emp[1] = t1 != 0;
      }
      RB_TICK(12)   // (part of 0) projection, taps, tile broadcast
      const bool cold = __any(pe_needs_exact(v3, ih.freq_factor));
      unsigned off_next[4];
      GRows rows;
      if (__builtin_expect(!cold, 1)) {
        // the previous iteration's tile reads have returned (s_waitcnt at its end): the ring may be overwritten
        wave_lds_fence();
        gl.tab[lane * 3 + 0] = (unsigned)tp.o00 * (HD * 4u), gl.tab[lane * 3 + 1] = (unsigned)tp.o01 * (HD * 4u), gl.tab[lane * 3 + 2] = (unsigned)tp.o10 * (HD * 4u);   // o11 = o10 + (o01 - o00)
        wave_lds_fence();
        gl_prologue<HD>(gl, rows, G, off_next);
      }
      RB_TICK(0)   // ray / gradient loads issued, geometry, taps, table, first three blocks out

      // ---------------- upstream gradient of this sample's weight: g_w = g_depth z + sum_j g_rgb_j . c_kj (+ g_weights_k)
      float g_w = g_depth * z;
      {
        if (qb->f.white_bkgd) g_w += g_bkgd;   // nerf.py:301-304: rgb = sum_k w_k c_k + 1 - sum_k w_k
        g_w += gw_k;
        if (have_cs) {
#pragma unroll
          for (int j = 0; j < NVMAX; ++j)
            g_w += g_rgb[3 * j] * cs_v[3 * j] + g_rgb[3 * j + 1] * cs_v[3 * j + 1] + g_rgb[3 * j + 2] * cs_v[3 * j + 2];
        } else {
#pragma unroll
          for (int j = 0; j < NVMAX; ++j) {
            if (j < nv) {
              const Cam cj = load_cam(qb->f.w2c_r + ((long)sample * nv + j) * 16, qb->f.K_r + ((long)sample * nv + j) * 9);
              const Proj pc = project<false>(cj, px, py, pz);
              const Taps tc = make_taps(pc.x, pc.y, H, W);
              const float4* img = reinterpret_cast<const float4*>(qb->f.imgs) + ((long)sample * nv + j) * H * W;
              const float4 a = img[tc.o00], b = img[tc.o01], cc = img[tc.o10], d = img[tc.o11];
              const float c0 = ((a.x * tc.w00 + b.x * tc.w01) + cc.x * tc.w10) + d.x * tc.w11;
              const float c1 = ((a.y * tc.w00 + b.y * tc.w01) + cc.y * tc.w10) + d.y * tc.w11;
              const float c2 = ((a.z * tc.w00 + b.z * tc.w01) + cc.z * tc.w10) + d.z * tc.w11;
              g_w += g_rgb[3 * j] * c0 + g_rgb[3 * j + 1] * c1 + g_rgb[3 * j + 2] * c2;
            }
          }
        }
      }


      // ---------------- compositing gradient (nerf.py:283-299):  g_alpha_k = g_w_k T_k - (sum_{m>k} g_w_m w_m) / (1 - alpha_k + 1e-10)
      float g_s = 0.0f;
      {
        float sigma = softplus(s_raw);
        const bool dead = (qb->f.empty_empty != 0) & pe.invalid;   // sigma forced to 0: no gradient
        if (dead) sigma = 0.0f;
        if (qb->f.sigma_noise) sigma += at32(qb->f.sigma_noise + rowk, kku);   // nerf.py:279-280: relu(sigma + noise) -- no gradient where the sum is <= 0
        const bool cut = sigma <= 0.0f && qb->f.sigma_noise != nullptr;
        const float delta = last ? 1e10f : (z_nx - z);
        const float ex = transmittance(delta, sigma);
        const bool capped = (qb->f.hard_cap != 0) & last;
        const float alpha = capped ? 1.0f : 1.0f - ex;
        const float gww = valid ? g_w * (alpha * T) : 0.0f;
        // exclusive suffix sum over the wave + the chunks behind; the wave's total moves on to the chunk in front
        float incl = gww;
#ifndef BTS_ABL_B6   // timing ablation: no suffix scan
        const int seg_end = mainl && PK ? 48 : 64;   // one past the last lane of this lane's ray
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const float y = __shfl_down(incl, off, 64);
          incl += (lane + off < seg_end) ? y : 0.0f;
        }
#endif
        const float below = __shfl_down(incl, 1, 64);
        const float S = (lane == seg_end - 1 ? 0.0f : below) + (mainl && PK ? 0.0f : S_carry);
        S_carry += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, incl), PK ? 48 : 0));
        float g_alpha = g_w * T - S / (capped ? 1e-10f : ex + 1e-10f);
        g_alpha += ga_k;
        if (!capped && !dead && !cut && valid) g_s = g_alpha * fabsf(delta) * ex * (s_raw > 20.0f ? 1.0f : sigmoidf(s_raw));
        if (valid) at32(qb->gs_ws + rowk, kku) = g_s;
      }
      db_acc += g_s;
      RB_TICK(1)   // compositing gradient (waits for the per-sample loads above)

      // ---------------- h0 = bilinear(G) + W_pe . PE + b, exactly as render_kernel_p evaluates it (accumulators carry 2^S)
      f32x16 acc[HT][2];
      if (__builtin_expect(cold, 0)) {
        // no gather is in flight (the prologue above was skipped).  One point tile per call: the tile lives in the wave's gather ring
        wave_lds_fence();
        lin_in_exact<C, HD, NB>(lds, G, qb->f.w2c_enc + sample * 16, qb->f.K_enc + sample * 9, H, W, fs, qb->f.code_mode, qb->f.inv_z, qb->f.inv_dmax, qb->f.inv_range,
                                qb->f.d_min, qb->f.range, qb->f.freq_factor, qb->f.learn_empty, px, py, pz, tile_a, 0);
        wave_lds_fence();
#pragma unroll
        for (int ht = 0; ht < HT; ++ht)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[ht][0][q] = tile_a[col * (HD + 1) + ht * 32 + mfma_row(q, h)] * scale;
        wave_lds_fence();
        lin_in_exact<C, HD, NB>(lds, G, qb->f.w2c_enc + sample * 16, qb->f.K_enc + sample * 9, H, W, fs, qb->f.code_mode, qb->f.inv_z, qb->f.inv_dmax, qb->f.inv_range,
                                qb->f.d_min, qb->f.range, qb->f.freq_factor, qb->f.learn_empty, px, py, pz, tile_a, 1);
        wave_lds_fence();
#pragma unroll
        for (int ht = 0; ht < HT; ++ht)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[ht][1][q] = tile_a[col * (HD + 1) + ht * 32 + mfma_row(q, h)] * scale;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else {
        f32x16 bias[HT];
        {
          const float* bl = lh + LH::W_RAW + 3 * HD + 4 * h;
#pragma unroll
          for (int ht = 0; ht < HT; ++ht)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float4 v = *reinterpret_cast<const float4*>(bl + ht * 32 + 8 * j);
              bias[ht][4 * j + 0] = v.x, bias[ht][4 * j + 1] = v.y, bias[ht][4 * j + 2] = v.z, bias[ht][4 * j + 3] = v.w;
            }
        }
        SinCos3 raw;
        pe_direct(raw, v3, ih.freq_factor);
        __builtin_amdgcn_sched_barrier(0);
        int lane4 = lane * 4;
        asm volatile("" : "+v"(lane4));
        region_seq_l<HD, 0>(acc, gl, rows, G, wq, off_next, lh + LH::W_F16 + lane4, LH::TERM_STRIDE, raw, v3, ih.freq_factor, bias);
        if constexpr (NS > kNumFreqs) {
          gl_consume<HD, 12>(acc, gl, rows, G, wq, off_next), gl_consume<HD, 13>(acc, gl, rows, G, wq, off_next);
          gl_consume<HD, 14>(acc, gl, rows, G, wq, off_next), gl_consume<HD, 15>(acc, gl, rows, G, wq, off_next);
        }
        if (qb->f.learn_empty && __any(use_empty)) {
#pragma unroll
          for (int ht = 0; ht < HT; ++ht)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              const float ev = lh[LH::EMPTY + ht * 32 + mfma_row(q, 0) + 4 * h];
#pragma unroll
              for (int pt = 0; pt < 2; ++pt) acc[ht][pt][q] += emp[pt] ? ev : 0.0f;
            }
        }
      }

      RB_TICK(2)   // forward pipeline: gather + encoding + lin_in
      // ---------------- g_s of both point tiles
      float gs_t[2], gs_v[2];
      {
        unsigned t0, t1;
        bcast_tiles(__float_as_uint(g_s), t0, t1);
        gs_t[0] = __uint_as_float(t0), gs_t[1] = __uint_as_float(t1);
        gs_v[0] = gs_t[0] * inv_s_v, gs_v[1] = gs_t[1] * inv_s_v;   // exact (a power of two): the vectors below carry s_v
      }
      // ---------------- one point tile (32 samples) at a time, forward through the ResnetBlockFC layers AND back: the block's input
      // h0, its inner activation n, its output h1 and the transient vectors of the backward are 16 registers each instead of 32 (both
      // tiles at once cost 64 spilled VGPRs and 150 scratch accesses per ray).  The tiles are independent columns of every product, so
      // the values -- and the relu gates -- are those of render_kernel_p bit for bit.
      // Backward of a tile: v = m1 . w_out, then back through the blocks (resnetfc.py:53-62: h1 = h0 + fc_1(relu(n)), n = fc_0(relu(h0))),
      // the fc_0 / fc_1 weight gradients of the tile, and u0 = g_s v0, the gradient row at lin_in's output
      int lane4b = lane * 4;
      asm volatile("" : "+v"(lane4b));
      int lane4t = lane * 4;
      asm volatile("" : "+v"(lane4t));   // opaque per iteration: the A operands of the transposed products stay inside the persistent loop
      float r_out[HT][16];   // dw_out += relu(h1) g_s (2^S removed through g_s), summed over the two tiles before the butterfly
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        f32x16 hin[NB > 0 ? NB : 1], net[NB > 0 ? NB : 1];
        if constexpr (NB > 0) {
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            hin[b] = acc[0][pt];
            f32x16 nt;
#pragma unroll
            for (int q = 0; q < 16; ++q) nt[q] = lh[LH::BIAS + b * 2 * HD + mfma_row(q, 0) + 4 * h];
            hidden_layer_h1(nt, acc[0][pt], lh + LH::W_BLK + (2 * b) * LH::BLK_LAYER_STRIDE + lane4b, LH::BLK_TERM_STRIDE, inv_scale);
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[0][pt][q] += lh[LH::BIAS + b * 2 * HD + HD + mfma_row(q, 0) + 4 * h];
            hidden_layer_h1(acc[0][pt], nt, lh + LH::W_BLK + (2 * b + 1) * LH::BLK_LAYER_STRIDE + lane4b, LH::BLK_TERM_STRIDE, inv_scale);
            net[b] = nt;
          }
        }
        RB_TICK(3)   // block forward of the tile
#pragma unroll
        for (int ht = 0; ht < HT; ++ht)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            // (the order of the two-tile form: relu(h1[1]) g_s[1] + relu(h1[0]) g_s[0] as one fma on top of the first product)
            if (pt == 0) r_out[ht][q] = relu1(acc[ht][0][q]) * (gs_t[0] * inv_scale);
            else r_out[ht][q] = __builtin_fmaf(relu1(acc[ht][1][q]), gs_t[1] * inv_scale, r_out[ht][q]);
          }
        if (pt == 1) {   // both tiles are in: the butterfly now, so that its 16 registers are free during this tile's backward
          // dw_acc[ht] belongs to channel ht*32 + mfma_row(col >> 1, h), on both lanes of the pair
#pragma unroll
          for (int ht = 0; ht < HT; ++ht) {
#ifdef BTS_ABL_B4   // timing ablation: no butterfly
            dw_acc[ht] += r_out[ht][0] + r_out[ht][5] + r_out[ht][15];
#else
            dw_acc[ht] += half_reduce16(r_out[ht], col);
#endif
          }
          RB_TICK(4)   // dw_out
        }
        f32x16 v[HT];   // the gate-dependent vector of the current layer times s_v (g_h = (g_s / s_v) v)
#pragma unroll
        for (int ht = 0; ht < HT; ++ht)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const float w2 = lds[L::W_OUT + ht * 32 + mfma_row(q, 0) + 4 * h] * s_v;
            v[ht][q] = acc[ht][pt][q] > 0.0f ? w2 : 0.0f;
          }
        if constexpr (NB > 0) {
#pragma unroll
          for (int b = NB - 1; b >= 0; --b) {
            const float* wt = lds + L::BLK + b * L::BLK_STRIDE + lane4t;
            // vn = mn . (W1^T v): the C layout of v is the B operand; the product carries 2^S (and s_v)
            f32x16 vn = zero_acc();
#ifdef BTS_ABL_B5   // timing ablation: no transposed products
            vn = v[0];
#else
            hidden_layer_ht(vn, v[0], wt + L::BLK_LAYER_STRIDE, L::BLK_TERM_STRIDE, 1.0f);
#endif
#pragma unroll
            for (int q = 0; q < 16; ++q) vn[q] = net[b][q] > 0.0f ? vn[q] * inv_scale : 0.0f;
            RB_TICK(5)   // v, vn = mn . W1^T v
#ifdef BTS_ABL_B3   // timing ablation: no fc_0 / fc_1 weight gradients
            if (false) {
#else
            if (qb->d_mlp) {
#endif
              // dW1[out][in] += sum_p (g_s v)[p][out] relu(n)[p][in];  db1[out] += sum_p (g_s v)[p][out]
              wave_lds_fence();
#pragma unroll
              for (int q = 0; q < 16; ++q) {
                tile_a[col * 33 + mfma_row(q, h)] = v[0][q] * gs_v[pt];
                tile_b[col * 33 + mfma_row(q, h)] = relu1(net[b][q]) * inv_scale;
              }
              wave_lds_fence();
#pragma unroll 4
              for (int s2 = 0; s2 < 16; ++s2) {
                const int pnt = 2 * s2 + h;
                const float a = tile_a[pnt * 33 + col];
                dbb[b][1] += a;
                dwb[b][1] = mfma(a, tile_b[pnt * 33 + col], dwb[b][1]);
              }
            }
            RB_TICK(6)   // dW1 tiles
            // t2 = W0^T vn;  v <- v + m0 . t2
            f32x16 t2 = zero_acc();
#ifdef BTS_ABL_B5
            t2 = vn;
#else
            hidden_layer_ht(t2, vn, wt, L::BLK_TERM_STRIDE, 1.0f);
#endif
#ifdef BTS_ABL_B3
            if (false) {
#else
            if (qb->d_mlp) {
#endif
              // dW0[out][in] += sum_p (g_s vn)[p][out] relu(h0)[p][in];  db0[out] += sum_p (g_s vn)[p][out]
              RB_TICK(7)   // t2 = W0^T vn
              wave_lds_fence();
#pragma unroll
              for (int q = 0; q < 16; ++q) {
                tile_a[col * 33 + mfma_row(q, h)] = vn[q] * gs_v[pt];
                tile_b[col * 33 + mfma_row(q, h)] = relu1(hin[b][q]) * inv_scale;
              }
              wave_lds_fence();
#pragma unroll 4
              for (int s2 = 0; s2 < 16; ++s2) {
                const int pnt = 2 * s2 + h;
                const float a = tile_a[pnt * 33 + col];
                dbb[b][0] += a;
                dwb[b][0] = mfma(a, tile_b[pnt * 33 + col], dwb[b][0]);
              }
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) v[0][q] += hin[b][q] > 0.0f ? t2[q] * inv_scale : 0.0f;
            RB_TICK(8)   // dW0 tiles
          }
        }
        // u0 = g_s v: one row per sample in the storage order of G.  In the C layout a lane holds four 16-byte pieces of ITS sample's row:
        // stored from there, every instruction drops 16 bytes into each of 32 different rows and the memory system sees quarter lines
        // (WRITE_SIZE 2.4 x the bytes, and -- vmcnt retiring in order -- everything the next chunk loads waits behind them).  The tile
        // goes through LDS instead, written in the gather ring's own row layout (gl.rd: the inverse of what the forward reads), and
        // comes back as whole rows: eight lanes per 128-byte row, eight full rows per store instruction.
        if (ro.u0_ws) {
#ifdef BTS_ABL_B2   // timing ablation: no row stores
          if (gs_v[pt] == 12345.0f)
#endif
          {
            char* const u0t = gather_lds + wave * kGatherLdsPerWave + (NB > 0 ? 2 * 32 * 33 * 4 : 0);   // behind the contraction tiles
            static_assert((NB > 0 ? 2 * 32 * 33 * 4 : 0) + HT * 4096 <= kGatherLdsPerWave, "u0 tile must fit the wave's ring memory");
            wave_lds_fence();
#pragma unroll
            for (int ht = 0; ht < HT; ++ht)
#pragma unroll
              for (int j = 0; j < 4; ++j)
                *reinterpret_cast<float4*>(u0t + ht * 4096 + gl.rd[j]) =
                    make_float4(v[ht][4 * j] * gs_v[pt], v[ht][4 * j + 1] * gs_v[pt], v[ht][4 * j + 2] * gs_v[pt], v[ht][4 * j + 3] * gs_v[pt]);
            wave_lds_fence();
            // position (jj, m = lane >> 3, lane & 7) of a block holds piece gl.piece16 / 16 (jj even) or gl.piece16x / 16 (jj odd) of row 8 jj + m (GatherLds)
#pragma unroll
            for (int ht = 0; ht < HT; ++ht)
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) {
                const float4 x = *reinterpret_cast<const float4*>(u0t + ht * 4096 + jj * 1024 + lane * 16);
                const int pos = pt * 32 + 8 * jj + gl.m;   // the wave position (= lane) of the sample this row belongs to
                const int ks = PK ? (pos < 48 ? pos : 16 * (kc >> 6) + pos - 48) : kc + pos;
                const int kr = PK ? (pos < 48 ? (kc >> 6) : 3) * K + ks : ks;
                if (ks < K)
                  *reinterpret_cast<float4*>(reinterpret_cast<char*>(ro.u0_ws + rowk * (long)HD) + (unsigned)(kr * HD * 4 + ht * 128) + (jj & 1 ? gl.piece16x : gl.piece16)) = x;
              }
          }
        }
      }
      RB_TICK(9)   // u0 row stores issued
      // the tile reads of this iteration must have returned before the next iteration's gather lands in the ring
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef BTS_TICKS
      ++n_iter;
#endif
    }
  }
#ifdef BTS_TICKS
  if (ro.ticks && lane == 0) {
    unsigned long long* d = ro.ticks + ((long)blockIdx.x * 4 + wave) * 16;
#pragma unroll
    for (int i = 0; i < 13; ++i) d[i] = t_acc[i];
    d[14] = n_iter, d[15] = __builtin_readcyclecounter() - t_begin;
  }
#endif

  // ---------------- flush: wave registers -> work-group LDS (the gather rings, idle now) -> one global atomic per parameter
  __syncthreads();
  if (bp.d_mlp) {
    float* red = reinterpret_cast<float*>(gather_lds);   // [HD + 1] dw_out, db_out; per block: dW0 [HD][HD], db0 [HD], dW1 [HD][HD], db1 [HD]
    constexpr int RED_BLK = HD + 1, RED_BLK_STRIDE = 2 * HD * HD + 2 * HD;
    constexpr int RED_TOTAL = RED_BLK + NB * RED_BLK_STRIDE;
    static_assert(RED_TOTAL * 4 <= 4 * kGatherLdsPerWave, "reduction buffer must fit the work-group's gather rings");
    for (int i = threadIdx.x; i < RED_TOTAL; i += blockDim.x) red[i] = 0.0f;
    __syncthreads();
    if ((lane & 1) == 0) {
#pragma unroll
      for (int ht = 0; ht < HT; ++ht) atomicAdd(&red[ht * 32 + mfma_row(col >> 1, h0)], dw_acc[ht]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) db_acc += __shfl_xor(db_acc, off, 64);
    if (lane == 0) atomicAdd(&red[HD], db_acc);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float* d = red + RED_BLK + b * RED_BLK_STRIDE;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float* dw = d + j * (HD * HD + HD);
#pragma unroll
        for (int q = 0; q < 16; ++q) atomicAdd(&dw[mfma_row(q, h0) * HD + col], dwb[b][j][q]);   // D[i = out][j = in]
        atomicAdd(&dw[HD * HD + col], dbb[b][j]);   // both lane halves hold a share of channel col
      }
    }
    __syncthreads();
    const MlpLayout ml{C + kPeDim, HD, NB};
    for (int i = threadIdx.x; i <= HD; i += blockDim.x) {
      const float vv = red[i];
      if (vv != 0.0f) flush_add_f32(bp.d_mlp + (i < HD ? ml.w_out() + i : ml.b_out()), vv);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float* d = red + RED_BLK + b * RED_BLK_STRIDE;
      for (int i = threadIdx.x; i < RED_BLK_STRIDE; i += blockDim.x) {
        const float vv = d[i];
        if (vv != 0.0f) flush_add_f32(bp.d_mlp + ml.blk(b) + i, vv);   // packed order: w0, b0, w1, b1 = the LDS order
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------------------------------------------
int launch_scatter_rows(const BwdParams& bp, const float* u0_ws, int HD, int n, hipStream_t s);
struct PassQueue;
PassQueue* pass_queue();
hipStream_t pass_fork(PassQueue* pq, hipStream_t s);
void pass_join(PassQueue* pq, hipStream_t side, hipStream_t s);
int launch_dwpe_rows(const FwdParams& p, const float* u0_ws, float* d_mlp, float* flush_ws, int C, int HD, int NB, int n, int grid, hipStream_t s,
                     bool flush_clean);

template <int C, int HD, int NB>
static int launch_rowsb(const BwdParams& bp, const RowsbOut& ro, int grid, hipStream_t s) {
  constexpr int dyn = 4 * kGatherLdsPerWave;
  auto go = [&](auto kern) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
    kern<<<grid, 256, dyn, s>>>(bp, ro);
  };
  if constexpr (NB == 1) {   // the 48-lane mode is built for the RE10K model (rowsb_packs_48 in bts_bwd.hip names the same condition)
    if (bp.f.lpr == 48) {
      if (bp.f.nv <= 1) go(rowsb_kernel<C, HD, NB, 1, true>);
      else if (bp.f.nv <= 2) go(rowsb_kernel<C, HD, NB, 2, true>);
      else if (bp.f.nv <= 4) go(rowsb_kernel<C, HD, NB, 4, true>);
      else go(rowsb_kernel<C, HD, NB, 8, true>);
      return launch_status();
    }
  }
  if (bp.f.lpr != 64) return BTS_E_UNSUPPORTED;
  if (bp.f.nv <= 1) go(rowsb_kernel<C, HD, NB, 1>);
  else if (bp.f.nv <= 2) go(rowsb_kernel<C, HD, NB, 2>);
  else if (bp.f.nv <= 4) go(rowsb_kernel<C, HD, NB, 4>);
  else go(rowsb_kernel<C, HD, NB, 8>);
  return launch_status();
}

// bp.gs_ws: (n*Bp, K) floats; u0_ws: (n*Bp, K, HD) floats; p.groups / chunk_log2 / lpr set for one ray per wave iteration
int launch_bwd_blocks(const BwdParams& bp, float* u0_ws, int C, int HD, int NB, int n, int grid, hipStream_t s) {
  RowsbOut ro;
  ro.u0_ws = (bp.d_proj || bp.d_empty_proj || bp.d_mlp) ? u0_ws : nullptr;
#ifdef BTS_TICKS
  ro.ticks = nullptr;
  if (const char* e = getenv("BTS_DBG_PTR")) ro.ticks = (unsigned long long*)strtoull(e, nullptr, 0);
#endif
  int rc = BTS_E_UNSUPPORTED;
  if (C == 64 && HD == 64 && NB == 0) rc = launch_rowsb<64, 64, 0>(bp, ro, grid, s);
  else if (C == 32 && HD == 32 && NB == 1) rc = launch_rowsb<32, 32, 1>(bp, ro, grid, s);
  else if (C == 32 && HD == 32 && NB == 0) rc = launch_rowsb<32, 32, 0>(bp, ro, grid, s);
  // pass C on a side queue of the library next to pass B (bts_bwd_rows.hip: pass_queue)
  const bool want_b = bp.d_proj || bp.d_empty_proj, want_c = bp.d_mlp != nullptr;
  PassQueue* pq = (rc == BTS_OK && want_b && want_c) ? pass_queue() : nullptr;
  const hipStream_t sc = pq ? pass_fork(pq, s) : s;
  if (rc == BTS_OK && want_c) rc = launch_dwpe_rows(bp.f, u0_ws, bp.d_mlp, bp.flush_ws, C, HD, NB, n, grid, sc, bp.flush_clean);
  if (rc == BTS_OK && want_b) rc = launch_scatter_rows(bp, u0_ws, HD, n, s);
  pass_join(pq, sc, s);
  return rc;
}

}  // namespace bts
