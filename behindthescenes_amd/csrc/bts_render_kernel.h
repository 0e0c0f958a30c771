// The render kernel of the default path (projected feature map G), software-pipelined for gfx950.
//
// lane = SAMPLE, like render_kernel in bts_field_kernel.h (one ray -- or 64/lpr short rays -- per wave iteration, XCD-contiguous
// ray ranges, persistent grid), but the iteration is laid out so that its long-latency pieces overlap instead of forming one
// dependency chain (section ablation of the previous kernel: 40 % of the time was exposed latency at 2 waves / SIMD):
//   * z of the NEXT ray is prefetched while the current ray is evaluated; with one ray per wave the ray itself is wave-uniform
//     and lives in scalar registers (s_load);
//   * colour taps (<= 2 views) are issued right after the geometry, they land during the MFMA phase;
//   * the gather of G runs two stages ahead in two register buffers and is blended INTO the running accumulators between the
//     positional-encoding octaves: stage s is blended after octave s while the MFMAs of the octave cover the latency of
//     stage s+1 / s+2 and the blend's packed FMAs fill the matrix pipe's issue gaps (acc starts at 0; summation order only);
//   * alpha compositing is a DPP scan (row_shr / row_bcast) instead of ds_bpermute shuffles.
// Replaces: nerf.py:210-313, models_bts.py:138-338, resnetfc.py:132-184, code.py:30-42 of the reference.
#pragma once
#include "bts_field_kernel.h"

// The gather of G goes through LDS (global_load_lds_dwordx4, see GatherLds below) unless the build says -DBTS_GATHER_REGS (the
// round-1 form: two register buffers per lane; kept for A/B as variants/libbts_gatherregs.so)
#if !defined(BTS_GATHER_REGS) && !defined(BTS_GATHER_LDS)
#define BTS_GATHER_LDS
#endif

namespace bts {

#ifdef BTS_PROBE
#define BTS_TICK(i)                                                  \
  if (p.ablate & 128) {                                              \
    const unsigned long long t_now = __builtin_readcyclecounter();   \
    t_acc[i] += t_now - t_last;                                      \
    t_last = t_now;                                                  \
  }
#else
#define BTS_TICK(i)
#endif

// ---- DPP (data-parallel primitives) helpers: lanes whose source is outside its row / whose row is masked keep `old` ----------
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL,
                                                               ROW_MASK, 0xF, false));
}
constexpr int kDppRowShr = 0x110;   // + n
constexpr int kDppWaveShr1 = 0x138;  // lane i <- lane i - 1
constexpr int kDppWaveShl1 = 0x130;  // lane i <- lane i + 1
constexpr int kDppBcast15 = 0x142;  // lane 15 of each row -> every lane of the next row
constexpr int kDppBcast31 = 0x143;  // lane 31 -> rows 2, 3

// inclusive prefix product over each group of lpr consecutive lanes (lpr in {8, 16, 32, 64}); kl = lane & (lpr - 1).
// row_shr leaves `old` (the operation's identity) in the lanes whose source lies outside their 16-lane row, so for lpr >= 16 the shifted
// value needs no per-lane select and the DPP move folds into the multiply / add (one VALU instruction per step); only groups of 8
// lanes, which share a row with a neighbour group, mask the steps.
__device__ __forceinline__ float seg_scan_mul(float x, int lpr, int kl) {
  float y;
  if (lpr == 48) {   // lanes 0-47 one segment (rows 0, 1, 2), lanes 48-63 (row 3) another: see render_kernel_p's 48-lane mode
    x *= dpp_f<kDppRowShr + 1>(1.0f, x);
    x *= dpp_f<kDppRowShr + 2>(1.0f, x);
    x *= dpp_f<kDppRowShr + 4>(1.0f, x);
    x *= dpp_f<kDppRowShr + 8>(1.0f, x);
    x *= dpp_f<kDppBcast15, 0x2>(1.0f, x);   // row 1 <- row 0's total
    x *= dpp_f<kDppBcast31, 0x4>(1.0f, x);   // row 2 <- the scan through row 1
    return x;
  }
  if (lpr >= 16) {
    x *= dpp_f<kDppRowShr + 1>(1.0f, x);
    x *= dpp_f<kDppRowShr + 2>(1.0f, x);
    x *= dpp_f<kDppRowShr + 4>(1.0f, x);
    x *= dpp_f<kDppRowShr + 8>(1.0f, x);
  } else {
    y = dpp_f<kDppRowShr + 1>(1.0f, x), x *= (kl >= 1) ? y : 1.0f;
    y = dpp_f<kDppRowShr + 2>(1.0f, x), x *= (kl >= 2) ? y : 1.0f;
    y = dpp_f<kDppRowShr + 4>(1.0f, x), x *= (kl >= 4) ? y : 1.0f;
  }
  if (lpr >= 32) x *= dpp_f<kDppBcast15, 0xA>(1.0f, x);
  if (lpr >= 64) x *= dpp_f<kDppBcast31, 0xC>(1.0f, x);
  return x;
}
// inclusive prefix sum; the LAST lane of each group holds the group's total
__device__ __forceinline__ float seg_scan_add(float x, int lpr, int kl) {
  float y;
  if (lpr == 48) {
    x += dpp_f<kDppRowShr + 1>(0.0f, x);
    x += dpp_f<kDppRowShr + 2>(0.0f, x);
    x += dpp_f<kDppRowShr + 4>(0.0f, x);
    x += dpp_f<kDppRowShr + 8>(0.0f, x);
    x += dpp_f<kDppBcast15, 0x2>(0.0f, x);
    x += dpp_f<kDppBcast31, 0x4>(0.0f, x);
    return x;
  }
  if (lpr >= 16) {
    x += dpp_f<kDppRowShr + 1>(0.0f, x);
    x += dpp_f<kDppRowShr + 2>(0.0f, x);
    x += dpp_f<kDppRowShr + 4>(0.0f, x);
    x += dpp_f<kDppRowShr + 8>(0.0f, x);
  } else {
    y = dpp_f<kDppRowShr + 1>(0.0f, x), x += (kl >= 1) ? y : 0.0f;
    y = dpp_f<kDppRowShr + 2>(0.0f, x), x += (kl >= 2) ? y : 0.0f;
    y = dpp_f<kDppRowShr + 4>(0.0f, x), x += (kl >= 4) ? y : 0.0f;
  }
  if (lpr >= 32) x += dpp_f<kDppBcast15, 0xA>(0.0f, x);
  if (lpr >= 64) x += dpp_f<kDppBcast31, 0xC>(0.0f, x);
  return x;
}

// gather stage S of the sequence over (point tile, hidden tile, tap pair)
template <int HD, int S>
struct GStage {
  static constexpr int HT = HD / 32;
  static constexpr int pt = S / (2 * HT), ht = (S / 2) % HT, tp2 = S % 2;
};

// 8 float4 of stage S: wave-uniform base (SGPR pair) + 32-bit per-lane byte offset (one sample's G is far below 4 GB), so that the
// address is one v_lshl_add per row instead of 64-bit arithmetic
template <int HD, int S>
__device__ __forceinline__ void stage_load(GBuf& b, const float4* __restrict__ G, const int (&o)[2][4], int h) {
  using St = GStage<HD, S>;
  const char* base = reinterpret_cast<const char*>(G);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const unsigned off = (unsigned)o[St::pt][2 * St::tp2 + t] * (unsigned)(HD * 4) + (unsigned)(St::ht * 128) + (unsigned)h * 64u;
    const float4* row = reinterpret_cast<const float4*>(base + off);
#pragma unroll
    for (int q = 0; q < 4; ++q) b.v[t][q] = row[q];
  }
}
template <int HD, int S>
__device__ __forceinline__ void stage_blend(f32x16 (&acc)[HD / 32][2], const GBuf& b, const float (&w)[2][4]) {
  using St = GStage<HD, S>;
  gblend<false>(acc[St::ht][St::pt], b, w[St::pt][2 * St::tp2], w[St::pt][2 * St::tp2 + 1]);
}

// ---------------------------------------------------------------------------------------------------------------
// Split-precision lin_in on the f16 matrix pipe.
// Measured on gfx950 (tools/ubench/mfma_valu_overlap*.hip): v_mfma_f32_32x32x2_f32 occupies the SIMD for its full 64 cycles -- no
// VALU instruction of the same or of another wave issues underneath it (fp32 "MFMA" runs at, and instead of, the vector FMA rate) --
// whereas v_mfma_f32_32x32x16_f16 costs ~5 cycles when 8+ VALU instructions sit between two of them.  So the 36 sin/cos inputs of
// lin_in (values in [-1, 1]) go through the f16 pipe as  W.X = Wh.Xh + Wl.Xh + Wh.Xl  with  x = xh + xl, xh = f16_rne(x),
// xl = f16_rne(x - xh)  (two round-to-nearest halves carry 12 + 12 significand bits; the dropped Wl.Xl term is 2^-24 relative; fp32
// accumulation in the MFMA).  tools/ubench/f16split_gemm.hip: max error 3.8e-7 / rms 8e-8 against fp64 at K = 48, |D| ~ 2 -- slightly
// better than the fp32-input MFMA (6.1e-7 / 9.3e-8).  Weights are pre-split once per work-group and scaled by 2^S (S chosen from
// max |w| so that the low halves stay normal f16 numbers); everything else that enters the accumulators carries the same 2^S
// (exact), removed again after lin_out.  The raw inputs x, y, code use the spare k rows of the slices (bounded on this path: larger
// arguments take the exact routine); the bias row is the fp32 C operand of each accumulator tile's first MFMA.
// ---------------------------------------------------------------------------------------------------------------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int C, int HD, int NB>
struct LdsH {  // in floats, placed behind Lds<C, HD, NB, true>
  static constexpr int HT = HD / 32;
  static constexpr int W_RAW = 0;                              // [4][HD] k-major rows x, y, code, bias -- times 2^S
  static constexpr int W_F16 = W_RAW + 4 * HD;                 // [term hi/lo][region 3][HT][64 lanes][8 halves = 4 dwords]
  static constexpr int TERM_STRIDE = 3 * HT * 64 * 4;
  static constexpr int BIAS = W_F16 + 2 * TERM_STRIDE;         // per block: b0 [HD], b1 [HD] -- times 2^S
  static constexpr int EMPTY = BIAS + NB * 2 * HD;             // projected empty feature -- times 2^S
  static constexpr int SCALE = EMPTY + HD;                     // [0] 2^S, [1] 2^-S, [2] max |w| (as int bits)
  // ResnetBlockFC linears on the f16 pipe (d_hidden = 32): per layer [term hi/lo][k-slice 2][64 lanes][8 halves] -- times 2^S.
  // k slot (slice s, lane half h, i) is the hidden row mfma_row(8 s + i, h): exactly the 8 accumulator values a lane holds for it,
  // so the C layout of one layer feeds the next layer's B operand from the lane's own registers.
  static constexpr int W_BLK = SCALE + 4;
  static constexpr int BLK_TERM_STRIDE = 2 * 64 * 4;
  static constexpr int BLK_LAYER_STRIDE = 2 * BLK_TERM_STRIDE;
  static constexpr int TOTAL = W_BLK + NB * 2 * BLK_LAYER_STRIDE;
};

// empty_proj: the projected empty feature [HD] (fp32, unscaled) as stage_weights left it in LDS
template <int C, int HD, int NB>
__device__ __forceinline__ void stage_weights_h(float* lh, const float* empty_proj, const float* __restrict__ mlp) {
  using LH = LdsH<C, HD, NB>;
  constexpr int HT = HD / 32;
  constexpr int D_IN = C + kPeDim;
  const MlpLayout ml{D_IN, HD, NB};
  int* mx = reinterpret_cast<int*>(lh + LH::SCALE + 2);
  if (threadIdx.x == 0) *mx = 0;
  __syncthreads();
  float m = 0.0f;
  for (int i = threadIdx.x; i < 39 * HD; i += blockDim.x) m = fmaxf(m, fabsf(mlp[ml.w_in() + (i % HD) * D_IN + C + i / HD]));  // x, y, code + 36 trig rows
  for (int i = threadIdx.x; i < NB * 2 * HD * HD; i += blockDim.x) {   // fc_0 / fc_1 weights share the scale (same f16 range)
    const int b = i / (2 * HD * HD), j = i % (2 * HD * HD);
    m = fmaxf(m, fabsf(mlp[(j < HD * HD ? ml.blk_w0(b) : ml.blk_w1(b)) + j % (HD * HD)]));
  }
  atomicMax(mx, __float_as_int(m));  // non-negative floats order like their bit patterns
  __syncthreads();
  const float wmax = __int_as_float(*mx);
  int ex = 0;
  if (wmax > 0.0f && wmax < 3.0e38f) frexpf(wmax, &ex);
  const int S = max(-40, min(40, 14 - ex));  // max |w| 2^S in [2^13, 2^14): far below the f16 limit, low halves normal down to |w| ~ 2^-13 max
  const float scale = ldexpf(1.0f, S);
  if (threadIdx.x == 0) lh[LH::SCALE] = scale, lh[LH::SCALE + 1] = ldexpf(1.0f, -S);
  for (int i = threadIdx.x; i < 4 * HD; i += blockDim.x) {
    const int k = i / HD, hid = i % HD;
    lh[LH::W_RAW + i] = (k < 3 ? mlp[ml.w_in() + hid * D_IN + C + k] : mlp[ml.b_in() + hid]) * scale;
  }
  _Float16* wf = reinterpret_cast<_Float16*>(lh + LH::W_F16);
  for (int i = threadIdx.x; i < 3 * HT * 64 * 8; i += blockDim.x) {
    const int e = i & 7, lane = (i >> 3) & 63, ht = (i >> 9) % HT, r = i / (HT * 512);
    const int slot = 8 * (lane >> 5) + e, hid = ht * 32 + (lane & 31);
    float w = 0.0f;
    if (slot < 12) w = mlp[ml.w_in() + hid * D_IN + C + 3 + 6 * (2 * r + slot / 6) + slot % 6] * scale;
    else if (r == 0 && slot < 14) w = mlp[ml.w_in() + hid * D_IN + C + (slot - 12)] * scale;   // spare rows of region 0: raw x, y
    else if (r == 1 && slot == 12) w = mlp[ml.w_in() + hid * D_IN + C + 2] * scale;             // spare row of region 1: raw depth code
    asm("" : "+v"(w));
    const _Float16 hi = (_Float16)w;
    wf[i] = hi;
    wf[i + LH::TERM_STRIDE * 2] = (_Float16)(w - (float)hi);  // TERM_STRIDE floats = 2 x halves
  }
  if constexpr (NB > 0) {
    static_assert(HD == 32, "f16 ResnetBlockFC layers are laid out for d_hidden = 32");
    _Float16* wb = reinterpret_cast<_Float16*>(lh + LH::W_BLK);
    for (int i = threadIdx.x; i < NB * 2 * 2 * 64 * 8; i += blockDim.x) {
      const int e = i & 7, lane = (i >> 3) & 63, sl = (i >> 9) & 1, layer = i >> 10;   // layer = 2 * block + {0: fc_0, 1: fc_1}
      const int kin = mfma_row(8 * sl + e, lane >> 5), out = lane & 31;
      float w = mlp[((layer & 1) ? ml.blk_w1(layer >> 1) : ml.blk_w0(layer >> 1)) + out * HD + kin] * scale;
      asm("" : "+v"(w));
      const _Float16 hi = (_Float16)w;
      _Float16* dst = wb + layer * LH::BLK_LAYER_STRIDE * 2 + (sl * 64 + lane) * 8 + e;
      dst[0] = hi;
      dst[LH::BLK_TERM_STRIDE * 2] = (_Float16)(w - (float)hi);
    }
  }
  for (int i = threadIdx.x; i < NB * 2 * HD; i += blockDim.x) {
    const int b = i / (2 * HD), j = i % (2 * HD);
    lh[LH::BIAS + i] = (j < HD ? mlp[ml.blk_b0(b) + j] : mlp[ml.blk_b1(b) + j - HD]) * scale;
  }
  for (int i = threadIdx.x; i < HD; i += blockDim.x) lh[LH::EMPTY + i] = empty_proj[i] * scale;
}

__device__ __forceinline__ void swap32u(unsigned& a, unsigned& b) {
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0], b = r[1];
}
__device__ __forceinline__ unsigned pack_h2(_Float16 a, _Float16 b) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, (h2){a, b});
}
// The f16 split of TWO fp32 values, packed: hi = {f16_rne(e0), f16_rne(e1)}, lo = {f16_rne(e0 - hi0), f16_rne(e1 - hi1)} in THREE
// instructions -- v_cvt_pk_f16_f32, v_fma_mixlo_f16, v_fma_mixhi_f16 (the mixed-precision fma reads the f16 half in place and rounds the
// EXACT difference once: e - hi has at most 13 significant bits, so this is the very value f16(e - float(hi)) of the scalar form).
// Rounds 2 - 5 wrote hi = (_Float16)e; lo = (_Float16)(e - (float)hi) per value, which hipcc compiles to EIGHT instructions per pair (each
// hi converted twice, once alone for the way back to fp32 and once packed; two conversions back; two subtractions; the packed lo).
// `neg1` = -1.0f as an OPAQUE scalar (opaque_neg1()): with the literal the compiler rewrites fma(x, -1, e) as e - x and the mixed form is gone.
// e0, e1 must be opaque fp32 VALUES (see f16_region: a producer folded into one of the conversions is rounded differently there).
__device__ __forceinline__ float opaque_neg1() {
  float k = -1.0f;
  asm("" : "+s"(k));
  return k;
}
__device__ __forceinline__ void split2_f16(float e0, float e1, float neg1, unsigned& hi, unsigned& lo) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const h2 hp = __builtin_convertvector((f2){e0, e1}, h2);
  h2 lp;
  lp[0] = (_Float16)__builtin_fmaf((float)hp[0], neg1, e0);
  lp[1] = (_Float16)__builtin_fmaf((float)hp[1], neg1, e1);
  hi = __builtin_bit_cast(unsigned, hp), lo = __builtin_bit_cast(unsigned, lp);
}

// 12 encoding entries (two octaves) of this lane's sample -> both point tiles' B operands (high and low halves) -> 12 f16 MFMAs
template <int HD, int NE = 12, bool FIRST = false>
__device__ __forceinline__ void f16_region(f32x16 (&acc)[HD / 32][2], const float* wf /* lane-resolved, this region */, int term_stride,
                                           const float (&e)[NE], const f32x16* bias = nullptr) {
  static_assert(NE >= 8 && NE <= 16, "one 16-row k-slice");
  constexpr int HT = HD / 32;
  // The entries as opaque fp32 VALUES.  Otherwise hipcc folds the producing fma / multiply into ONE of the two conversions of the split
  // (v_fma_mixlo_f16: a single rounding of the exact result) while the other goes through v_cvt_pk_f16_f32 of the rounded fp32
  // value; in the 2^-13 of cases where double rounding matters the MFMA operand hi and the hi inside lo then differ by one f16
  // ulp and hi + lo misses e by 5e-4 |e| (tools/ubench/f16_split_fusion.hip; seen as 1e-4 jumps of the MLP output).
  float ev[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    ev[i] = i < NE ? e[i] : 0.0f;
    if (i < NE) asm("" : "+v"(ev[i]));
  }
  const float neg1 = opaque_neg1();
  unsigned ph[4], qh[4], pl[4], ql[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    split2_f16(ev[2 * j], ev[2 * j + 1], neg1, ph[j], pl[j]);
    if (8 + 2 * j < NE) split2_f16(ev[8 + 2 * j], ev[9 + 2 * j], neg1, qh[j], ql[j]);   // (an odd NE: entry NE is the constant 0, hi = lo = 0)
    else qh[j] = 0u, ql[j] = 0u;
    swap32u(ph[j], qh[j]);  // p*: point tile 0 (k 0-7 from its own lanes, k 8-15 from the partner half), q*: point tile 1
    swap32u(pl[j], ql[j]);
  }
  const h8 b0h = __builtin_bit_cast(h8, (u32x4){ph[0], ph[1], ph[2], ph[3]});
  const h8 b1h = __builtin_bit_cast(h8, (u32x4){qh[0], qh[1], qh[2], qh[3]});
  const h8 b0l = __builtin_bit_cast(h8, (u32x4){pl[0], pl[1], pl[2], pl[3]});
  const h8 b1l = __builtin_bit_cast(h8, (u32x4){ql[0], ql[1], ql[2], ql[3]});
  h8 ah[HT], al[HT];
#pragma unroll
  for (int ht = 0; ht < HT; ++ht) {
    ah[ht] = *reinterpret_cast<const h8*>(wf + ht * 256);
    al[ht] = *reinterpret_cast<const h8*>(wf + term_stride + ht * 256);
  }
#pragma unroll
  for (int ht = 0; ht < HT; ++ht) {
    // FIRST: the accumulators are born here, from the bias row (both point tiles share it)
    acc[ht][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ht], b0h, FIRST ? bias[ht] : acc[ht][0], 0, 0, 0);
    acc[ht][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ht], b1h, FIRST ? bias[ht] : acc[ht][1], 0, 0, 0);
  }
#pragma unroll
  for (int ht = 0; ht < HT; ++ht) {
    acc[ht][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ht], b0h, acc[ht][0], 0, 0, 0);
    acc[ht][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ht], b1h, acc[ht][1], 0, 0, 0);
  }
#pragma unroll
  for (int ht = 0; ht < HT; ++ht) {
    acc[ht][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ht], b0l, acc[ht][0], 0, 0, 0);
    acc[ht][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ht], b1l, acc[ht][1], 0, 0, 0);
  }
}

// out[pt] += (W 2^S) . relu(in[pt]) 2^-S for one ResnetBlockFC linear of width 32 on the f16 pipe (split precision, as lin_in):
// the lane's own 16 accumulator values of a point tile are the B operand of two 16-row k-slices -- no data movement between layers.
// `in` carries 2^S (like every accumulator of this path); it is unscaled on the way into f16 and the pre-scaled weights put 2^S back.
__device__ __forceinline__ void hidden_layer_h(f32x16 (&out)[1][2], const f32x16 (&in)[1][2], const float* wl /* lane-resolved, this layer */,
                                               int term_stride, float inv_scale) {
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    const h8 ah = *reinterpret_cast<const h8*>(wl + sl * 256);
    const h8 al = *reinterpret_cast<const h8*>(wl + term_stride + sl * 256);
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      float vc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        // relu, then back to the true magnitude; the upper clamp only keeps a (never observed) 6e4 activation from turning into inf
        const float v = __builtin_amdgcn_fmed3f(in[0][pt][8 * sl + i], 0.0f, 3.4028234663852886e38f) * inv_scale;
        vc[i] = fminf(v, 6.0e4f);
        asm("" : "+v"(vc[i]));   // opaque fp32 value: both conversions of the split must see the same rounding (see f16_region)
      }
      const float neg1 = opaque_neg1();
      unsigned uh[4], ul[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) split2_f16(vc[2 * j], vc[2 * j + 1], neg1, uh[j], ul[j]);
      const h8 bh = __builtin_bit_cast(h8, (u32x4){uh[0], uh[1], uh[2], uh[3]});
      const h8 bl = __builtin_bit_cast(h8, (u32x4){ul[0], ul[1], ul[2], ul[3]});
      out[0][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, out[0][pt], 0, 0, 0);
      out[0][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, out[0][pt], 0, 0, 0);
      out[0][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, out[0][pt], 0, 0, 0);
    }
  }
}

// The octave chain of the three encoding regions (region R = octaves 2R, 2R + 1).  Direct evaluations (Cody-Waite + polynomials, ~26
// instructions per argument) at octaves 0 and 3 only; 1, 2 and 4, 5 by angle doubling (fl(v 2f) = 2 fl(v f) exactly; ~1e-7 absolute error
// per doubling, two in a row at most).  -DBTS_PE_DIRECT3 restores rounds 1 - 2: direct at 0, 2, 4, one doubling each (+69 instructions
// per ray).  `raw` = octave 2R on entry and octave 2R + 2 on exit; `pre` carries octave 3 from region 0 (where it is evaluated, a region
// ahead of its use like every direct evaluation) to region 1.
template <int R>
__device__ __forceinline__ void pe_second_octave(SinCos3& r1, const SinCos3& raw, const SinCos3& pre) {
#ifdef BTS_PE_DIRECT3
  pe_double(r1, raw);
#else
  if constexpr (R == 1) r1 = pre;
  else pe_double(r1, raw);
#endif
}
template <int R>
__device__ __forceinline__ void pe_advance(SinCos3& raw, SinCos3& pre, const SinCos3& r1, const float (&v3)[3], float ff) {
  if constexpr (R + 1 < 3) {
#ifdef BTS_PE_DIRECT3
    pe_direct(raw, v3, ff * 4.0f);
#else
    pe_double(raw, r1);                                   // octave 2R + 2 from octave 2R + 1
    if constexpr (R == 0) pe_direct(pre, v3, ff * 8.0f);  // octave 3
#endif
  }
}

// Region R = octaves 2R (sines computed directly, one region ahead) and 2R+1 (by angle doubling), with gather stages 2R and 2R+1
// blended behind the region's MFMAs and stages 2R+2, 2R+3 issued.  The regions are no longer fenced from each other: the scheduler
// may pull the next region's trigonometry under this region's MFMAs (3 % faster, +3 spilled VGPRs).
template <int HD, int R>
__device__ __forceinline__ void region_seq_i(f32x16 (&acc)[HD / 32][2], GBuf& ba, GBuf& bb, const float4* __restrict__ G,
                                             const int (&o)[2][4], const float (&wq)[2][4], int h, const float* wf, int term_stride,
                                             SinCos3& raw, SinCos3& pre, const float (&v3)[3], float ff, const f32x16* bias, bool nosin = false,
                                             bool nomfma = false) {
  constexpr int HT = HD / 32;
  constexpr int NS = 4 * HT;
  if constexpr (R < 3) {
    constexpr int NE = R == 0 ? 14 : (R == 1 ? 13 : 12);   // + raw x, y (region 0) / raw depth code (region 1) in the spare k rows
    float e[NE];
    if (nosin) {   // probe builds only: no trigonometry
#pragma unroll
      for (int i = 0; i < 12; ++i) e[i] = v3[i % 3] * ff;
    } else {
      float t[6];
      pe_entries(t, raw, v3, ff);
#pragma unroll
      for (int i = 0; i < 6; ++i) e[i] = t[i];
      SinCos3 r1;
      pe_second_octave<R>(r1, raw, pre);
      pe_entries(t, r1, v3, ff * 2.0f);
#pragma unroll
      for (int i = 0; i < 6; ++i) e[6 + i] = t[i];
      pe_advance<R>(raw, pre, r1, v3, ff);
    }
    if constexpr (R == 0) e[12] = v3[0], e[13] = v3[1];
    if constexpr (R == 1) e[12] = v3[2];
    if (nomfma) {  // probe builds only: no split, no MFMAs
#pragma unroll
      for (int i = 0; i < NE; ++i) acc[0][0][i] += e[i];
      if constexpr (R == 0) {
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) acc[ht][0] += bias[ht], acc[ht][1] = bias[ht];
      }
    } else {
      f16_region<HD, NE, R == 0>(acc, wf + R * HT * 256, term_stride, e, bias);
    }
    if constexpr (2 * R < NS) {
      stage_blend<HD, 2 * R>(acc, ba, wq);
      if constexpr (2 * R + 2 < NS) stage_load<HD, 2 * R + 2>(ba, G, o, h);
      stage_blend<HD, 2 * R + 1>(acc, bb, wq);
      if constexpr (2 * R + 3 < NS) stage_load<HD, 2 * R + 3>(bb, G, o, h);
    }
#ifdef BTS_REGION_BARRIER   // one scheduling region per encoding region: was needed against spills in r01c, costs 3 % now (r01g A/B)
    __builtin_amdgcn_sched_barrier(0);
#endif
    region_seq_i<HD, R + 1>(acc, ba, bb, G, o, wq, h, wf, term_stride, raw, pre, v3, ff * 4.0f, bias, nosin, nomfma);
  }
}
template <int HD, int R>
__device__ __forceinline__ void region_seq(f32x16 (&acc)[HD / 32][2], GBuf& ba, GBuf& bb, const float4* __restrict__ G,
                                           const int (&o)[2][4], const float (&wq)[2][4], int h, const float* wf, int term_stride,
                                           SinCos3& raw, const float (&v3)[3], float ff, const f32x16* bias, bool nosin = false,
                                           bool nomfma = false) {
  static_assert(R == 0, "entered at region 0");
  SinCos3 pre;
  region_seq_i<HD, 0>(acc, ba, bb, G, o, wq, h, wf, term_stride, raw, pre, v3, ff, bias, nosin, nomfma);
}

#ifdef BTS_GATHER_LDS
constexpr int kGatherLdsPerWave = 3 * 4096 + 768;
// ---------------------------------------------------------------------------------------------------------------
// The gather of G through LDS.
//
// PMC and the per-view split of the eval frame say the forward is bound by the texture addresser for every ray that does NOT pass
// through the encoder camera (profiles/README.md, r02f): there each sample of a ray hits its own texels, lane (h, col) reads its 64
// bytes as four 16-byte loads, and every one of the 64 gather instructions of a ray presents 64 different cache lines to the L1
// tag pipeline (one per clock): 4 096 clocks per ray and CU, exactly the measured 6.4 ns per ray against 4.2 ns for encoder-camera
// rays (whose samples share their texels: one line per instruction).
// Here EIGHT ADJACENT LANES fetch one whole 128-byte row half (global_load_lds_dwordx4: 16 bytes per lane straight into LDS, no
// VGPRs): 8 lines per instruction instead of 64.  The hardware puts lane L's 16 bytes at M0 + 16 L, so the row of point 8j + m
// (instruction j of a block, m = L >> 3) lands at block + j * 1024 + m * 128; the pieces of row r = 8 j + m are fetched rotated by
// (r >> 1) & 7 = (4 j + (m >> 1)) & 7 so that the consumer -- lane (h, col) reading the four pieces 4h .. 4h + 3 of row col as
// ds_read_b128 -- spreads over all banks.  gfx950 serves a ds_read_b128 in four 16-lane groups, {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}
// and the same + 32 (MI355X_MICROARCH.md, LDS), 256 bytes = 16 slots of 16 bytes per cycle: a group's eight even rows need eight different
// rotations, and so do its eight odd rows (slot = 8 (r & 1) + ((piece - rot) & 7)).  (r >> 1) & 7 is a bijection on each of those sets;
// rounds 2 - 5 rotated by (r & 7) >> 1, which pairs rows 0-3 with 24-27 and 12-15 with 20-23: every read took two cycles per group
// -- SQ_LDS_BANK_CONFLICT 62.9 M cycles per eval frame = 64 reads x 4 extra cycles x 245 760 rays, profiles/r05m/traffic_fwd_def.json.)
// A block = one tap of one (point tile, hidden tile) = 32 rows = 4 KB; a ring of three blocks per wave; block T + 3 is issued into
// the slot of block T once T has been blended.  The tap offsets of the 64 samples go through a 768-byte per-wave table (lane = sample
// writes, lane (m, piece) reads the row of sample 32 pt + 8 j + m).  52 KB of dynamic LDS per work-group on top of the weights: two
// work-groups per CU still fit for every shape (the RE10K model's 29 KB of weights included).
struct GatherLds {
  const char* ring;     // this wave's ring (3 blocks of 4 KB), generic pointer
  unsigned ring_m0;     // ... as an LDS byte address (M0 of the DMA loads)
  unsigned* tab;        // this wave's tap table: [64 samples][3] byte offsets into G of the taps nw, ne, sw
  unsigned rd[4];       // byte offset inside a block of this lane's piece q as the CONSUMER (h, col)
  unsigned piece16;     // 16 * the piece this lane FETCHES in the EVEN instructions of a block: ((L & 7) + (L >> 4)) & 7, i.e. rotated by m >> 1, m = L >> 3
  unsigned piece16x;    // ... in the ODD instructions (rows 8 .. 15, 24 .. 31): rotated by 4 more = piece16 ^ 64
  int m;                // L >> 3
};
template <int HD, int T>
struct GBlock {   // block T of a ray, in the order the stages are blended (GStage)
  static constexpr int HT = HD / 32;
  static constexpr int S = T / 2, pt = S / (2 * HT), ht = (S / 2) % HT, tap = 2 * (S % 2) + T % 2, slot = T % 3;
  static constexpr int NBLK = 8 * HT;
};
// The four per-lane byte offsets (instructions j = 0..3) of block T.  The instruction offset of an LDS-DMA load is added to the global
// AND to the LDS address; j's 1 KB step inside the block therefore rides in the INSTRUCTION (one M0 per block instead of one per load:
// s_mov m0 + its wait state + the scalar add were 192 of the ~2 600 instructions a wave issues per ray) and is taken back out of the
// global side here: offsets are relative to G - kGlBias (gl_issue passes that base), + kGlBias - 1024 j >= 0.
constexpr unsigned kGlBias = 3 * 1024;
template <int HD, int T>
__device__ __forceinline__ void gl_offsets(const GatherLds& c, unsigned (&off)[4]) {
  using B = GBlock<HD, T>;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned* row = c.tab + (32 * B::pt + 8 * j + c.m) * 3;   // [o00, o01, o10]; o11 = o10 + (o01 - o00) (clamped taps included)
    off[j] = (B::tap < 3 ? row[B::tap] : row[2] + row[1] - row[0]) + ((j & 1 ? c.piece16x : c.piece16) + kGlBias - 1024u * j);
  }
}
// what the overwrite of a slot must wait for: one value computed from EACH of the four row pieces (ds_read_b128) of the block that
// lived there.  See gl_issue.
struct GDep {
  float d[4];
};
template <int HD, int T>
__device__ __forceinline__ void gl_issue(const GatherLds& c, const float4* G, const unsigned (&off)[4], const GDep& dep) {
  using B = GBlock<HD, T>;
  // `dep`: the DMA write into LDS does not queue behind this wave's ds_reads -- a ds_read_b128 of the slot's previous block that is
  // still waiting in the LDS queue when the new rows arrive returns the NEW rows (seen on the RE10K shapes, where eight waves'
  // weight reads keep the queue hundreds of cycles deep: a handful of rays per frame differed from run to run, r03i / r03j).  The
  // memory clobber orders only the ISSUE of those reads, so the issue takes one input computed from each of the four pieces: the
  // compiler has to wait for all four reads to RETURN before the first load of the block goes out.  (One value of the last piece
  // is not enough: the scheduler reorders the four reads and sinks the blend of the others below the loads.)
  // M0 = the slot's LDS address minus the hidden tile's share of the instruction offset (ht * 128: that part belongs to the global
  // address only); the loads' instruction offsets ht * 128 + j * 1024 then place row group j at slot + j * 1024.
  const unsigned m0v = c.ring_m0 + B::slot * 4096 - B::ht * 128;
  const char* Gb = reinterpret_cast<const char*>(G) - kGlBias;
  asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %0, %5 offset:%6\n\t"
               "global_load_lds_dwordx4 %1, %5 offset:%7\n\t"
               "global_load_lds_dwordx4 %2, %5 offset:%8\n\t"
               "global_load_lds_dwordx4 %3, %5 offset:%9"
               :: "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(m0v), "s"(Gb), "n"(B::ht * 128), "n"(B::ht * 128 + 1024),
                  "n"(B::ht * 128 + 2048), "n"(B::ht * 128 + 3072), "v"(dep.d[0]), "v"(dep.d[1]), "v"(dep.d[2]), "v"(dep.d[3])
               : "memory", "m0");
}
struct GRows {
  float4 v[2][4];   // the lane's four 16-byte pieces of two blocks (even / odd T)
};
// wait until block T has landed, read this lane's pieces.  ISSUED = blocks issued after T at this point of the schedule
template <int HD, int T, int ISSUED>
__device__ __forceinline__ void gl_fetch(const GatherLds& c, GRows& r) {
  using B = GBlock<HD, T>;
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * ISSUED) : "memory");
#pragma unroll
  for (int q = 0; q < 4; ++q) r.v[T & 1][q] = *reinterpret_cast<const float4*>(c.ring + B::slot * 4096 + c.rd[q]);
}
// one step: block T (its rows requested from LDS one step ago) is blended (acc += w_tap * row, the order of gblend), block T + 3 goes
// out into T's slot, the offsets of block T + 4 are fetched from the table, and the rows of block T + 1 are requested from LDS -- last,
// when block T + 1 has had the whole step to land (two blocks may still be in flight behind it).  -DBTS_GL_FETCH_EARLY requests them
// first instead (rounds 1 - 2 shipped that order; 3 % slower on the eval frame, profiles/r03_experiments/r03k).
template <int HD, int T>
__device__ __forceinline__ void gl_consume(f32x16 (&acc)[HD / 32][2], const GatherLds& c, GRows& r, const float4* G, const float (&w)[2][4],
                                           unsigned (&off_next)[4]) {
  using B = GBlock<HD, T>;
#ifdef BTS_GL_FETCH_EARLY
  if constexpr (T + 1 < B::NBLK) gl_fetch<HD, T + 1, (T + 2 < B::NBLK ? 1 : 0)>(c, r);   // issued so far: blocks 0 .. T + 2
#endif
  const f32x2 wv = {w[B::pt][B::tap], w[B::pt][B::tap]};
  f32x16& a = acc[B::ht][B::pt];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
      const float* x = reinterpret_cast<const float*>(&r.v[T & 1][q]) + e;
      const f32x2 xv = {x[0], x[1]};
      const f32x2 res = __builtin_elementwise_fma(xv, wv, (f32x2){a[4 * q + e], a[4 * q + e + 1]});
      a[4 * q + e] = res[0], a[4 * q + e + 1] = res[1];
    }
  }
  if constexpr (T + 3 < B::NBLK) {
    gl_issue<HD, T + 3>(c, G, off_next, GDep{{a[3], a[7], a[11], a[15]}});
    if constexpr (T + 4 < B::NBLK) gl_offsets<HD, T + 4>(c, off_next);
  }
#ifndef BTS_GL_FETCH_EARLY
  if constexpr (T + 1 < B::NBLK) gl_fetch<HD, T + 1, (T + 3 < B::NBLK ? 2 : (T + 2 < B::NBLK ? 1 : 0))>(c, r);   // issued: blocks 0 .. T + 3
#endif
}
// start of a ray: table written and fenced by the caller; blocks 0, 1, 2 go out, block 0 is requested from LDS, the offsets of block 3
// wait in off_next
template <int HD>
__device__ __forceinline__ void gl_prologue(const GatherLds& c, GRows& r, const float4* G, unsigned (&off_next)[4]) {
  unsigned o0[4], o1[4], o2[4];
  gl_offsets<HD, 0>(c, o0), gl_offsets<HD, 1>(c, o1), gl_offsets<HD, 2>(c, o2);
  // nothing of the previous ray may still be queued in LDS when its slots are overwritten (gl_issue): the table reads above need the
  // wait anyway, LDS returns in order
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const GDep none = {{0.0f, 0.0f, 0.0f, 0.0f}};
  gl_issue<HD, 0>(c, G, o0, none), gl_issue<HD, 1>(c, G, o1, none), gl_issue<HD, 2>(c, G, o2, none);
  gl_offsets<HD, 3>(c, off_next);
}
// region_seq with the gather through LDS: region R blends the blocks of stages 2R and 2R + 1 behind its MFMAs
template <int HD, int R>
__device__ __forceinline__ void region_seq_li(f32x16 (&acc)[HD / 32][2], const GatherLds& gl, GRows& rows, const float4* __restrict__ G,
                                              const float (&wq)[2][4], unsigned (&off_next)[4], const float* wf, int term_stride, SinCos3& raw,
                                              SinCos3& pre, const float (&v3)[3], float ff, const f32x16* bias) {
  constexpr int HT = HD / 32;
  constexpr int NS = 4 * HT;
  if constexpr (R < 3) {
    constexpr int NE = R == 0 ? 14 : (R == 1 ? 13 : 12);
    float e[NE];
    float t[6];
    pe_entries(t, raw, v3, ff);
#pragma unroll
    for (int i = 0; i < 6; ++i) e[i] = t[i];
    SinCos3 r1;
    pe_second_octave<R>(r1, raw, pre);
    pe_entries(t, r1, v3, ff * 2.0f);
#pragma unroll
    for (int i = 0; i < 6; ++i) e[6 + i] = t[i];
    if constexpr (R == 0) e[12] = v3[0], e[13] = v3[1];
    if constexpr (R == 1) e[12] = v3[2];
    pe_advance<R>(raw, pre, r1, v3, ff);
    f16_region<HD, NE, R == 0>(acc, wf + R * HT * 256, term_stride, e, bias);
    if constexpr (2 * R < NS) {
      if constexpr (R == 0) gl_fetch<HD, 0, 2>(gl, rows);   // blocks 1, 2 were issued after block 0
      gl_consume<HD, 4 * R + 0>(acc, gl, rows, G, wq, off_next);
      gl_consume<HD, 4 * R + 1>(acc, gl, rows, G, wq, off_next);
      gl_consume<HD, 4 * R + 2>(acc, gl, rows, G, wq, off_next);
      gl_consume<HD, 4 * R + 3>(acc, gl, rows, G, wq, off_next);
    }
    region_seq_li<HD, R + 1>(acc, gl, rows, G, wq, off_next, wf, term_stride, raw, pre, v3, ff * 4.0f, bias);
  }
}
template <int HD, int R>
__device__ __forceinline__ void region_seq_l(f32x16 (&acc)[HD / 32][2], const GatherLds& gl, GRows& rows, const float4* __restrict__ G,
                                             const float (&wq)[2][4], unsigned (&off_next)[4], const float* wf, int term_stride, SinCos3& raw,
                                             const float (&v3)[3], float ff, const f32x16* bias) {
  static_assert(R == 0, "entered at region 0");
  SinCos3 pre;
  region_seq_li<HD, 0>(acc, gl, rows, G, wq, off_next, wf, term_stride, raw, pre, v3, ff, bias);
}
#endif  // BTS_GATHER_LDS


// Cold path: a wave in which some sample's encoding argument leaves the fast sincos range (|arg| > 1e5: points within millimetres
// of the encoder's camera plane) evaluates that iteration with the compact lane = point routine (libm range reduction inside).
// Kept out of line -- inlining 36 libm sines next to the pipelined path costs ~240 spilled VGPRs on the HOT path.
template <int C, int HD, int NB>
__device__ __attribute__((noinline)) float eval_point_exact(const float* lds, const float4* G, const float* w2c, const float* Kc, int H, int W,
                                                            int fs, int code_mode, int inv_z, float inv_dmax, float inv_range, float d_min, float range,
                                                            float freq_factor, int learn_empty, float b_out, float px, float py, float pz) {
  FwdParams q;
  q.H = H, q.W = W, q.fs = fs, q.code_mode = code_mode, q.inv_z = inv_z, q.inv_dmax = inv_dmax, q.inv_range = inv_range, q.d_min = d_min;
  q.range = range, q.freq_factor = freq_factor, q.learn_empty = learn_empty, q.ablate = 0;
  const Cam enc = load_cam(w2c, Kc);
  Proj pe;
  return eval_point<C, HD, NB, true>(q, lds, enc, G, (int)(threadIdx.x & 63), b_out, px, py, pz, pe);
}

// What the top of an iteration of the persistent kernels needs of the kernel parameters (cameras, ray, sample positions, projection,
// depth code).  With PIN they are fetched from the kernarg segment in ONE batch (the empty asm takes every member as an
// operand, so the compiler has to issue all the scalar loads in front of it and waits once) instead of in ten dependent round trips;
// measured, that loses (below), so the product reads them where they are used like every other parameter.
template <bool PIN>
struct IterHeadT {
  const float* w2c_enc;
  const float* K_enc;
  const float* proj;
  const float* rays;
  const float* z_samp;
  const float* jitter;
  int K, H, W, nv, fs, code_mode, inv_z, learn_empty;
  float inv_dmax, inv_range, d_min, range, freq_factor;
  template <typename Q>
  __device__ __forceinline__ explicit IterHeadT(Q q) {
    const FwdParams __attribute__((address_space(4)))* f = head_of(q);
    w2c_enc = f->w2c_enc, K_enc = f->K_enc, proj = f->proj, rays = f->rays, z_samp = f->z_samp, jitter = f->jitter;
    K = f->K, H = f->H, W = f->W, nv = f->nv, fs = f->fs, code_mode = f->code_mode, inv_z = f->inv_z, learn_empty = f->learn_empty;
    inv_dmax = f->inv_dmax, inv_range = f->inv_range, d_min = f->d_min, range = f->range, freq_factor = f->freq_factor;
    // A/B (profiles/r03_experiments/r03q): with the pin the FORWARD is 1.5 % slower -- the second wave of the SIMD hides the scalar round trips, and
    // the 22 more spilled SGPRs are VALU instructions in a VALU-bound loop.  Without it this struct is only a list of names: the
    // compiler sinks every load to its use.  The backward's pass A is the other way round (it waits, it does not issue): PIN = true.
    if constexpr (PIN)
    asm volatile("" : "+s"(w2c_enc), "+s"(K_enc), "+s"(proj), "+s"(rays), "+s"(z_samp), "+s"(K), "+s"(H), "+s"(W), "+s"(nv), "+s"(fs),
                 "+s"(code_mode), "+s"(inv_z), "+s"(learn_empty), "+s"(inv_dmax), "+s"(inv_range), "+s"(d_min), "+s"(range), "+s"(freq_factor));
  }
  // the FwdParams at the head of the parameter block: the block itself (forward) or its first member (backward / query blocks)
  static __device__ __forceinline__ const FwdParams __attribute__((address_space(4)))* head_of(const FwdParams __attribute__((address_space(4)))* q) { return q; }
  template <typename B>
  static __device__ __forceinline__ const FwdParams __attribute__((address_space(4)))* head_of(const B __attribute__((address_space(4)))* q) { return &q->f; }
};
using IterHead = IterHeadT<false>;

// EPI: also reduce weights * invalid and max invalid over each ray's samples (BtsRenderArgs.invalid_wsum / invalid_any).  A template
// parameter, not a run-time test: the evaluation instantiations carry no trace of it (16 more spilled SGPRs otherwise).
template <int C, int HD, int NB, int NVMAX, bool ONE_RAY, bool F16, bool EPI = false>
__global__ __launch_bounds__(256, BTS_FWD_WAVES) void render_kernel_p(const FwdParams p) {
  static_assert(F16, "lin_in runs on the f16 matrix pipe in split precision: the fp32-input-MFMA form of rounds 1 - 2 is gone (git history)");
  using L = Lds<C, HD, NB, true>;
  using LH = LdsH<C, HD, NB>;
  constexpr int HT = HD / 32;
  constexpr int NS = 4 * HT;
  __shared__ __attribute__((aligned(16))) float lds[L::TOTAL + (F16 ? LH::TOTAL + 4 : 0)];
  float* const lh = lds + ((L::TOTAL + 3) & ~3);  // 16-byte aligned: the f16 A operands are read as ds_read_b128
  stage_weights<C, HD, NB, true>(lds, p.mlp, p.empty_feature);
  __syncthreads();
  if constexpr (F16) {
    stage_weights_h<C, HD, NB>(lh, lds + L::EMPTY, p.mlp);
    __syncthreads();
  }
  // 2^S carried by the accumulators of the f16 path (1 on the fp32 path)
  const float scale = F16 ? __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lh[LH::SCALE]))) : 1.0f;
  const float inv_scale = F16 ? __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lh[LH::SCALE + 1]))) : 1.0f;

  const int lane = threadIdx.x & 63;
  const int h0 = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef BTS_GATHER_LDS
  extern __shared__ __attribute__((aligned(128))) char gather_lds[];   // per wave: ring of 3 x 4 KB + 768 B tap table
  GatherLds gl;
  {
    char* base = gather_lds + wave * kGatherLdsPerWave;
    gl.ring = base;
    gl.ring_m0 = (unsigned)(unsigned long)base;     // low 32 bits of a generic LDS address = the LDS byte address
    gl.tab = reinterpret_cast<unsigned*>(base + 3 * 4096);
    gl.m = lane >> 3;
    gl.piece16 = 16u * (unsigned)(((lane & 7) + (lane >> 4)) & 7);
    gl.piece16x = gl.piece16 ^ 64u;
    const int col = lane & 31;
#pragma unroll
    for (int q = 0; q < 4; ++q) gl.rd[q] = (unsigned)(col * 128 + ((4 * h0 + q - (col >> 1)) & 7) * 16);
  }
#endif
  const int nwg = gridDim.x;  // multiple of 8
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int wg_per_xcd = nwg >> 3;
  const int xcd = wg / wg_per_xcd;
  const int lw = (wg - xcd * wg_per_xcd) * 4 + wave;  // wave index inside its XCD
  const int waves_per_xcd = wg_per_xcd * 4;
  // 48-lane mode (render_geometry: 32 < K <= 48 and the rays of a batch element a multiple of four -- exp_re10k.yaml's n_coarse = 48):
  // a wave takes FOUR rays in THREE iterations -- lanes 0-47 one whole ray per iteration, lanes 48-63 one 16-sample row of the
  // fourth -- instead of idling a quarter of its lanes on every instruction.  The fourth ray's transmittance and partial sums cross
  // the three iterations (the chunk loop below, reused); everything per sample is lane-local anyway.
  const int lpr = ONE_RAY ? 64 : p.lpr;
  const bool pk48 = !ONE_RAY && lpr == 48;
  const bool mainl = lane < 48;
  const int R = pk48 ? 4 : 64 / lpr;   // rays per group
  const int kl = pk48 ? (mainl ? lane : lane - 48) : lane & (lpr - 1);
  auto lane_ray = [&](int gg, int it) -> long { return pk48 ? (long)gg * 4 + (mainl ? it : 3) : (long)gg * R + lane / lpr; };
  auto lane_k = [&](int it) -> int { return pk48 && !mainl ? 16 * it + kl : kl; };
  // Work distribution.  Ray groups are cut into chunks of 2^chunk_log2 consecutive groups; chunk c belongs to XCD c % 8, and the
  // waves of an XCD walk its chunks in order (consecutive waves = consecutive rays, so the texel footprints of the waves resident
  // on an XCD still overlap in its L2).  Round 1 gave every XCD ONE contiguous eighth of the rays: with the two stereo views of
  // eval_depth in one launch, XCDs 0-3 got the encoder camera's own rays (0.56 ms for all of them alone) and XCDs 4-7 the partner's
  // (0.86 ms alone) -- the launch took 1.69 ms, the slower half's 2 x 0.86, instead of 1.42.
  const int CHL = p.chunk_log2;
  // (group indices fit 32 bits -- render_fwd_impl refuses more than 2^31 - 2^20 groups: 64-bit scalar arithmetic here was ~120 SALU
  // instructions per iteration)
  const int n_groups = (int)p.groups;
  const int n_chunks = (n_groups + (1 << CHL) - 1) >> CHL;
  auto group_of = [&](int idx) -> int {   // idx-th group of this XCD's chunk list, or -1 past the end
    const int c = ((idx >> CHL) << 3) + xcd;
    const int gg = (c << CHL) + (idx & ((1 << CHL) - 1));
    return (c < n_chunks && gg < n_groups) ? gg : -1;
  };
  const int Bp = p.Bp;
  const float b_out = as_const(p.mlp)[MlpLayout{C + kPeDim, HD, NB}.b_out()];
  const int lane_off0 = h0 * HD + (lane & 31);
  const bool nomfma = BTS_ABL(4), nosin = BTS_ABL(2);

#ifdef BTS_PROBE
  unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last = __builtin_readcyclecounter();
  const unsigned long long t_begin = t_last;
#endif
  const int groups_per_sample = Bp / R;   // Bp % R == 0 (render_geometry)
  int sample_end = groups_per_sample;
  int sample = 0;
  int idx = lw;
  int g = group_of(idx);
  // z of the first ray group -- or, without z_samp, its jitter u: sample_coarse then runs at the top of the iteration (coarse_depth)
  float z_pre = 0.0f, zn_pre = 0.0f;
  if (g >= 0) {
    const int K = p.K;
    const long row = lane_ray(g, 0) * K;
    const int kk = min(lane_k(0), K - 1);
    if (p.z_samp) z_pre = p.z_samp[row + kk], zn_pre = p.z_samp[row + min(kk + 1, K - 1)];
    else z_pre = p.jitter[row + kk];
  }
  // one ray per iteration: the ray (origin, direction, near, far: 32 bytes, wave-uniform) of the NEXT iteration is fetched while this one
  // is evaluated -- as a VECTOR load (all lanes the same address: one request), because a scalar load in flight turns every LDS wait
  // behind it into lgkmcnt(0) (scalar loads return out of order), i.e. the round trip would stall the first weight read instead of the
  // loop head.  Lane i (mod 8) holds float i of the record: one register, one 32-byte request; the loop head moves eight lanes into SGPRs.
  float nrec = 0.0f;
  if constexpr (ONE_RAY) {
    if (g >= 0) nrec = p.rays[(long)g * 8 + (lane & 7)];
  }
  const float step0 = 1.0f / (float)p.K;                       // nerf.py:107
  const float base0 = coarse_base(p.K, min(kl, p.K - 1));      // linspace(0, 1 - step, K)[k] of this lane's sample (first chunk)

  for (; g >= 0; idx += waves_per_xcd, g = group_of(idx)) {
    // the parameters of this iteration: re-read from the kernarg segment where they are used instead of held (and spilled) for the
    // whole life of the kernel (bts_common.h: kernarg_view)
    auto q = kernarg_view<FwdParams>();
    asm volatile("" : "+s"(q));
    IterHead ih(q);   // the geometry's share of the parameters (see IterHead: batching their loads was measured and not kept)
    const int K = ih.K, H = ih.H, W = ih.W, nv = ih.nv, fs = ih.fs;
    long ray = lane_ray(g, 0);   // (48-lane mode: lanes 0-47 move on to the group's next ray with every chunk of the loop below)
    // all rays of a group belong to one batch element; g only grows along a wave's chunk list, so the element is tracked by a
    // running boundary (the 64-bit division this replaces was ~140 dependent scalar instructions at the top of every iteration)
    while (g >= sample_end) ++sample, sample_end += groups_per_sample;
    const Cam enc = load_cam(ih.w2c_enc + sample * 16, ih.K_enc + sample * 9);
    const float4* __restrict__ G = reinterpret_cast<const float4*>(ih.proj) + (long)sample * (H >> fs) * (W >> fs) * (HD / 4);
    float ox, oy, oz, dx, dy, dz, near = 0.0f, far = 0.0f;
    const bool from_jitter = ih.z_samp == nullptr;   // wave-uniform
    if constexpr (ONE_RAY) {  // wave-uniform ray, fetched during the previous iteration (nr0 / nr1 above)
      ox = lane_value(nrec, 0), oy = lane_value(nrec, 1), oz = lane_value(nrec, 2), dx = lane_value(nrec, 3), dy = lane_value(nrec, 4);
      dz = lane_value(nrec, 5), near = lane_value(nrec, 6), far = lane_value(nrec, 7);
    } else {
      const float4 r0 = reinterpret_cast<const float4*>(ih.rays)[ray * 2];
      const float4 r1 = reinterpret_cast<const float4*>(ih.rays)[ray * 2 + 1];
      ox = r0.x, oy = r0.y, oz = r0.z, dx = r0.w, dy = r1.x, dz = r1.y, near = r1.z, far = r1.w;
    }
    const float* zsrc = from_jitter ? ih.jitter : ih.z_samp;
    const float* zrow = zsrc + ray * K;
    float z_cur = z_pre, zn_cur = zn_pre;
    // the samples (or the jitter) of what this wave evaluates next: they land while the current unit is evaluated
    auto prefetch_z = [&](int gg, int it) {
      const long row = lane_ray(gg, it) * K;
      const int kk = min(lane_k(it), K - 1);
      if (from_jitter) z_pre = ih.jitter[row + kk];
      else z_pre = ih.z_samp[row + kk], zn_pre = ih.z_samp[row + min(kk + 1, K - 1)];
    };
    const int g_next = group_of(idx + waves_per_xcd);
    if (!pk48 && g_next >= 0) prefetch_z(g_next, 0);
    if constexpr (ONE_RAY) {
      if (g_next >= 0) nrec = ih.rays[(long)g_next * 8 + (lane & 7)];
    }

    float T_carry = 1.0f, depth_part = 0.0f, w_part = 0.0f;
    float Tc_D = 1.0f;   // 48-lane mode: transmittance in front of the fourth ray's current row
    float rgb_part[NVMAX * 3];
#pragma unroll
    for (int i = 0; i < NVMAX * 3; ++i) rgb_part[i] = 0.0f;

    float *o_rgb = nullptr, *o_depth = nullptr;   // (read in the chunk loop with the other output pointers, used behind it as well)
    int white_bkgd = 0;
    for (int kc = 0; kc < (pk48 ? 192 : K); kc += 64) {
      const int it = kc >> 6;
      const int k = pk48 ? lane_k(it) : kc + kl;
      const bool valid = k < K;
      if (pk48) {
        ray = lane_ray(g, it);
        zrow = zsrc + ray * K;
        if (it > 0) z_cur = z_pre, zn_cur = zn_pre;
        if (it < 2) prefetch_z(g, it + 1);
        else if (g_next >= 0) prefetch_z(g_next, 0);
        if (mainl && it > 0) {   // lanes 0-47 start a new ray
          depth_part = 0.0f, w_part = 0.0f;
#pragma unroll
          for (int i = 0; i < NVMAX * 3; ++i) rgb_part[i] = 0.0f;
        }
      }
      // The weights in LDS are the same for every ray: without this the compiler hoists all ~150 weight reads out of the persistent
      // loop and keeps them in VGPRs (then spills the gather buffers).  Make the LDS offsets opaque per iteration.
      int lane_off = lane_off0, h = h0;
      asm volatile("" : "+v"(lane_off), "+v"(h));
      if (kc > 0 && !pk48) {
        const int kk = valid ? k : K - 1;
        z_cur = zrow[kk];
        if (!from_jitter) zn_cur = zrow[min(kk + 1, K - 1)];
      }
      if (from_jitter) {
        // NeRFRenderer.sample_coarse in here (nerf.py:103-123; bit-identical to bts_sample_coarse: the same routine): one launch and
        // 8 B per sample of HBM traffic less per render.  The next sample's depth is the neighbour lane's (rays of more than 64
        // samples: computed from its own jitter, the neighbour of lane 63 belongs to the next chunk).
        const int kk = valid ? k : K - 1;
        const bool lindisp = q->lindisp != 0;
        z_cur = coarse_depth(z_cur, (kc == 0 && !pk48) ? base0 : coarse_base(K, kk), step0, near, far, lindisp);
        zn_cur = dpp_f<kDppWaveShl1>(z_cur, z_cur);
        if (K > 64 || pk48) zn_cur = coarse_depth(zrow[min(kk + 1, K - 1)], coarse_base(K, min(kk + 1, K - 1)), step0, near, far, lindisp);
        if (q->z_out && valid) q->z_out[ray * K + k] = z_cur;
      }
      const float z = z_cur, z_nx = zn_cur;
      // nerf.py:231  points = o + z * d   (mul, then add)
      const float px = ox + z * dx, py = oy + z * dy, pz = oz + z * dz;
      if constexpr (!ONE_RAY) {
        if (pk48 && it < 2) {   // the next iteration's ray (lanes 0-47 change theirs): its registers are free from here on
          const long rn = lane_ray(g, it + 1);
          const float4 r0 = reinterpret_cast<const float4*>(ih.rays)[rn * 2];
          const float4 r1 = reinterpret_cast<const float4*>(ih.rays)[rn * 2 + 1];
          ox = r0.x, oy = r0.y, oz = r0.z, dx = r0.w, dy = r1.x, dz = r1.y, near = r1.z, far = r1.w;
        }
      }

      // ---------------- encoder view: projection, taps, depth code
      const Proj pe = ih.code_mode == 1 ? project<true>(enc, px, py, pz) : project<false>(enc, px, py, pz);
      Taps tp = make_taps(pe.x, pe.y, H, W, fs);
      float v3[3];
      v3[0] = pe.x, v3[1] = pe.y;
      v3[2] = depth_code(pe, ih.code_mode == 1, ih.inv_z != 0, ih.inv_dmax, ih.inv_range, ih.d_min, ih.range);
      const bool use_empty = (ih.learn_empty != 0) & pe.invalid;
      const Taps tp_enc = tp;   // as grid_sample has them: a render view that IS the encoder view (FwdParams::enc_view) takes its colour taps from here
      if (use_empty) tp.w00 = tp.w01 = tp.w10 = tp.w11 = 0.0f;  // the empty feature is added after the blend
      if constexpr (F16) tp.w00 *= scale, tp.w01 *= scale, tp.w10 *= scale, tp.w11 *= scale;  // exact: power of two

      float col[NVMAX * 3];
      bool inv[NVMAX];

      // ---------------- tap offsets / weights of both point tiles on every lane
      int o[2][4];
      float wq[2][4];
      bool emp[2];
      {
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
      }


      float s_raw;
      if (__builtin_expect(__any(pe_needs_exact(v3, ih.freq_factor)), 0)) {
        s_raw = eval_point_exact<C, HD, NB>(lds, G, q->w2c_enc + sample * 16, q->K_enc + sample * 9, H, W, fs, q->code_mode, q->inv_z, q->inv_dmax,
                                            q->inv_range, q->d_min, q->range, q->freq_factor, q->learn_empty, b_out, px, py, pz);
      } else {
      BTS_TICK(0)
      // ---------------- h = bilinear(G) + W_pe . PE + b: gather two stages ahead, blend between the octaves
      f32x16 acc[HT][2];
#ifdef BTS_GATHER_LDS
      static_assert(F16, "the LDS gather is wired into the f16 path only");
      unsigned off_next[4];
      GRows rows;
      {
        // tap table of the wave's 64 samples (byte offsets into G), lane = sample
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        gl.tab[lane * 3 + 0] = (unsigned)tp.o00 * (HD * 4u), gl.tab[lane * 3 + 1] = (unsigned)tp.o01 * (HD * 4u), gl.tab[lane * 3 + 2] = (unsigned)tp.o10 * (HD * 4u);   // o11 = o10 + (o01 - o00)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        gl_prologue<HD>(gl, rows, G, off_next);
      }
#else
      GBuf ba, bb;
      const bool nogather = BTS_ABL(1);
      if (!nogather) {
        stage_load<HD, 0>(ba, G, o, h);
        stage_load<HD, 1>(bb, G, o, h);
      }
#endif
      if constexpr (F16) {
        // bias row (times 2^S, fp32): the C operand of the first MFMA of every accumulator tile.  The raw inputs x, y, code ride in
        // the spare k rows of the f16 slices (|x|, |y| <= 2083 here -- beyond that the wave took the exact path above), so the
        // fp32-input MFMAs, which block the VALU for 64 cycles each, are gone from this path.
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
        asm volatile("" : "+v"(lane4));  // keep the A-operand reads inside the loop (see lane_off above)
#ifdef BTS_GATHER_LDS
        region_seq_l<HD, 0>(acc, gl, rows, G, wq, off_next, lh + LH::W_F16 + lane4, LH::TERM_STRIDE, raw, v3, ih.freq_factor, bias);
        if constexpr (NS > kNumFreqs) {   // HD = 64: the blocks of stages 6 and 7
          gl_consume<HD, 12>(acc, gl, rows, G, wq, off_next), gl_consume<HD, 13>(acc, gl, rows, G, wq, off_next);
          gl_consume<HD, 14>(acc, gl, rows, G, wq, off_next), gl_consume<HD, 15>(acc, gl, rows, G, wq, off_next);
        }
#else
        region_seq<HD, 0>(acc, ba, bb, G, o, wq, h, lh + LH::W_F16 + lane4, LH::TERM_STRIDE, raw, v3, q->freq_factor, bias, nosin, nomfma);
#endif
      }
#ifndef BTS_GATHER_LDS
      if constexpr (NS > kNumFreqs) {  // HD = 64: stages 6 and 7 are still in the buffers
        if (!nogather) {
          stage_blend<HD, 6>(acc, ba, wq);
          stage_blend<HD, 7>(acc, bb, wq);
        }
      }
#endif
      if (q->learn_empty && __any(use_empty)) {
#pragma unroll
        for (int ht = 0; ht < HT; ++ht)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const float ev = F16 ? lh[LH::EMPTY + ht * 32 + mfma_row(q, 0) + 4 * h] : lds[L::EMPTY + ht * 32 + mfma_row(q, 0) + 4 * h];
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) acc[ht][pt][q] += emp[pt] ? ev : 0.0f;
          }
      }

      // ---------------- ResnetBlockFC layers: h = h + fc_1(relu(fc_0(relu(h))))   (resnetfc.py:53-62)
      int lane4b = lane * 4;
      asm volatile("" : "+v"(lane4b));   // keep the weight reads inside the persistent loop (see lane_off)
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float* base = lds + L::BLK + b * L::BLK_STRIDE;
        f32x16 net[HT][2];
#pragma unroll
        for (int ot = 0; ot < HT; ++ot)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int row = ot * 32 + mfma_row(q, 0) + 4 * h;
            const float bias = F16 ? lh[LH::BIAS + b * 2 * HD + row] : base[HD * HD + row];
            net[ot][0][q] = bias, net[ot][1][q] = bias;
          }
        if constexpr (F16 && HD == 32) hidden_layer_h(net, acc, lh + LH::W_BLK + (2 * b) * LH::BLK_LAYER_STRIDE + lane4b, LH::BLK_TERM_STRIDE, inv_scale);
        else hidden_layer<HD>(net, acc, base, lane);
#pragma unroll
        for (int ot = 0; ot < HT; ++ot)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int row = ot * 32 + mfma_row(q, 0) + 4 * h;
            const float bias = F16 ? lh[LH::BIAS + b * 2 * HD + HD + row] : base[2 * HD * HD + HD + row];
            acc[ot][0][q] += bias, acc[ot][1][q] += bias;
          }
        if constexpr (F16 && HD == 32) hidden_layer_h(acc, net, lh + LH::W_BLK + (2 * b + 1) * LH::BLK_LAYER_STRIDE + lane4b, LH::BLK_TERM_STRIDE, inv_scale);
        else hidden_layer<HD>(acc, net, base + HD * HD + HD, lane);
      }

      BTS_TICK(1)
      // ---------------- lin_out: in-lane dot over the hidden rows this lane holds, then fold the two lane halves
      float p0 = 0.0f, p1 = 0.0f;
      if (BTS_ABL(32)) {
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) p0 += acc[ht][0][0] + acc[ht][0][5] + acc[ht][0][15], p1 += acc[ht][1][0] + acc[ht][1][5] + acc[ht][1][15];
      } else {
        // both point tiles' running sums as ONE packed FMA per hidden row (the same two chains p0, p1 as scalar code: order unchanged)
        f32x2 pp = {0.0f, 0.0f};
#pragma unroll
        for (int ht = 0; ht < HT; ++ht)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const float w2 = lds[L::W_OUT + ht * 32 + mfma_row(q, 0) + 4 * h];
            // w2 as src0: a broadcast from an odd register in src1 is the gfx950 op_sel erratum (tools/check_pk_opsel.py)
            pp = __builtin_elementwise_fma((f32x2){w2, w2}, (f32x2){relu1(acc[ht][0][q]), relu1(acc[ht][1][q])}, pp);
          }
        p0 = pp[0], p1 = pp[1];
      }
      swap32(p0, p1);  // p0 = {tile0.lo, tile1.lo}, p1 = {tile0.hi, tile1.hi}: lane l now holds both halves of ITS sample
      s_raw = F16 ? __builtin_fmaf(p0 + p1, inv_scale, b_out) : (p0 + p1) + b_out;
      }
      float sigma = softplus(s_raw);
      if (q->empty_empty) sigma = pe.invalid ? 0.0f : sigma;
      if (q->sigma_noise) sigma += q->sigma_noise[ray * K + min(k, K - 1)];   // nerf.py:279-280, the caller drew it
      BTS_TICK(2)

      // ---------------- the iteration's ten output pointers in ONE batch of scalar loads (they are neighbours in the kernarg segment),
      // issued here so that the round trip runs under the colour taps.  Read where they are used -- each inside its own `if (pointer)` --
      // they were ten dependent load / wait / branch steps behind each other in the store section (profiles/r04t: 3.8 k of the training
      // forward's 28 k cycles per iteration).
      o_rgb = q->rgb, o_depth = q->depth, white_bkgd = q->white_bkgd;
      float *o_weights = q->weights, *o_alphas = q->alphas, *o_invalid = q->invalid, *o_rgb_samps = q->rgb_samps;
      float *o_sigma_raw = q->sigma_raw, *o_trans = q->trans, *o_iw = q->invalid_wsum, *o_ia = q->invalid_any;
      asm volatile("" : "+s"(o_rgb), "+s"(o_depth), "+s"(o_weights), "+s"(o_alphas), "+s"(o_invalid), "+s"(o_rgb_samps), "+s"(o_sigma_raw), "+s"(o_trans),
                   "+s"(o_iw), "+s"(o_ia));

      // ---------------- colours (models_bts.py:218-264): projection into each render view + 4-tap fetch of the rgb0-packed frame.
      // Issued here, after lin_out, rather than before the MFMA phase: the taps' registers are not live across the accumulators
      // (0 spilled VGPRs, 5 % faster).  Round 1 had this order fail parity for nv <= 2 -- that was the packed-FP32 operand-select
      // erratum of gfx950 (tools/ubench/pk_opsel_lanes.hip), not the order; see DESIGN.md section 3.
      // All views of a batch (four at a time) go through the three phases together -- cameras + projections + taps, then the 4 x 4
      // texel loads, then the blends: view after view, each view's loads were waited for before the next view's went out (one memory
      // round trip per view: 25 % of the training forward's iteration at nv = 4, profiles/r04t).  Views beyond nv repeat view nv - 1 so
      // that no branch sits between the loads (a uniform branch per view splits them into blocks the scheduler cannot merge); their
      // results are dropped.
#pragma unroll
      for (int j = 0; j < NVMAX; ++j) col[3 * j] = col[3 * j + 1] = col[3 * j + 2] = 0.0f, inv[j] = pe.invalid;
      if (nv > 0 && !BTS_ABL(8)) {
#pragma unroll
        for (int j0 = 0; j0 < NVMAX; j0 += 4) {
          constexpr int NB4 = NVMAX < 4 ? NVMAX : 4;
          Taps tcs[NB4];
          bool invs[NB4];
#pragma unroll
          for (int b = 0; b < NB4; ++b) {
            const int jj = min(j0 + b, nv - 1);   // uniform
            tcs[b] = tp_enc, invs[b] = pe.invalid;
            if (jj != q->enc_view) {   // wave-uniform
              const Cam cj = load_cam(q->w2c_r + ((long)sample * nv + jj) * 16, q->K_r + ((long)sample * nv + jj) * 9);
              const Proj pc = project<false>(cj, px, py, pz);
              tcs[b] = make_taps(pc.x, pc.y, H, W);
              invs[b] = pc.invalid | pe.invalid;
            }
          }
          float4 tex[NB4][4];
#pragma unroll
          for (int b = 0; b < NB4; ++b) {
            const float4* img = reinterpret_cast<const float4*>(q->imgs) + ((long)sample * nv + min(j0 + b, nv - 1)) * H * W;
            tex[b][0] = img[tcs[b].o00], tex[b][1] = img[tcs[b].o01], tex[b][2] = img[tcs[b].o10], tex[b][3] = img[tcs[b].o11];
          }
          if constexpr (NB4 > 1) __builtin_amdgcn_sched_barrier(0);   // every load of the batch is out before the first blend
#pragma unroll
          for (int b = 0; b < NB4; ++b) {
            const int j = j0 + b;
            const Taps& tc = tcs[b];
            const float4 a = tex[b][0], bb = tex[b][1], cc = tex[b][2], d = tex[b][3];
            const float c0 = ((a.x * tc.w00 + bb.x * tc.w01) + cc.x * tc.w10) + d.x * tc.w11;
            const float c1 = ((a.y * tc.w00 + bb.y * tc.w01) + cc.y * tc.w10) + d.y * tc.w11;
            const float c2 = ((a.z * tc.w00 + bb.z * tc.w01) + cc.z * tc.w10) + d.z * tc.w11;
            if (j < nv) col[3 * j + 0] = c0, col[3 * j + 1] = c1, col[3 * j + 2] = c2, inv[j] = invs[b];
          }
        }
      }

      // ---------------- alpha compositing (nerf.py:225-299): segmented DPP scan over the lanes of each ray
      const float delta = (k + 1 < K) ? (z_nx - z) : 1e10f;
      float alpha = 1.0f - transmittance(delta, sigma);
      if (q->hard_cap && k == K - 1) alpha = 1.0f;
      const float t = valid ? (1.0f - alpha) + 1e-10f : 1.0f;
      const float incl = seg_scan_mul(t, lpr, kl);
      float excl = dpp_f<kDppWaveShr1>(1.0f, incl);
      if (kl == 0) excl = 1.0f;   // (48-lane mode: lanes 0 and 48)
      const float T = (pk48 ? (mainl ? 1.0f : Tc_D) : T_carry) * excl;
      if (pk48) Tc_D = Tc_D * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, incl), 63));
      if (ONE_RAY && K > 64) T_carry = T_carry * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, incl), 63));
      const float wgt = valid ? alpha * T : 0.0f;
      BTS_TICK(3)
      // ---------------- per-ray reductions for the loss' invalid-ray policies (loss.py:100-118), instead of weights + invalid in HBM
      if constexpr (EPI) {
        // every view's sum and flag first (no branch between the scans), then ONE block of stores on the ray's last lane: view after
        // view -- scan, ballot, lane branch, two pointer branches, two one-lane stores -- was 16 branches per iteration at nv = 4
        float ws[NVMAX], any[NVMAX];
        const unsigned long long seg = pk48 ? (mainl ? 0x0000FFFFFFFFFFFFull : 0xFFFF000000000000ull)
                                            : (lpr == 64 ? ~0ull : (((1ull << lpr) - 1ull) << ((lane | (lpr - 1)) - (lpr - 1))));
#pragma unroll
        for (int j = 0; j < NVMAX; ++j) {
          ws[j] = seg_scan_add((valid && inv[j]) ? wgt : 0.0f, lpr, kl);       // the last lane of each ray has the sum
          const unsigned long long hit = __ballot(valid && inv[j]);
          any[j] = (hit & seg) ? 1.0f : 0.0f;
        }
        if (pk48 ? (lane == 47 || lane == 63) : kl == lpr - 1) {
          const bool more = pk48 ? (!mainl && it > 0) : kc > 0;   // K > 64: chunk after chunk; 48-lane mode: the fourth ray's rows
          const long idx = ray * nv;
          if (nv == NVMAX && NVMAX % 4 == 0 && !more) {   // the common case: whole rows, nothing to merge with
#pragma unroll
            for (int j4 = 0; j4 < NVMAX / 4; ++j4) {
              if (o_iw) reinterpret_cast<float4*>(o_iw + idx)[j4] = make_float4(ws[4 * j4], ws[4 * j4 + 1], ws[4 * j4 + 2], ws[4 * j4 + 3]);
              if (o_ia) reinterpret_cast<float4*>(o_ia + idx)[j4] = make_float4(any[4 * j4], any[4 * j4 + 1], any[4 * j4 + 2], any[4 * j4 + 3]);
            }
          } else {
#pragma unroll
            for (int j = 0; j < NVMAX; ++j)
              if (j < nv) {
                if (o_iw) o_iw[idx + j] = (more ? o_iw[idx + j] : 0.0f) + ws[j];
                if (o_ia) o_ia[idx + j] = more ? fmaxf(o_ia[idx + j], any[j]) : any[j];
              }
          }
        }
      }
      depth_part = depth_part + wgt * z;
      w_part = w_part + wgt;
#pragma unroll
      for (int i = 0; i < NVMAX * 3; ++i) rgb_part[i] = rgb_part[i] + wgt * col[i];
      if (valid && !BTS_ABL(16)) {
        const long pk = ray * K + k;
        if (o_weights) o_weights[pk] = wgt;
        if (o_alphas) o_alphas[pk] = alpha;
        if (o_sigma_raw) o_sigma_raw[pk] = s_raw;
        if (o_trans) o_trans[pk] = T;
        if (o_invalid) {
#pragma unroll
          for (int j = 0; j < NVMAX; ++j)
            if (j < nv) o_invalid[pk * nv + j] = inv[j] ? 1.0f : 0.0f;
        }
        if (o_rgb_samps) {
          // a sample's nv * 3 colours are contiguous: with every view present they leave as 16- (or 8-) byte pieces -- as 4-byte stores
          // 48 bytes apart between lanes, every instruction touched 24 lines for 4 bytes each (12 of them per sample at nv = 4)
          float* dst = o_rgb_samps + pk * (nv * 3);
          if (nv == NVMAX && (NVMAX * 3) % 4 == 0) {
#pragma unroll
            for (int i = 0; i < NVMAX * 3 / 4; ++i)
              reinterpret_cast<float4*>(dst)[i] = make_float4(col[4 * i], col[4 * i + 1], col[4 * i + 2], col[4 * i + 3]);
          } else if (nv == NVMAX && (NVMAX * 3) % 2 == 0) {
#pragma unroll
            for (int i = 0; i < NVMAX * 3 / 2; ++i) reinterpret_cast<float2*>(dst)[i] = make_float2(col[2 * i], col[2 * i + 1]);
          } else {
#pragma unroll
            for (int i = 0; i < NVMAX * 3; ++i)
              if (i < nv * 3) dst[i] = col[i];
          }
        }
      }
      if (pk48) {
        // per-ray sums of this iteration: lane 47 has the totals of lanes 0-47's ray; lane 63 those of the fourth ray's rows so far
        // (its lanes keep accumulating), written after its last row
        const float dsum = seg_scan_add(depth_part, 48, kl), wsum = seg_scan_add(w_part, 48, kl);
        float csum[NVMAX * 3];
#pragma unroll
        for (int i = 0; i < NVMAX * 3; ++i) csum[i] = i < nv * 3 ? seg_scan_add(rgb_part[i], 48, kl) : 0.0f;
        if (lane == 47 || (lane == 63 && it == 2)) {
          o_depth[ray] = dsum;
#pragma unroll
          for (int i = 0; i < NVMAX * 3; ++i)
            if (i < nv * 3) o_rgb[ray * nv * 3 + i] = white_bkgd ? (csum[i] + 1.0f) - wsum : csum[i];  // nerf.py:301-304
        }
      }
    }
    BTS_TICK(4)
    // ---------------- per-ray sums: the last lane of each ray ends up with the totals
    if (!pk48) {
      depth_part = seg_scan_add(depth_part, lpr, kl);
      w_part = seg_scan_add(w_part, lpr, kl);
#pragma unroll
      for (int i = 0; i < NVMAX * 3; ++i)
        if (i < nv * 3) rgb_part[i] = seg_scan_add(rgb_part[i], lpr, kl);
      if (kl == lpr - 1) {
        o_depth[ray] = depth_part;
        float out[NVMAX * 3];
#pragma unroll
        for (int i = 0; i < NVMAX * 3; ++i) out[i] = white_bkgd ? (rgb_part[i] + 1.0f) - w_part : rgb_part[i];  // nerf.py:301-304
        if (nv == NVMAX && (NVMAX * 3) % 4 == 0) {   // one lane, whole row: 16-byte stores
#pragma unroll
          for (int i = 0; i < NVMAX * 3 / 4; ++i)
            reinterpret_cast<float4*>(o_rgb + ray * (NVMAX * 3))[i] = make_float4(out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3]);
        } else {
#pragma unroll
          for (int i = 0; i < NVMAX * 3; ++i)
            if (i < nv * 3) o_rgb[ray * nv * 3 + i] = out[i];
        }
      }
    }
    BTS_TICK(5)
  }
#ifdef BTS_PROBE
  if ((p.ablate & 128) && p.dbg && lane == 0) {
    unsigned long long* d = p.dbg + ((long)blockIdx.x * 4 + wave) * 8;
#pragma unroll
    for (int i = 0; i < 6; ++i) d[i] = t_acc[i];
    d[6] = __builtin_readcyclecounter() - t_begin;
  }
#endif
}

#ifndef BTS_NO_LAUNCH_GLUE
template <int C, int HD, int NB, int NVMAX, bool EPI>
static int launch_render_p_one(const FwdParams& p, int grid, hipStream_t s) {
#ifdef BTS_GATHER_LDS
  constexpr int dyn = 4 * kGatherLdsPerWave;
  auto go = [&](auto kern) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
    kern<<<grid, 256, dyn, s>>>(p);
  };
  if (p.lpr == 64) go(render_kernel_p<C, HD, NB, NVMAX, true, true, EPI>);
  else go(render_kernel_p<C, HD, NB, NVMAX, false, true, EPI>);
#else
  if (p.lpr == 64) render_kernel_p<C, HD, NB, NVMAX, true, true, EPI><<<grid, 256, 0, s>>>(p);
  else render_kernel_p<C, HD, NB, NVMAX, false, true, EPI><<<grid, 256, 0, s>>>(p);
#endif
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: kernel launch failed (%ld)", hipGetErrorString(e), (long)e);
    return BTS_E_LAUNCH;
  }
  return BTS_OK;
}

template <int C, int HD, int NB, bool EPI>
static int launch_render_p_nv(const FwdParams& p, int grid, hipStream_t s) {
  if (p.nv <= 1) return launch_render_p_one<C, HD, NB, 1, EPI>(p, grid, s);
  if (p.nv <= 2) return launch_render_p_one<C, HD, NB, 2, EPI>(p, grid, s);
  if (p.nv <= 4) return launch_render_p_one<C, HD, NB, 4, EPI>(p, grid, s);
  return launch_render_p_one<C, HD, NB, 8, EPI>(p, grid, s);
}

template <bool EPI>
inline int launch_render_p(const FwdParams& p, int C, int HD, int NB, int grid, hipStream_t s) {
  if (C == 64 && HD == 64 && NB == 0) return launch_render_p_nv<64, 64, 0, EPI>(p, grid, s);
  if (C == 32 && HD == 32 && NB == 1) return launch_render_p_nv<32, 32, 1, EPI>(p, grid, s);
  if (C == 32 && HD == 32 && NB == 0) return launch_render_p_nv<32, 32, 0, EPI>(p, grid, s);
  set_error("%s: unsupported MLP shape C=%ld d_hidden=%ld n_blocks=%ld", "bts", C, HD, NB);
  return BTS_E_UNSUPPORTED;
}
#endif  // BTS_NO_LAUNCH_GLUE

}  // namespace bts
