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

namespace bts {

// ---- DPP (data-parallel primitives) helpers: lanes whose source is outside its row / whose row is masked keep `old` ----------
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL,
                                                               ROW_MASK, 0xF, false));
}
constexpr int kDppRowShr = 0x110;   // + n
constexpr int kDppWaveShr1 = 0x138;
constexpr int kDppBcast15 = 0x142;  // lane 15 of each row -> every lane of the next row
constexpr int kDppBcast31 = 0x143;  // lane 31 -> rows 2, 3

// inclusive prefix product over each group of lpr consecutive lanes (lpr in {8, 16, 32, 64}); kl = lane & (lpr - 1)
__device__ __forceinline__ float seg_scan_mul(float x, int lpr, int kl) {
  float y;
  y = dpp_f<kDppRowShr + 1>(1.0f, x), x *= (kl >= 1) ? y : 1.0f;
  y = dpp_f<kDppRowShr + 2>(1.0f, x), x *= (kl >= 2) ? y : 1.0f;
  y = dpp_f<kDppRowShr + 4>(1.0f, x), x *= (kl >= 4) ? y : 1.0f;
  if (lpr >= 16) y = dpp_f<kDppRowShr + 8>(1.0f, x), x *= y;
  if (lpr >= 32) x *= dpp_f<kDppBcast15, 0xA>(1.0f, x);
  if (lpr >= 64) x *= dpp_f<kDppBcast31, 0xC>(1.0f, x);
  return x;
}
// inclusive prefix sum; the LAST lane of each group holds the group's total
__device__ __forceinline__ float seg_scan_add(float x, int lpr, int kl) {
  float y;
  y = dpp_f<kDppRowShr + 1>(0.0f, x), x += (kl >= 1) ? y : 0.0f;
  y = dpp_f<kDppRowShr + 2>(0.0f, x), x += (kl >= 2) ? y : 0.0f;
  y = dpp_f<kDppRowShr + 4>(0.0f, x), x += (kl >= 4) ? y : 0.0f;
  if (lpr >= 16) x += dpp_f<kDppRowShr + 8>(0.0f, x);
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

template <int HD, int S>
__device__ __forceinline__ void stage_load(GBuf& b, const float4* __restrict__ G, const int (&o)[2][4], int h) {
  using St = GStage<HD, S>;
  gload<HD>(b, G, o[St::pt], St::tp2, St::ht * 8 + 4 * h);
}
template <int HD, int S>
__device__ __forceinline__ void stage_blend(f32x16 (&acc)[HD / 32][2], const GBuf& b, const float (&w)[2][4]) {
  using St = GStage<HD, S>;
  gblend<false>(acc[St::ht][St::pt], b, w[St::pt][2 * St::tp2], w[St::pt][2 * St::tp2 + 1]);
}

// octave OCT of the positional encoding with gather stage OCT blended behind its MFMAs (buffers alternate by parity).
// EXACT selects libm sines (wave-level slow path for arguments beyond the fast range); the fast variant is branch-free so that the
// whole gather + encoding + MFMA phase is ONE basic block the scheduler can interleave.
template <bool EXACT>
__device__ __forceinline__ void pe_octave_sel(float (&sc)[6], const float (&v3)[3], float ff, bool nosin) {
  if (nosin) {
#pragma unroll
    for (int i = 0; i < 6; ++i) sc[i] = v3[i % 3] * ff;
  } else if constexpr (EXACT) {
    pe_octave_exact(sc, v3, ff);
  } else {
    pe_octave_fast(sc, v3, ff);
  }
}

template <int HD, int OCT, bool EXACT>
__device__ __forceinline__ void octave_seq(f32x16 (&acc)[HD / 32][2], GBuf& ba, GBuf& bb, const float4* __restrict__ G,
                                           const int (&o)[2][4], const float (&wq)[2][4], int h, const float* wl, float (&sc)[6],
                                           const float (&v3)[3], float ff, bool nomfma, bool nosin, bool nogather) {
  constexpr int NS = 4 * (HD / 32);
  if constexpr (OCT < kNumFreqs) {
    // one scheduling region per octave: the sines of octave OCT+1, the 12 MFMAs of octave OCT, the blend of gather stage OCT and
    // the loads of stage OCT+2 may interleave freely; nothing moves across the region boundary (bounds the live ranges)
    float sn[6];
    if constexpr (OCT + 1 < kNumFreqs) pe_octave_sel<EXACT>(sn, v3, ff * 2.0f, nosin);
    kstep<HD>(acc, wl, 0, sc[0], sc[1], nomfma);
    kstep<HD>(acc, wl + 2 * HD, 0, sc[2], sc[3], nomfma);
    kstep<HD>(acc, wl + 4 * HD, 0, sc[4], sc[5], nomfma);
    if constexpr (OCT < NS) {
      if (!nogather) {
        stage_blend<HD, OCT>(acc, ba, wq);
        if constexpr (OCT + 2 < NS) stage_load<HD, OCT + 2>(ba, G, o, h);
      }
    }
    if constexpr (OCT + 1 < kNumFreqs) {
#pragma unroll
      for (int i = 0; i < 6; ++i) sc[i] = sn[i];
    }
    __builtin_amdgcn_sched_barrier(0);
    octave_seq<HD, OCT + 1, EXACT>(acc, bb, ba, G, o, wq, h, wl + 6 * HD, sc, v3, ff * 2.0f, nomfma, nosin, nogather);
  }
}

// Cold path: a wave in which some sample's encoding argument leaves the fast sincos range (|arg| > 1e5: points within millimetres
// of the encoder's camera plane) evaluates that iteration with the compact lane = point routine (libm range reduction inside).
// Kept out of line -- inlining 36 libm sines next to the pipelined path costs ~240 spilled VGPRs on the HOT path.
template <int C, int HD, int NB>
__device__ __attribute__((noinline)) float eval_point_exact(const float* lds, const float4* G, const float* w2c, const float* Kc, int H, int W,
                                                            int code_mode, int inv_z, float inv_dmax, float inv_range, float d_min, float range,
                                                            float freq_factor, int learn_empty, float b_out, float px, float py, float pz) {
  FwdParams q;
  q.H = H, q.W = W, q.code_mode = code_mode, q.inv_z = inv_z, q.inv_dmax = inv_dmax, q.inv_range = inv_range, q.d_min = d_min;
  q.range = range, q.freq_factor = freq_factor, q.learn_empty = learn_empty, q.ablate = 0;
  const Cam enc = load_cam(w2c, Kc);
  Proj pe;
  return eval_point<C, HD, NB, true>(q, lds, enc, G, (int)(threadIdx.x & 63), b_out, px, py, pz, pe);
}

template <int C, int HD, int NB, int NVMAX, bool ONE_RAY>
__global__ __launch_bounds__(256, 2) void render_kernel_p(const FwdParams p) {
  using L = Lds<C, HD, NB, true>;
  constexpr int HT = HD / 32;
  constexpr int NS = 4 * HT;
  constexpr bool EARLY_COL = NVMAX <= 2;  // colour taps issued before the MFMA phase (16 VGPRs per view)
  __shared__ float lds[L::TOTAL];
  stage_weights<C, HD, NB, true>(lds, p.mlp, p.empty_feature);
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int h0 = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwg = gridDim.x;  // multiple of 8
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int wg_per_xcd = nwg >> 3;
  const int xcd = wg / wg_per_xcd;
  const int lw = (wg - xcd * wg_per_xcd) * 4 + wave;  // wave index inside its XCD
  const int waves_per_xcd = wg_per_xcd * 4;
  const int lpr = ONE_RAY ? 64 : p.lpr, R = 64 / lpr;
  const int kl = lane & (lpr - 1);
  const long gx = (p.groups + 7) >> 3;
  const long g_end = min(p.groups, (xcd + 1) * gx);
  const int Bp = p.Bp, K = p.K, H = p.H, W = p.W, nv = p.nv;
  const float b_out = as_const(p.mlp)[MlpLayout{C + kPeDim, HD, NB}.b_out()];
  const int lane_off0 = h0 * HD + (lane & 31);
  const bool nomfma = BTS_ABL(4), nosin = BTS_ABL(2);

  long g = xcd * gx + lw;
  // z of the first ray group
  float z_pre = 0.0f, zn_pre = 0.0f;
  if (g < g_end) {
    const float* zr = p.z_samp + (g * R + lane / lpr) * K;
    const int kk = min(kl, K - 1);
    z_pre = zr[kk], zn_pre = zr[min(kk + 1, K - 1)];
  }

  for (; g < g_end; g += waves_per_xcd) {
    const long ray = g * R + lane / lpr;
    const int sample = __builtin_amdgcn_readfirstlane((int)((g * R) / Bp));  // all rays of a group belong to one batch element
    const Cam enc = load_cam(p.w2c_enc + sample * 16, p.K_enc + sample * 9);
    const float4* __restrict__ G = reinterpret_cast<const float4*>(p.proj) + (long)sample * H * W * (HD / 4);
    float ox, oy, oz, dx, dy, dz;
    if constexpr (ONE_RAY) {  // wave-uniform ray: scalar loads
      const cfp rp = as_const(p.rays) + g * 8;
      ox = rp[0], oy = rp[1], oz = rp[2], dx = rp[3], dy = rp[4], dz = rp[5];
    } else {
      const float4 r0 = reinterpret_cast<const float4*>(p.rays)[ray * 2];
      const float4 r1 = reinterpret_cast<const float4*>(p.rays)[ray * 2 + 1];
      ox = r0.x, oy = r0.y, oz = r0.z, dx = r0.w, dy = r1.x, dz = r1.y;
    }
    const float* zrow = p.z_samp + ray * K;
    float z_cur = z_pre, zn_cur = zn_pre;
    {  // prefetch the next group's samples; they land while this group is evaluated
      const long gn = g + waves_per_xcd;
      if (gn < g_end) {
        const float* zr = p.z_samp + (gn * R + lane / lpr) * K;
        const int kk = min(kl, K - 1);
        z_pre = zr[kk], zn_pre = zr[min(kk + 1, K - 1)];
      }
    }

    float T_carry = 1.0f, depth_part = 0.0f, w_part = 0.0f;
    float rgb_part[NVMAX * 3];
#pragma unroll
    for (int i = 0; i < NVMAX * 3; ++i) rgb_part[i] = 0.0f;

    for (int kc = 0; kc < K; kc += 64) {
      const int k = kc + kl;
      const bool valid = k < K;
      // The weights in LDS are the same for every ray: without this the compiler hoists all ~150 weight reads out of the persistent
      // loop and keeps them in VGPRs (then spills the gather buffers).  Make the LDS offsets opaque per iteration.
      int lane_off = lane_off0, h = h0;
      asm volatile("" : "+v"(lane_off), "+v"(h));
      if (kc > 0) {
        const int kk = valid ? k : K - 1;
        z_cur = zrow[kk], zn_cur = zrow[min(kk + 1, K - 1)];
      }
      const float z = z_cur, z_nx = zn_cur;
      // nerf.py:231  points = o + z * d   (mul, then add)
      const float px = ox + z * dx, py = oy + z * dy, pz = oz + z * dz;

      // ---------------- encoder view: projection, taps, depth code
      const Proj pe = p.code_mode == 1 ? project<true>(enc, px, py, pz) : project<false>(enc, px, py, pz);
      Taps tp = make_taps(pe.x, pe.y, H, W);
      float v3[3];
      v3[0] = pe.x, v3[1] = pe.y;
      v3[2] = depth_code(p.code_mode == 1 ? pe.dist : pe.z, p.inv_z != 0, p.inv_dmax, p.inv_range, p.d_min, p.range);
      const bool use_empty = (p.learn_empty != 0) & pe.invalid;
      if (use_empty) tp.w00 = tp.w01 = tp.w10 = tp.w11 = 0.0f;  // the empty feature is added after the blend

      // ---------------- colour views: projection + taps; loads issued now for <= 2 views (models_bts.py:218-264)
      float col[NVMAX * 3];
      bool inv[NVMAX];
      float4 ct[EARLY_COL ? NVMAX : 1][4];
      float cw[EARLY_COL ? NVMAX : 1][4];
      if constexpr (EARLY_COL) {
#pragma unroll
        for (int j = 0; j < NVMAX; ++j) {
          inv[j] = pe.invalid;
#pragma unroll
          for (int t = 0; t < 4; ++t) ct[j][t] = make_float4(0.0f, 0.0f, 0.0f, 0.0f), cw[j][t] = 0.0f;
          if (j < nv && !BTS_ABL(8)) {
            const Cam cj = load_cam(p.w2c_r + ((long)sample * nv + j) * 16, p.K_r + ((long)sample * nv + j) * 9);
            const Proj pc = project<false>(cj, px, py, pz);
            const Taps tc = make_taps(pc.x, pc.y, H, W);
            const float4* img = reinterpret_cast<const float4*>(p.imgs) + ((long)sample * nv + j) * H * W;
            ct[j][0] = img[tc.o00], ct[j][1] = img[tc.o01], ct[j][2] = img[tc.o10], ct[j][3] = img[tc.o11];
            cw[j][0] = tc.w00, cw[j][1] = tc.w01, cw[j][2] = tc.w10, cw[j][3] = tc.w11;
            inv[j] = pc.invalid | pe.invalid;
          }
        }
      }

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
      if (__builtin_expect(__any(pe_needs_exact(v3, p.freq_factor)), 0)) {
        s_raw = eval_point_exact<C, HD, NB>(lds, G, p.w2c_enc + sample * 16, p.K_enc + sample * 9, H, W, p.code_mode, p.inv_z, p.inv_dmax,
                                            p.inv_range, p.d_min, p.range, p.freq_factor, p.learn_empty, b_out, px, py, pz);
      } else {
      // ---------------- h = bilinear(G) + W_pe . PE + b: gather two stages ahead, blend between the octaves
      f32x16 acc[HT][2];
#pragma unroll
      for (int ht = 0; ht < HT; ++ht)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[ht][pt][q] = 0.0f;
      GBuf ba, bb;
      const bool nogather = BTS_ABL(1);
      if (!nogather) {
        stage_load<HD, 0>(ba, G, o, h);
        stage_load<HD, 1>(bb, G, o, h);
      }
      const float* wl = lds + L::W_IN + lane_off;
      kstep<HD>(acc, wl, 0, v3[0], v3[1], nomfma);
      kstep<HD>(acc, wl + 2 * HD, 0, v3[2], 1.0f, nomfma);
      float sc[6];
      __builtin_amdgcn_sched_barrier(0);
      pe_octave_sel<false>(sc, v3, p.freq_factor, nosin);
      __builtin_amdgcn_sched_barrier(0);
      octave_seq<HD, 0, false>(acc, ba, bb, G, o, wq, h, wl + 4 * HD, sc, v3, p.freq_factor, nomfma, nosin, nogather);
      if constexpr (NS > kNumFreqs) {  // HD = 64: stages 6 and 7 are still in the buffers
        if (!nogather) {
          stage_blend<HD, 6>(acc, ba, wq);
          stage_blend<HD, 7>(acc, bb, wq);
        }
      }
      if (p.learn_empty && __any(use_empty)) {
#pragma unroll
        for (int ht = 0; ht < HT; ++ht)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const float ev = lds[L::EMPTY + ht * 32 + mfma_row(q, 0) + 4 * h];
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) acc[ht][pt][q] += emp[pt] ? ev : 0.0f;
          }
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
            const float bias = base[HD * HD + ot * 32 + mfma_row(q, 0) + 4 * h];
            net[ot][0][q] = bias, net[ot][1][q] = bias;
          }
        hidden_layer<HD>(net, acc, base, lane);
#pragma unroll
        for (int ot = 0; ot < HT; ++ot)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const float bias = base[2 * HD * HD + HD + ot * 32 + mfma_row(q, 0) + 4 * h];
            acc[ot][0][q] += bias, acc[ot][1][q] += bias;
          }
        hidden_layer<HD>(acc, net, base + HD * HD + HD, lane);
      }

      // ---------------- lin_out: in-lane dot over the hidden rows this lane holds, then fold the two lane halves
      float p0 = 0.0f, p1 = 0.0f;
      if (BTS_ABL(32)) {
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) p0 += acc[ht][0][0] + acc[ht][0][5] + acc[ht][0][15], p1 += acc[ht][1][0] + acc[ht][1][5] + acc[ht][1][15];
      } else {
#pragma unroll
        for (int ht = 0; ht < HT; ++ht)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const float w2 = lds[L::W_OUT + ht * 32 + mfma_row(q, 0) + 4 * h];
            p0 = __builtin_fmaf(fmaxf(acc[ht][0][q], 0.0f), w2, p0);
            p1 = __builtin_fmaf(fmaxf(acc[ht][1][q], 0.0f), w2, p1);
          }
      }
      swap32(p0, p1);  // p0 = {tile0.lo, tile1.lo}, p1 = {tile0.hi, tile1.hi}: lane l now holds both halves of ITS sample
      s_raw = (p0 + p1) + b_out;
      }
      float sigma = softplus(s_raw);
      if (p.empty_empty) sigma = pe.invalid ? 0.0f : sigma;

      // ---------------- colours
      if constexpr (EARLY_COL) {
#pragma unroll
        for (int j = 0; j < NVMAX; ++j) {
          col[3 * j + 0] = ((ct[j][0].x * cw[j][0] + ct[j][1].x * cw[j][1]) + ct[j][2].x * cw[j][2]) + ct[j][3].x * cw[j][3];
          col[3 * j + 1] = ((ct[j][0].y * cw[j][0] + ct[j][1].y * cw[j][1]) + ct[j][2].y * cw[j][2]) + ct[j][3].y * cw[j][3];
          col[3 * j + 2] = ((ct[j][0].z * cw[j][0] + ct[j][1].z * cw[j][1]) + ct[j][2].z * cw[j][2]) + ct[j][3].z * cw[j][3];
        }
      } else {
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
      }

      // ---------------- alpha compositing (nerf.py:225-299): segmented DPP scan over the lanes of each ray
      const float delta = (k + 1 < K) ? (z_nx - z) : 1e10f;
      float alpha = 1.0f - expf(-fabsf(delta) * fmaxf(sigma, 0.0f));
      if (p.hard_cap && k == K - 1) alpha = 1.0f;
      const float t = valid ? (1.0f - alpha) + 1e-10f : 1.0f;
      const float incl = seg_scan_mul(t, lpr, kl);
      float excl = dpp_f<kDppWaveShr1>(1.0f, incl);
      if (kl == 0) excl = 1.0f;
      const float T = T_carry * excl;
      if (ONE_RAY && K > 64) T_carry = T_carry * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, incl), 63));
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
    // ---------------- per-ray sums: the last lane of each ray ends up with the totals
    depth_part = seg_scan_add(depth_part, lpr, kl);
    w_part = seg_scan_add(w_part, lpr, kl);
#pragma unroll
    for (int i = 0; i < NVMAX * 3; ++i)
      if (i < nv * 3) rgb_part[i] = seg_scan_add(rgb_part[i], lpr, kl);
    if (kl == lpr - 1) {
      p.depth[ray] = depth_part;
#pragma unroll
      for (int i = 0; i < NVMAX * 3; ++i)
        if (i < nv * 3) p.rgb[ray * nv * 3 + i] = p.white_bkgd ? (rgb_part[i] + 1.0f) - w_part : rgb_part[i];  // nerf.py:301-304
    }
  }
}

#ifndef BTS_NO_LAUNCH_GLUE
template <int C, int HD, int NB, int NVMAX>
static int launch_render_p_one(const FwdParams& p, int grid, hipStream_t s) {
  if (p.lpr == 64) render_kernel_p<C, HD, NB, NVMAX, true><<<grid, 256, 0, s>>>(p);
  else render_kernel_p<C, HD, NB, NVMAX, false><<<grid, 256, 0, s>>>(p);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: kernel launch failed (%ld)", hipGetErrorString(e), (long)e);
    return BTS_E_LAUNCH;
  }
  return BTS_OK;
}

template <int C, int HD, int NB>
static int launch_render_p_nv(const FwdParams& p, int grid, hipStream_t s) {
  if (p.nv <= 1) return launch_render_p_one<C, HD, NB, 1>(p, grid, s);
  if (p.nv <= 2) return launch_render_p_one<C, HD, NB, 2>(p, grid, s);
  if (p.nv <= 4) return launch_render_p_one<C, HD, NB, 4>(p, grid, s);
  return launch_render_p_one<C, HD, NB, 8>(p, grid, s);
}

inline int launch_render_p(const FwdParams& p, int C, int HD, int NB, int grid, hipStream_t s) {
  if (C == 64 && HD == 64 && NB == 0) return launch_render_p_nv<64, 64, 0>(p, grid, s);
  if (C == 32 && HD == 32 && NB == 1) return launch_render_p_nv<32, 32, 1>(p, grid, s);
  if (C == 32 && HD == 32 && NB == 0) return launch_render_p_nv<32, 32, 0>(p, grid, s);
  set_error("%s: unsupported MLP shape C=%ld d_hidden=%ld n_blocks=%ld", "bts", C, HD, NB);
  return BTS_E_UNSUPPORTED;
}
#endif  // BTS_NO_LAUNCH_GLUE

}  // namespace bts
