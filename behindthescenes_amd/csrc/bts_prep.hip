// Hand-over from the (PyTorch) encoder, fused with the feature half of lin_in.
//
//   project_features:      G[pix][hid] = sum_c F[c][pix] * w_in[hid][c]          (N,C,H,W) -> (N,H,W,Hd)
//   project_features_bwd:  dF[c][pix]  = sum_hid dG[pix][hid] * w_in[hid][c]     (N,H,W,Hd) -> (N,C,H,W)
//                          dW[hid][c] += sum_pix dG[pix][hid] * F[c][pix]
//
// These are HBM-bound streaming kernels (read 4*C + write 4*Hd bytes per pixel); the contraction rides along on
// v_mfma_f32_32x32x2_f32 so that the pass costs what the plain NCHW<->NHWC transposes it replaces would cost.  Layouts
// are chosen so that every global access is a coalesced row: F is read along pixels (NCHW rows), G / dG along channels.
#include "bts_common.h"
#include <cstdlib>

namespace bts {

void set_error(const char* fmt, const char* a = "", long b = 0, long c = 0, long d = 0);

// texel (index inside its image) of slot i of the tile whose first texel is p0: bts_common.h, tile_rs
__device__ __forceinline__ int tile_px(int p0, int i, int rs) { return p0 + i + (i >> 4) * rs; }
// the same for a slot = cslot (wave-uniform, mostly a compile-time constant) + lslot (per lane, never carrying into bit 4 of the slot):
// the row piece (cslot >> 4) stays a uniform -- the address is scalar base + lane offset + immediate instead of one VGPR per access
__device__ __forceinline__ int tile_px_c(int p0, int cslot, int lslot, int rs) { return p0 + cslot + (cslot >> 4) * rs + lslot; }

// ---- forward: one wave = 64 pixels x all Hd outputs.  D[pix][hid] = A[pix][c] . B[c][hid]:
// A operand = F (lane l: pixel l&31, channel parity l>>5) read straight from the NCHW rows, B = w_in^T from LDS (k-major),
// D rows (pixels) sit in registers, columns (hidden) across lanes -> every store is a 128-byte row segment of G.
// Persistent work-groups (the weights are staged once per work-group, not once per 256 pixels) and ALL of a tile's loads issued
// before its first MFMA: round 3 kept 8 loads (2 KB) per wave in flight and ran at 45 % of HBM speed (latency-bound by Little's law);
// 64 loads per wave keep 16 KB in flight.
// FEAT_CL: F is channels-last, (N, H, W, C) in memory (a torch tensor of shape (N, C, H, W) in channels_last format: what MIOpen's NHWC
// kernels and bts_conv3x3_fwd produce): a lane reads ITS pixel's channels as float4 pieces -- k-step (q, e) pairs channel 8 q + e
// (lane half 0) with 8 q + 4 + e (half 1) -- and a tile is 64 x C x 4 contiguous bytes instead of C row pieces of 256.
template <int C, int HD, bool FEAT_CL>
__global__ __launch_bounds__(256, 2) void project_kernel(const float* __restrict__ feat, const float* __restrict__ mlp, float* __restrict__ proj,
                                                      int HW, int tiles_per_img, int n_tiles, const unsigned char* __restrict__ tiles, int Wm, int tw,
                                                      const int* __restrict__ list = nullptr) {
  constexpr int HT = HD / 32;
  constexpr int D_IN = C + kPeDim;
  __shared__ float wl[C * HD];  // wl[c*HD + s] = w_in[hidden_of_storage(s)][c]: G comes out in its storage channel order
  for (int i = threadIdx.x; i < C * HD; i += blockDim.x) {
    const int c = i / HD, st = i % HD;
    wl[i] = mlp[proj_hidden_of_storage(st) * D_IN + c];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, col = lane & 31;
  // `tiles` (bts_project_features_tiles): only the flagged tiles are computed -- a training step's samples read 38 % of them
  // (exp_kitti_360.yaml's batch), the rest of the map is never looked at and stays as it is.  A tile's flag is fetched one tile ahead.
  // `list` (bts_train_step_fwd, large maps): the flagged tiles as a compact list (compact_tiles_kernel), dealt round-robin -- every wave
  // the same number of tiles to within one instead of a binomial share of the flags it happens to walk over
  const int tile0 = blockIdx.x * 4 + wave, tstride = gridDim.x * 4;
  const int n_work = list ? min(n_tiles, list[0]) : n_tiles;
  int flag_n = list ? (tile0 < n_work ? list[4 + tile0] : 0) : ((tiles && tile0 < n_tiles) ? (int)tiles[tile0] : 1);
  for (int it = tile0; it < n_work; it += tstride) {   // tile = 64 pixels of one image
    const int fetched = __builtin_amdgcn_readfirstlane(flag_n);
    flag_n = list ? (it + tstride < n_work ? list[4 + it + tstride] : 0) : ((tiles && it + tstride < n_tiles) ? (int)tiles[it + tstride] : 1);
    if (!list && fetched == 0) continue;
    const int tile = list ? fetched : it;
    const int img = tile / tiles_per_img;
    // slot i of the tile is texel p0 + i + (i >> 4) * rs of the image (bts_common.h: a 16 x 4 block, or 64 consecutive texels with rs = 0)
    const int p0 = tile_base(tile - img * tiles_per_img, Wm, tw), rs = tile_rs(Wm, tw);
    const int px0 = p0 + col + (col >> 4) * rs, px1 = p0 + 32 + col + ((32 + col) >> 4) * rs;   // this lane's texels of the two point tiles
    const float* F = feat + (long)img * C * HW;
    float* G = proj + (long)img * HW * HD;
    f32x16 acc[2][HT];
    if constexpr (FEAT_CL) {
      const float4* F4 = reinterpret_cast<const float4*>(F);
      const unsigned q0 = (unsigned)(min(px0, HW - 1) * (C / 4) + h), q1 = (unsigned)(min(px1, HW - 1) * (C / 4) + h);
      float4 v0[C / 8], v1[C / 8];
#pragma unroll
      for (int q = 0; q < C / 8; ++q) v0[q] = F4[q0 + 2 * q], v1[q] = F4[q1 + 2 * q];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) acc[pt][ht] = zero_acc();
#pragma unroll
      for (int q = 0; q < C / 8; ++q) {
        const float* a0 = reinterpret_cast<const float*>(&v0[q]);
        const float* a1 = reinterpret_cast<const float*>(&v1[q]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = 8 * q + 4 * h + e;
#pragma unroll
          for (int ht = 0; ht < HT; ++ht) {
            const float b = wl[c * HD + ht * 32 + col];
            acc[0][ht] = mfma(a0[e], b, acc[0][ht]);
            acc[1][ht] = mfma(a1[e], b, acc[1][ht]);
          }
        }
      }
    } else {
    // wave-uniform base + 32-bit lane offset (one image's map is far below 4 GB): scalar-base loads, no 64-bit address per load
    const unsigned o0 = (unsigned)(h * HW + min(px0, HW - 1)), o1 = (unsigned)(h * HW + min(px1, HW - 1));
    float a0[C / 2], a1[C / 2];
#pragma unroll
    for (int s = 0; s < C / 2; ++s) {
      const float* row = F + (long)(2 * s) * HW;   // uniform
      a0[s] = row[o0], a1[s] = row[o1];
    }
    __builtin_amdgcn_sched_barrier(0);   // the scheduler otherwise sinks every load to its MFMA (two loads in flight, measured 2.9 TB/s)
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
      for (int ht = 0; ht < HT; ++ht) acc[pt][ht] = zero_acc();
#pragma unroll
    for (int s = 0; s < C / 2; ++s) {
      const int c = 2 * s + h;
#pragma unroll
      for (int ht = 0; ht < HT; ++ht) {
        const float b = wl[c * HD + ht * 32 + col];
        acc[0][ht] = mfma(a0[s], b, acc[0][ht]);
        acc[1][ht] = mfma(a1[s], b, acc[1][ht]);
      }
    }
    }
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int pix = tile_px_c(p0, pt * 32 + mfma_row(q, 0), 4 * h, rs);   // mfma_row(q, h) = mfma_row(q, 0) + 4 h, (.. & 15) <= 11
        if (pix < HW) {
#pragma unroll
          for (int ht = 0; ht < HT; ++ht) G[(unsigned)(pix * HD + ht * 32 + col)] = acc[pt][ht][q];
        }
      }
  }
}

// ---- backward: ONE kernel for both gradients, so that dG (the largest tensor of a training step: 503 MB at exp_kitti_360.yaml's
// batch) leaves HBM once.  The waves of a work-group split the roles and walk the same tiles of 64 pixels:
//   waves 0, 1 (when the weight gradient is wanted): dW[hid][c] += sum_pix dG[pix][hid] F[c][pix]   (contraction over pixels)
//   waves 2, 3 (when the feature gradient is wanted): dF[c][pix] = sum_hid dG[pix][hid] w_in[hid][c]
// so a tile of dG is fetched by one role and found in L2 by the other (round 3 ran two kernels: 2 x 503 MB of dG reads).
//
// dW: A[i = stored channel][k = pix] comes straight from dG (32 lanes = 32 consecutive channels of one pixel: a 128-byte row segment);
//     B[k = pix][j = c] = F[c][pix]: lane (c, half h) reads FOUR consecutive pixels of its channel row as one float4 -- k-step (t, e)
//     pairs pixel 8t + e (lane half 0) with pixel 8t + 4 + e (half 1), so a row's 32-byte piece is one request per lane and a 128-byte
//     line serves four of them (round 3 staged every tile of F through 66 KB of LDS for this; the k order of an MFMA is free).
// dF: D[c][pix] = A[c][hid] . B[hid][pix]; A = w_in^T[c][hid] from LDS, B = dG (lane l: pixel l&31, hidden parity l>>5), read as
//     float4 rows and consumed over 4 k-steps: k-step (q, e) pairs hidden 8q + e (half 0) with 8q + 4 + e (half 1).  D rows
//     (channels) in registers, columns (pixels) across lanes -> 128-byte NCHW row stores.
// Every load of a (half) tile is issued before its first MFMA (sched_barrier: the scheduler sinks loads to their uses otherwise; keeping
// the NEXT unit's loads in flight as well -- a register double buffer -- measured slower, profiles/r04g).  The weight gradient is reduced registers -> LDS -> one atomic per (hid, c).
// dW += dG^T F over one tile of 64 pixels (p0 ..): a tile = 8 groups of 8 pixels, taken as two halves of four groups: all loads of a
// half (a float4 of F per channel tile and four dG values per hidden tile and group: 64 registers) are issued before its first MFMA
template <int C, int HD>
__device__ __forceinline__ void project_dw_tile(const float* __restrict__ F, const float* __restrict__ dG, int p0, int HW, bool vec4, int h, int col,
                                                f32x16 (&accw)[HD / 32][C / 32], int rs = 0) {
  constexpr int HT = HD / 32, CT = C / 32;
  const bool full = vec4 && tile_px(p0, 63, rs) < HW;   // wave-uniform: whole tile inside the image, rows 16-byte aligned -> no per-load guards
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    float4 fb[4][CT];
    float av[4][4][HT];
    if (full) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int pix4 = tile_px_c(p0, 8 * (4 * half + t), 4 * h, rs);   // four consecutive texels (slots 4-aligned: inside one 16-texel row piece)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) fb[t][ct] = *reinterpret_cast<const float4*>(F + (unsigned)((ct * 32 + col) * HW + pix4));
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int ht = 0; ht < HT; ++ht) av[t][e][ht] = dG[(unsigned)((pix4 + e) * HD + ht * 32 + col)];   // stored channel ht*32 + col
      }
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int pix4 = tile_px_c(p0, 8 * (4 * half + t), 4 * h, rs);   // four consecutive texels (slots 4-aligned: inside one 16-texel row piece)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const unsigned ro = (unsigned)((ct * 32 + col) * HW + pix4);     // element offset inside the image's map (< 2^32)
          fb[t][ct] = make_float4(pix4 < HW ? F[ro] : 0.0f, pix4 + 1 < HW ? F[ro + 1] : 0.0f, pix4 + 2 < HW ? F[ro + 2] : 0.0f,
                                  pix4 + 3 < HW ? F[ro + 3] : 0.0f);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int pix = pix4 + e;
#pragma unroll
          for (int ht = 0; ht < HT; ++ht) av[t][e][ht] = pix < HW ? dG[(unsigned)(pix * HD + ht * 32 + col)] : 0.0f;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // all loads of the half are out before its first MFMA (the scheduler would sink them)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int ht = 0; ht < HT; ++ht)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            const float* fp = reinterpret_cast<const float*>(&fb[t][ct]);
            accw[ht][ct] = mfma(av[t][e][ht], fp[e], accw[ht][ct]);
          }
      }
  }
}

// dF = dG w_in[:, :C] over one tile of 64 pixels; wl[hid*C + c] = w_in[hid][c] in LDS
template <int C, int HD>
__device__ __forceinline__ void project_df_tile(const float* __restrict__ dG, float* __restrict__ dF, const float* wl, int p0, int HW, int h, int col, int rs = 0) {
  constexpr int CT = C / 32;
  const float4* dG4 = reinterpret_cast<const float4*>(dG);
  const int px0 = min(tile_px(p0, col, rs), HW - 1), px1 = min(tile_px(p0, 32 + col, rs), HW - 1);
  float4 v0[HD / 8], v1[HD / 8];
#pragma unroll
  for (int qq = 0; qq < HD / 8; ++qq) {
    // storage float4 (ht*8 + 4h + q) holds hidden ht*32 + 8q + 4h + e, e = 0..3  (proj_storage_index)
    const int ht = qq >> 2, q = qq & 3;
    v0[qq] = dG4[(unsigned)(px0 * (HD / 4) + ht * 8 + 4 * h + q)];
    v1[qq] = dG4[(unsigned)(px1 * (HD / 4) + ht * 8 + 4 * h + q)];
  }
  __builtin_amdgcn_sched_barrier(0);   // all 16 row loads are out before the first MFMA
  f32x16 acc[CT][2];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) acc[ct][pt] = zero_acc();
#pragma unroll
  for (int qq = 0; qq < HD / 8; ++qq) {
    const int ht = qq >> 2, q = qq & 3;
    const float* b0 = reinterpret_cast<const float*>(&v0[qq]);
    const float* b1 = reinterpret_cast<const float*>(&v1[qq]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int hid = ht * 32 + 8 * q + 4 * h + e;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const float a = wl[hid * C + ct * 32 + col];
        acc[ct][0] = mfma(a, b0[e], acc[ct][0]);
        acc[ct][1] = mfma(a, b1[e], acc[ct][1]);
      }
    }
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const int pix = tile_px(p0, pt * 32 + col, rs);
      if (pix < HW) {
#pragma unroll
        for (int q = 0; q < 16; ++q) dF[(unsigned)((ct * 32 + mfma_row(q, h)) * HW + pix)] = acc[ct][pt][q];
      }
    }
}

// the same, one half of the tile (32 pixels) after the other: half the registers, for the kernel whose waves hold the weight
// gradient's accumulators as well
template <int C, int HD>
__device__ __forceinline__ void project_df_half_tiles(const float* __restrict__ dG, float* __restrict__ dF, const float* wl, int p0, int HW, int h, int col,
                                                      int rs = 0) {
  constexpr int CT = C / 32;
  const float4* dG4 = reinterpret_cast<const float4*>(dG);
#pragma unroll 1
  for (int pt = 0; pt < 2; ++pt) {
    const int pix = tile_px(p0, pt * 32 + col, rs);
    const int px = min(pix, HW - 1);
    float4 v[HD / 8];
#pragma unroll
    for (int qq = 0; qq < HD / 8; ++qq) v[qq] = dG4[(unsigned)(px * (HD / 4) + (qq >> 2) * 8 + 4 * h + (qq & 3))];
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[ct] = zero_acc();
#pragma unroll
    for (int qq = 0; qq < HD / 8; ++qq) {
      const float* b = reinterpret_cast<const float*>(&v[qq]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int hid = (qq >> 2) * 32 + 8 * (qq & 3) + 4 * h + e;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct] = mfma(wl[hid * C + ct * 32 + col], b[e], acc[ct]);
      }
    }
    if (pix < HW) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int q = 0; q < 16; ++q) dF[(unsigned)((ct * 32 + mfma_row(q, h)) * HW + pix)] = acc[ct][q];
    }
  }
}

// ---- the two contractions of a tile with F and dF channels-last ((N, H, W, C) in memory):
//   dW: B[k = pix][j = c] = F[pix][c]: 32 lanes = 32 consecutive channels of one pixel, like the dG operand -- 128-byte pieces both
//   dF: D[pix][c] = A[pix][hid] . B[hid][c]: the roles of the NCHW form swapped -- dG rows are the A operand, w_in from LDS the B operand,
//       D rows (pixels) sit in registers, columns (channels) across the lanes: every store is a 128-byte piece of a pixel's channel vector
template <int C, int HD>
__device__ __forceinline__ void project_dw_tile_cl(const float* __restrict__ F, const float* __restrict__ dG, int p0, int HW, int h, int col,
                                                   f32x16 (&accw)[HD / 32][C / 32], int rs = 0) {
  constexpr int HT = HD / 32, CT = C / 32;
  const bool full = tile_px(p0, 63, rs) < HW;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    float fv[4][4][CT];
    float av[4][4][HT];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int pix4 = tile_px_c(p0, 8 * (4 * half + t), 4 * h, rs);   // four consecutive texels (slots 4-aligned: inside one 16-texel row piece)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int pix = pix4 + e;
        const bool in = full || pix < HW;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) fv[t][e][ct] = in ? F[(unsigned)(pix * C + ct * 32 + col)] : 0.0f;
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) av[t][e][ht] = in ? dG[(unsigned)(pix * HD + ht * 32 + col)] : 0.0f;
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // all loads of the half are out before its first MFMA
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int ht = 0; ht < HT; ++ht)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) accw[ht][ct] = mfma(av[t][e][ht], fv[t][e][ct], accw[ht][ct]);
  }
}
template <int C, int HD>
__device__ __forceinline__ void project_df_half_tiles_cl(const float* __restrict__ dG, float* __restrict__ dF, const float* wl, int p0, int HW, int h, int col,
                                                         int rs = 0) {
  constexpr int CT = C / 32;
  const float4* dG4 = reinterpret_cast<const float4*>(dG);
#pragma unroll 1
  for (int pt = 0; pt < 2; ++pt) {
    const int px = min(tile_px(p0, pt * 32 + col, rs), HW - 1);
    float4 v[HD / 8];
#pragma unroll
    for (int qq = 0; qq < HD / 8; ++qq) v[qq] = dG4[(unsigned)(px * (HD / 4) + (qq >> 2) * 8 + 4 * h + (qq & 3))];
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[ct] = zero_acc();
#pragma unroll
    for (int qq = 0; qq < HD / 8; ++qq) {
      const float* a = reinterpret_cast<const float*>(&v[qq]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int hid = (qq >> 2) * 32 + 8 * (qq & 3) + 4 * h + e;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct] = mfma(a[e], wl[hid * C + ct * 32 + col], acc[ct]);
      }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int pix = tile_px_c(p0, pt * 32 + mfma_row(q, 0), 4 * h, rs);
      if (pix < HW) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) dF[(unsigned)(pix * C + ct * 32 + col)] = acc[ct][q];
      }
    }
  }
}

// the weight gradient's way out: registers -> work-group LDS (red = HD x C floats, reused from the staged weights) -> one atomic per (hid, c)
template <int C, int HD>
__device__ __forceinline__ void project_dw_flush(float* red, const f32x16 (&accw)[HD / 32][C / 32], bool has_acc, float* __restrict__ d_mlp, int h, int col) {
  constexpr int HT = HD / 32, CT = C / 32, D_IN = C + kPeDim;
  __syncthreads();
  for (int i = threadIdx.x; i < HD * C; i += blockDim.x) red[i] = 0.0f;
  __syncthreads();
  if (has_acc) {   // accumulator rows are STORED channels: map back to hidden units
#pragma unroll
    for (int ht = 0; ht < HT; ++ht)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int q = 0; q < 16; ++q) atomicAdd(&red[proj_hidden_of_storage(ht * 32 + mfma_row(q, h)) * C + ct * 32 + col], accw[ht][ct][q]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < HD * C; i += blockDim.x) {
    const int hid = i / C, c = i % C;
#ifndef BTS_ABL_NOFLUSH
    if (red[i] != 0.0f) atomicAdd(&d_mlp[hid * D_IN + c], red[i]);
#else
    if (red[i] == 1.2345e-30f) d_mlp[hid * D_IN + c] = red[i];
#endif
  }
}

template <int C, int HD>
__global__ __launch_bounds__(256, 2) void project_bwd_kernel(const float* __restrict__ feat, const float* __restrict__ dproj, const float* __restrict__ mlp,
                                                          float* __restrict__ dfeat, float* __restrict__ d_mlp, int HW, int tiles_per_img, int n_tiles) {
  constexpr int HT = HD / 32, CT = C / 32;
  constexpr int D_IN = C + kPeDim;
  __shared__ float wl[HD * C];  // wl[hid*C + c] = w_in[hid][c]; reused as the weight gradient's reduction buffer after the tile loop
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, col = lane & 31;
  if (dfeat) {
    for (int i = threadIdx.x; i < HD * C; i += blockDim.x) {
      const int hid = i / C, c = i % C;
      wl[i] = mlp[hid * D_IN + c];
    }
  }
  __syncthreads();
  // roles: with both gradients wanted waves 0, 1 reduce dW and waves 2, 3 write dF, each pair walking ALL tiles of the work-group
  // (role-local wave index rw of nr); with one gradient wanted all four waves take that role
  const bool both = dfeat && d_mlp;
  const bool role_w = d_mlp && (!both || wave < 2);
  const int nr = both ? 2 : 4, rw = both ? (wave & 1) : wave;
  const bool vec4 = (HW & 3) == 0;   // float4 reads of F rows need 16-byte aligned rows
  f32x16 accw[HT][CT];
#pragma unroll
  for (int ht = 0; ht < HT; ++ht)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) accw[ht][ct] = zero_acc();

  for (int tile = blockIdx.x * nr + rw; tile < n_tiles; tile += gridDim.x * nr) {
    const int img = tile / tiles_per_img;
    const int p0 = (tile - img * tiles_per_img) * 64;
    const float* dG = dproj + (long)img * HW * HD;
    if (role_w) project_dw_tile<C, HD>(feat + (long)img * C * HW, dG, p0, HW, vec4, h, col, accw);
    else project_df_tile<C, HD>(dG, dfeat + (long)img * C * HW, wl, p0, HW, h, col);
  }
  if (d_mlp) project_dw_flush<C, HD>(wl, accw, role_w, d_mlp, h, col);
}

// ---- the same backward over a SPARSE dG: `tiles` holds one byte per tile of 64 pixels, set by bts_render_bwd's scatter pass for every
// tile it added into (BtsRenderGrads.d_proj_tiles); everything else of dproj is zero by contract and is never read.  A training step
// renders a few thousand 8 x 8 patches: 8-15 % of the tiles (exp_re10k.yaml / exp_kitti_360.yaml's batches), so the pass costs the
// dense write of dF plus a tenth of the reads.  Every wave takes whole tiles: a dirty one gets both contractions (the second read of
// its dG rows comes from L1 / L2), a clean one 16 wide zero stores.  With `clear` the wave then writes zeros over the dirty tile and
// resets its byte -- the (dproj, tiles) pair is all zero again when the kernel ends, and the caller never fills 503 MB before a step.
// FEAT_CL: F and dF channels-last (see project_kernel).  tiles == NULL: every tile counts as flagged and nothing is cleared (the dense
// backward of a channels-last map).
// LIST (bts_train_step_bwd, large maps): the flagged tiles come as a compact list (compact_tiles_kernel below) and are dealt round-robin
// over the waves -- every wave gets the same number of dirty tiles to within one.  With the flags read in place a wave's share of the
// dirty tiles is binomial: at exp_kitti_raw.yaml's batch 1.15 on average and 5 for the unluckiest of the 2 048 waves, and the kernel lasts
// as long as that one.  The clean tiles (zero rows of dF) follow in a second round-robin loop over `copy`, the compaction's copy of the
// flags -- the flags themselves are already back to zero.
template <int C, int HD, bool FEAT_CL, bool LIST = false>
__global__ __launch_bounds__(256, 2) void project_bwd_tiles_kernel(const float* __restrict__ feat, float* dproj, unsigned char* tiles, const float* __restrict__ mlp,
                                                                float* __restrict__ dfeat, float* __restrict__ d_mlp, int HW, int tiles_per_img, int n_tiles,
                                                                int clear, int Wm, int tw, const int* __restrict__ list = nullptr,
                                                                const unsigned char* __restrict__ copy = nullptr) {
  constexpr int HT = HD / 32, CT = C / 32;
  constexpr int D_IN = C + kPeDim;
  __shared__ float wl[HD * C];
  const int rs = tile_rs(Wm, tw);   // slot i of a tile is texel p0 + i + (i >> 4) * rs (bts_common.h: a 16 x 4 block, or rs = 0: 64 consecutive texels)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, col = lane & 31;
  if (dfeat) {
    for (int i = threadIdx.x; i < HD * C; i += blockDim.x) {
      const int hid = i / C, c = i % C;
      wl[i] = mlp[hid * D_IN + c];
    }
  }
  __syncthreads();
  const bool vec4 = (HW & 3) == 0;
  f32x16 accw[HT][CT];
#pragma unroll
  for (int ht = 0; ht < HT; ++ht)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) accw[ht][ct] = zero_acc();
  const float4 z4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  const int n_runs = rs ? 4 : 1, run_step = rs ? Wm : 0;   // a tile as runs of consecutive texels: ONE run of (up to) 64 in the linear form, four rows of 16 in the block form

  auto do_dirty = [&](int tile, bool reset_flag) {
    const int img = tile / tiles_per_img;
    const int p0 = tile_base(tile - img * tiles_per_img, Wm, tw);
    const int run_px = rs ? 16 : min(64, HW - p0);
    float* dG = dproj + (long)img * HW * HD;
    if constexpr (FEAT_CL) {
      if (d_mlp) project_dw_tile_cl<C, HD>(feat + (long)img * C * HW, dG, p0, HW, h, col, accw, rs);
      if (dfeat) project_df_half_tiles_cl<C, HD>(dG, dfeat + (long)img * C * HW, wl, p0, HW, h, col, rs);
    } else {
      if (d_mlp) project_dw_tile<C, HD>(feat + (long)img * C * HW, dG, p0, HW, vec4, h, col, accw, rs);
      if (dfeat) project_df_half_tiles<C, HD>(dG, dfeat + (long)img * C * HW, wl, p0, HW, h, col, rs);
    }
    if (LIST || (clear && tiles)) {
      // (this wave is the only reader of the tile and its loads have returned: their values went through the MFMAs above)
      __builtin_amdgcn_sched_barrier(0);
      for (int r = 0; r < n_runs; ++r) {
        float4* row = reinterpret_cast<float4*>(dG + (long)(p0 + r * run_step) * HD);   // run_px texels x HD floats, contiguous
        for (int i = lane; i < run_px * (HD / 4); i += 64) row[i] = z4;
      }
      if (reset_flag && lane == 0) tiles[tile] = 0;
    }
  };
  auto do_clean = [&](int tile) {
    const int img = tile / tiles_per_img;
    const int p0 = tile_base(tile - img * tiles_per_img, Wm, tw);
    const int run_px = rs ? 16 : min(64, HW - p0);
    float* dF = dfeat + (long)img * C * HW;
    if constexpr (FEAT_CL) {       // a run's texels x C floats are contiguous
      for (int r = 0; r < n_runs; ++r) {
        float4* row = reinterpret_cast<float4*>(dF + (long)(p0 + r * run_step) * C);
        for (int i = lane; i < run_px * (C / 4); i += 64) row[i] = z4;
      }
    } else
    if (vec4 && tile_px(p0, 63, rs) < HW) {   // 16 stores of 4 channel rows x 64 texels (lane & 15: four texels of slot group 4 (lane & 15))
      const int px4 = tile_px(p0, 4 * (lane & 15), rs);
#pragma unroll
      for (int i = 0; i < C / 4; ++i) *reinterpret_cast<float4*>(dF + (unsigned)((4 * i + (lane >> 4)) * HW + px4)) = z4;
    } else if (tile_px(p0, lane, rs) < HW) {
      const int px = tile_px(p0, lane, rs);
#pragma unroll 8
      for (int c = 0; c < C; ++c) dF[(unsigned)(c * HW + px)] = 0.0f;
    }
  };

  const int tile0 = blockIdx.x * 4 + wave, stride = gridDim.x * 4;
  if constexpr (LIST) {
    const int n_dirty = min(n_tiles, list[0]);
    int id_n = tile0 < n_dirty ? list[4 + tile0] : 0;               // an entry is fetched one tile ahead
    for (int j = tile0; j < n_dirty; j += stride) {
      const int tile = __builtin_amdgcn_readfirstlane(id_n);
      id_n = j + stride < n_dirty ? list[4 + j + stride] : 0;
      do_dirty(tile, false);
    }
    if (dfeat) {
      int flag_n = tile0 < n_tiles ? (int)copy[tile0] : 1;
      for (int tile = tile0; tile < n_tiles; tile += stride) {
        const bool was_dirty = __builtin_amdgcn_readfirstlane(flag_n) != 0;
        flag_n = tile + stride < n_tiles ? (int)copy[tile + stride] : 1;
        if (!was_dirty) do_clean(tile);
      }
    }
  } else {
    int flag_n = !tiles ? 1 : (tile0 < n_tiles ? (int)tiles[tile0] : 0);   // a tile's flag is fetched one tile ahead
    for (int tile = tile0; tile < n_tiles; tile += stride) {
      const bool dirty = __builtin_amdgcn_readfirstlane(flag_n) != 0;
      flag_n = !tiles ? 1 : (tile + stride < n_tiles ? (int)tiles[tile + stride] : 0);
      if (dirty) do_dirty(tile, true);
      else if (dfeat) do_clean(tile);
    }
  }
  if (d_mlp) project_dw_flush<C, HD>(wl, accw, true, d_mlp, h, col);
}

// The compaction in front of the LIST form: list[0] (zeroed by the launcher) counts the flagged tiles, list[4 ..] holds their indices (in
// no particular order: one atomic per work-group reserves its stretch), copy[t] keeps the flag of tile t for the kernel's clean pass, and
// the flags themselves go back to zero -- the (d_proj, tiles) pair is all zero again once the LIST kernel has run.  One thread per tile.
// copy == NULL (the forward's sampled-tile flags): the flags are only read.
__global__ __launch_bounds__(256) void compact_tiles_kernel(unsigned char* __restrict__ tiles, int n_tiles, int* __restrict__ list,
                                                          unsigned char* __restrict__ copy) {
  __shared__ int wave_sum[4], base;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool on = t < n_tiles && tiles[t] != 0;
  if (t < n_tiles && copy) {
    copy[t] = on ? 1 : 0;
    if (on) tiles[t] = 0;
  }
  const unsigned long long m = __ballot(on);
  const int before = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) wave_sum[wave] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int total = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
    base = total ? atomicAdd(&list[0], total) : 0;
  }
  __syncthreads();
  int off = base + before;
  for (int w = 0; w < wave; ++w) off += wave_sum[w];
  if (on) list[4 + off] = t;
}

constexpr long kListMinTiles = 4096;
// (A/B switch of the list-driven forms: the product reads no environment variables -- probe build only, as every other switch)
static bool tile_list_on() {
#ifdef BTS_PROBE
  return getenv("BTS_NO_TILE_LIST") == nullptr;
#else
  return true;
#endif
}
// workspace of the LIST form for a map of n_tiles tiles: the count (16 bytes), the indices, the copy of the flags
size_t project_bwd_list_bytes(long n_tiles) { return (size_t)(4 + n_tiles) * sizeof(int) + (size_t)n_tiles + 16; }

static int prep_cus() {
  static thread_local int cus[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
  if (!cus[dev]) {
    hipDeviceProp_t prop;
    cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return cus[dev];
}

template <int C, int HD>
static int run_fwd(const float* feat, const float* mlp, int N, int HW, float* proj, const unsigned char* tiles, bool feat_cl, hipStream_t s, int Wm,
                   void* list_ws, size_t list_ws_bytes) {
  const int tpi = (HW + 63) / 64;
  const int tw = Wm > 0 && HW % Wm == 0 ? tile_cols(HW / Wm, Wm, 1) : 0;   // (Wm = 0: runs of 64 consecutive texels)
  const long n_tiles = (long)N * tpi;
  const long want = (n_tiles + 3) / 4, cap = 4L * prep_cus();     // persistent: <= 4 work-groups per CU (16 KB of LDS each)
  const int grid = (int)(want < cap ? want : cap);
  if (list_ws && tiles && n_tiles >= kListMinTiles && list_ws_bytes >= project_bwd_list_bytes(n_tiles) && tile_list_on()) {
    int* list = static_cast<int*>(list_ws);    // the balanced form (see project_bwd_tiles_kernel): the flags stay as they are
    if (hipMemsetAsync(list, 0, 16, s) != hipSuccess) return BTS_E_LAUNCH;
    compact_tiles_kernel<<<(int)((n_tiles + 255) / 256), 256, 0, s>>>(const_cast<unsigned char*>(tiles), (int)n_tiles, list, nullptr);
    if (feat_cl) project_kernel<C, HD, true><<<grid, 256, 0, s>>>(feat, mlp, proj, HW, tpi, (int)n_tiles, tiles, Wm, tw, list);
    else project_kernel<C, HD, false><<<grid, 256, 0, s>>>(feat, mlp, proj, HW, tpi, (int)n_tiles, tiles, Wm, tw, list);
    return hipGetLastError() == hipSuccess ? BTS_OK : BTS_E_LAUNCH;
  }
  if (feat_cl) project_kernel<C, HD, true><<<grid, 256, 0, s>>>(feat, mlp, proj, HW, tpi, (int)n_tiles, tiles, Wm, tw);
  else project_kernel<C, HD, false><<<grid, 256, 0, s>>>(feat, mlp, proj, HW, tpi, (int)n_tiles, tiles, Wm, tw);
  return hipGetLastError() == hipSuccess ? BTS_OK : BTS_E_LAUNCH;
}

template <int C, int HD>
static int run_bwd(const float* feat, const float* dproj, const float* mlp, int N, int HW, float* dfeat, float* d_mlp, hipStream_t s) {
  if (!dfeat && !d_mlp) return BTS_OK;
  const int tiles = (HW + 63) / 64;
  const long n_tiles = (long)N * tiles;
  const int nr = (dfeat && d_mlp) ? 2 : 4;
  const long want = (n_tiles + nr - 1) / nr, cap = 2L * prep_cus();   // <= 2 work-groups per CU: the roles hold up to 256 VGPRs
  project_bwd_kernel<C, HD><<<(int)(want < cap ? want : cap), 256, 0, s>>>(feat, dproj, mlp, dfeat, d_mlp, HW, tiles, (int)n_tiles);
  return hipGetLastError() == hipSuccess ? BTS_OK : BTS_E_LAUNCH;
}

template <int C, int HD>
static int run_bwd_tiles(const float* feat, float* dproj, unsigned char* tiles, const float* mlp, int N, int HW, float* dfeat, float* d_mlp, int clear,
                         bool feat_cl, hipStream_t s, int Wm, void* list_ws, size_t list_ws_bytes) {
  if (!dfeat && !d_mlp && !(clear && tiles)) return BTS_OK;
  const int tpi = (HW + 63) / 64;
  const int tw = Wm > 0 && HW % Wm == 0 ? tile_cols(HW / Wm, Wm, 1) : 0;
  const long n_tiles = (long)N * tpi;
  const long want = (n_tiles + 3) / 4, cap = 2L * prep_cus();
  const int grid = (int)(want < cap ? want : cap);
  // the balanced form: worth its two small extra launches on maps of a few thousand tiles and more (below that a wave holds a tile or two)
  if (list_ws && tiles && clear && n_tiles >= kListMinTiles && list_ws_bytes >= project_bwd_list_bytes(n_tiles) && tile_list_on()) {
    int* list = static_cast<int*>(list_ws);
    unsigned char* copy = reinterpret_cast<unsigned char*>(list + 4 + n_tiles);
    if (hipMemsetAsync(list, 0, 16, s) != hipSuccess) return BTS_E_LAUNCH;
    compact_tiles_kernel<<<(int)((n_tiles + 255) / 256), 256, 0, s>>>(tiles, (int)n_tiles, list, copy);
    if (feat_cl) project_bwd_tiles_kernel<C, HD, true, true><<<grid, 256, 0, s>>>(feat, dproj, tiles, mlp, dfeat, d_mlp, HW, tpi, (int)n_tiles, 1, Wm, tw, list, copy);
    else project_bwd_tiles_kernel<C, HD, false, true><<<grid, 256, 0, s>>>(feat, dproj, tiles, mlp, dfeat, d_mlp, HW, tpi, (int)n_tiles, 1, Wm, tw, list, copy);
    return hipGetLastError() == hipSuccess ? BTS_OK : BTS_E_LAUNCH;
  }
  if (feat_cl) project_bwd_tiles_kernel<C, HD, true><<<grid, 256, 0, s>>>(feat, dproj, tiles, mlp, dfeat, d_mlp, HW, tpi, (int)n_tiles, clear, Wm, tw);
  else project_bwd_tiles_kernel<C, HD, false><<<grid, 256, 0, s>>>(feat, dproj, tiles, mlp, dfeat, d_mlp, HW, tpi, (int)n_tiles, clear, Wm, tw);
  return hipGetLastError() == hipSuccess ? BTS_OK : BTS_E_LAUNCH;
}

int project_features_impl(int C, int HD, const float* feat, const float* mlp, int N, int HW, float* proj, const unsigned char* tiles, hipStream_t s,
                          bool feat_cl, int Wm, void* list_ws, size_t list_ws_bytes) {
  if (C == 64 && HD == 64) return run_fwd<64, 64>(feat, mlp, N, HW, proj, tiles, feat_cl, s, Wm, list_ws, list_ws_bytes);
  if (C == 32 && HD == 32) return run_fwd<32, 32>(feat, mlp, N, HW, proj, tiles, feat_cl, s, Wm, list_ws, list_ws_bytes);
  return BTS_E_UNSUPPORTED;
}

// ---- which tiles of the projected map will a render read?  One thread per sample: the render kernels' own depth (coarse_depth), point
// (o + z d: mul, then add), projection and tap routine -- the same texels bit for bit -- and a byte store per tap into the tile flags.
__global__ __launch_bounds__(256) void mark_tiles_kernel(const float* __restrict__ rays, const float* __restrict__ z_samp, const float* __restrict__ jitter,
                                                       const float* __restrict__ w2c_enc, const float* __restrict__ K_enc, long B, int Bp, int K,
                                                       int lindisp, int H, int W, int fs, int tiles_per_img, unsigned char* __restrict__ tiles, int tw) {
  // lane = sample, one ray per wave iteration (ray, camera: wave-uniform scalar loads), as the render kernels walk them
  const int lane = threadIdx.x & 63;
  const long wave = blockIdx.x * 4L + (threadIdx.x >> 6), n_waves = gridDim.x * 4L;
  const float step = 1.0f / (float)K;
  const int Wm = W >> fs;   // (tw: the map's tile geometry, bts_common.h)
  for (long b = wave; b < B; b += n_waves) {
    const int sample = (int)(b / Bp);
    const cfp rp = as_const(rays) + b * 8;
    const float ox = rp[0], oy = rp[1], oz = rp[2], dx = rp[3], dy = rp[4], dz = rp[5], near = rp[6], far = rp[7];
    const Cam enc = load_cam(w2c_enc + sample * 16, K_enc + sample * 9);
    unsigned char* t = tiles + (long)sample * tiles_per_img;
    for (int k = lane; k < K; k += 64) {
      const float z = z_samp ? z_samp[b * K + k] : coarse_depth(jitter[b * K + k], coarse_base(K, k), step, near, far, lindisp != 0);
      const float px = ox + z * dx, py = oy + z * dy, pz = oz + z * dz;
      const Proj pe = project<false>(enc, px, py, pz);
      int x0, y0, x1, y1;
      (void)make_taps_xy(pe.x, pe.y, H, W, x0, y0, x1, y1, fs);      // (x0 .. y1: texels of the map in memory, H >> fs x W >> fs)
      // a sample's two taps of a row share their tile but for a tile border; neighbouring samples mostly share both: one store per
      // distinct tile and lane run instead of four per sample
      const unsigned ta = tile_of(y0, x0, Wm, tw), tb = tile_of(y0, x1, Wm, tw), tc = tile_of(y1, x0, Wm, tw), td = tile_of(y1, x1, Wm, tw);
      const unsigned pa = (unsigned)__builtin_amdgcn_update_dpp((int)~0u, (int)ta, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
      const unsigned pc = (unsigned)__builtin_amdgcn_update_dpp((int)~0u, (int)tc, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
      if (ta != pa) t[ta] = 1;
      if (tb != ta) t[tb] = 1;
      if (tc != pc) t[tc] = 1;
      if (td != tc) t[td] = 1;
    }
  }
}

int mark_tiles_impl(const float* rays, const float* z_samp, const float* jitter, const float* w2c_enc, const float* K_enc, long B, int Bp, int K, int lindisp,
                    int H, int W, int fs, unsigned char* tiles, hipStream_t s, int blocks) {
  const long total = B * K;
  const int tpi = (int)((((long)(H >> fs) * (W >> fs)) + 63) / 64);
  const long want = (B + 3) / 4;   // one ray per wave iteration
  (void)total;
  mark_tiles_kernel<<<(int)(want < 8192 ? want : 8192), 256, 0, s>>>(rays, z_samp, jitter, w2c_enc, K_enc, B, Bp, K, lindisp, H, W, fs, tpi, tiles,
                                                                     tile_cols(H >> fs, W >> fs, blocks));
  return hipGetLastError() == hipSuccess ? BTS_OK : BTS_E_LAUNCH;
}

int project_features_bwd_impl(int C, int HD, const float* feat, const float* dproj, const float* mlp, int N, int HW, float* dfeat,
                              float* d_mlp, hipStream_t s) {
  if (C == 64 && HD == 64) return run_bwd<64, 64>(feat, dproj, mlp, N, HW, dfeat, d_mlp, s);
  if (C == 32 && HD == 32) return run_bwd<32, 32>(feat, dproj, mlp, N, HW, dfeat, d_mlp, s);
  return BTS_E_UNSUPPORTED;
}

int project_features_bwd_tiles_impl(int C, int HD, const float* feat, float* dproj, unsigned char* tiles, const float* mlp, int N, int HW, float* dfeat,
                                    float* d_mlp, int clear, hipStream_t s, bool feat_cl, int Wm, void* list_ws, size_t list_ws_bytes) {
  if (C == 64 && HD == 64) return run_bwd_tiles<64, 64>(feat, dproj, tiles, mlp, N, HW, dfeat, d_mlp, clear, feat_cl, s, Wm, list_ws, list_ws_bytes);
  if (C == 32 && HD == 32) return run_bwd_tiles<32, 32>(feat, dproj, tiles, mlp, N, HW, dfeat, d_mlp, clear, feat_cl, s, Wm, list_ws, list_ws_bytes);
  return BTS_E_UNSUPPORTED;
}

}  // namespace bts
