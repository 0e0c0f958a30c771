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

namespace bts {

void set_error(const char* fmt, const char* a = "", long b = 0, long c = 0, long d = 0);

// ---- forward: one wave = 64 pixels x all Hd outputs.  D[pix][hid] = A[pix][c] . B[c][hid]:
// A operand = F (lane l: pixel l&31, channel parity l>>5) read straight from the NCHW rows, B = w_in^T from LDS (k-major),
// D rows (pixels) sit in registers, columns (hidden) across lanes -> every store is a 128-byte row segment of G.
template <int C, int HD>
__global__ __launch_bounds__(256) void project_kernel(const float* __restrict__ feat, const float* __restrict__ mlp, float* __restrict__ proj,
                                                      int HW, int tiles_per_img) {
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
  const int wg = blockIdx.x;
  const int img = wg / tiles_per_img;
  const int p0 = (wg - img * tiles_per_img) * 256 + wave * 64;
  if (p0 >= HW) return;
  const float* F = feat + (long)img * C * HW;
  float* G = proj + (long)img * HW * HD;

  f32x16 acc[2][HT];
#pragma unroll
  for (int pt = 0; pt < 2; ++pt)
#pragma unroll
    for (int ht = 0; ht < HT; ++ht) acc[pt][ht] = zero_acc();
  const int px0 = min(p0 + col, HW - 1), px1 = min(p0 + 32 + col, HW - 1);
#pragma unroll 4
  for (int s = 0; s < C / 2; ++s) {
    const int c = 2 * s + h;
    const float a0 = F[(long)c * HW + px0];
    const float a1 = F[(long)c * HW + px1];
#pragma unroll
    for (int ht = 0; ht < HT; ++ht) {
      const float b = wl[c * HD + ht * 32 + col];
      acc[0][ht] = mfma(a0, b, acc[0][ht]);
      acc[1][ht] = mfma(a1, b, acc[1][ht]);
    }
  }
#pragma unroll
  for (int pt = 0; pt < 2; ++pt)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int pix = p0 + pt * 32 + mfma_row(q, h);
      if (pix < HW) {
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) G[(long)pix * HD + ht * 32 + col] = acc[pt][ht][q];
      }
    }
}

// ---- backward, feature gradient: D[c][pix] = A[c][hid] . B[hid][pix]; A = w_in^T[c][hid] from LDS, B = dG (lane l: pixel
// l&31, hidden parity l>>5).  D rows (channels) in registers, columns (pixels) across lanes -> 128-byte NCHW row stores.
// dG rows are read as float4 (4 consecutive hidden units per lane half) and consumed over 4 k-steps:
// k-step (q, e) pairs hidden 8q + e (half 0) with 8q + 4 + e (half 1).
template <int C, int HD>
__global__ __launch_bounds__(256) void project_bwd_feat_kernel(const float* __restrict__ dproj, const float* __restrict__ mlp,
                                                               float* __restrict__ dfeat, int HW, int tiles_per_img) {
  constexpr int CT = C / 32;
  constexpr int D_IN = C + kPeDim;
  __shared__ float wl[HD * C];  // wl[hid*C + c] = w_in[hid][c]
  for (int i = threadIdx.x; i < HD * C; i += blockDim.x) {
    const int hid = i / C, c = i % C;
    wl[i] = mlp[hid * D_IN + c];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, col = lane & 31;
  const int wg = blockIdx.x;
  const int img = wg / tiles_per_img;
  const int p0 = (wg - img * tiles_per_img) * 256 + wave * 64;
  if (p0 >= HW) return;
  const float4* dG = reinterpret_cast<const float4*>(dproj + (long)img * HW * HD);
  float* dF = dfeat + (long)img * C * HW;
  f32x16 acc[CT][2];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) acc[ct][pt] = zero_acc();
  const int px0 = min(p0 + col, HW - 1), px1 = min(p0 + 32 + col, HW - 1);
#pragma unroll 2
  for (int qq = 0; qq < HD / 8; ++qq) {
    // storage float4 (ht*8 + 4h + q) holds hidden ht*32 + 8q + 4h + e, e = 0..3  (proj_storage_index)
    const int ht = qq >> 2, q = qq & 3;
    const float4 v0 = dG[(long)px0 * (HD / 4) + ht * 8 + 4 * h + q];
    const float4 v1 = dG[(long)px1 * (HD / 4) + ht * 8 + 4 * h + q];
    const float* b0 = reinterpret_cast<const float*>(&v0);
    const float* b1 = reinterpret_cast<const float*>(&v1);
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
      const int pix = p0 + pt * 32 + col;
      if (pix < HW) {
#pragma unroll
        for (int q = 0; q < 16; ++q) dF[(long)(ct * 32 + mfma_row(q, h)) * HW + pix] = acc[ct][pt][q];
      }
    }
}

// ---- backward, weight gradient: dW[hid][c] = sum_pix dG[pix][hid] * F[c][pix]  (contraction over pixels).
// A[i = stored channel s][k = pix] comes straight from dG (32 lanes = 32 consecutive channels of one pixel: a 128-byte row segment);
// B[k = pix][j = c] = F[c][pix] would be a 4-byte gather with stride H*W from the NCHW map, so every wave first stages its 64-pixel
// tile of F through LDS: coalesced 256-byte rows in (lane = pixel), [c][pix] with an odd leading dimension out (conflict-free).
// k-step s of a tile pairs pixel s (lane half 0) with pixel s + 32 (half 1).  Each work-group reduces a slab of pixels into
// registers, then LDS, then one atomic per (hid, c) into d_mlp.  HBM-bound: reads 4*(C + Hd) bytes per pixel once.
template <int C, int HD>
__global__ __launch_bounds__(256) void project_bwd_weight_kernel(const float* __restrict__ feat, const float* __restrict__ dproj,
                                                                 float* __restrict__ d_mlp, int HW, int slabs_per_img, int tiles_per_slab) {
  constexpr int HT = HD / 32, CT = C / 32;
  constexpr int D_IN = C + kPeDim;
  constexpr int LDF = 65;
  static_assert(4 * C * LDF >= HD * C, "the reduction buffer aliases the staging tiles");
  __shared__ float ftile[4 * C * LDF];
  float* red = ftile;                            // reused after the pixel loop
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, col = lane & 31;
  const int img = blockIdx.x / slabs_per_img;
  const int slab = blockIdx.x - img * slabs_per_img;
  const float* F = feat + (long)img * C * HW;
  const float* dG = dproj + (long)img * HW * HD;
  float* ft = ftile + wave * C * LDF;
  f32x16 acc[HT][CT];
#pragma unroll
  for (int ht = 0; ht < HT; ++ht)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[ht][ct] = zero_acc();
  for (int t = wave; t < tiles_per_slab; t += 4) {
    const int p0 = (slab * tiles_per_slab + t) * 64;
    if (p0 >= HW) break;
    // every global load of the tile is issued before the first use (the wave is alone on its SIMD half the time)
    const int px = min(p0 + lane, HW - 1);
    const bool okl = p0 + lane < HW;
    float fv[C], av[32][HT];
#pragma unroll
    for (int c = 0; c < C; ++c) fv[c] = F[(long)c * HW + px];   // rows of F: 256-byte coalesced reads
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const int pix = min(p0 + s + 32 * h, HW - 1);
#pragma unroll
      for (int ht = 0; ht < HT; ++ht) av[s][ht] = dG[(long)pix * HD + ht * 32 + col];   // stored channel ht*32 + col
    }
#pragma unroll
    for (int c = 0; c < C; ++c) ft[c * LDF + lane] = okl ? fv[c] : 0.0f;
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const int pl = s + 32 * h;                 // pixel of this lane half inside the tile
      const bool ok = p0 + pl < HW;
      float b[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) b[ct] = ft[(ct * 32 + col) * LDF + pl];
#pragma unroll
      for (int ht = 0; ht < HT; ++ht)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ht][ct] = mfma(ok ? av[s][ht] : 0.0f, b[ct], acc[ht][ct]);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < HD * C; i += blockDim.x) red[i] = 0.0f;
  __syncthreads();
  // accumulator rows are STORED channels: map back to hidden units
#pragma unroll
  for (int ht = 0; ht < HT; ++ht)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int q = 0; q < 16; ++q) atomicAdd(&red[proj_hidden_of_storage(ht * 32 + mfma_row(q, h)) * C + ct * 32 + col], acc[ht][ct][q]);
  __syncthreads();
  for (int i = threadIdx.x; i < HD * C; i += blockDim.x) {
    const int hid = i / C, c = i % C;
    atomicAdd(&d_mlp[hid * D_IN + c], red[i]);
  }
}

template <int C, int HD>
static int run_fwd(const float* feat, const float* mlp, int N, int HW, float* proj, hipStream_t s) {
  const int tiles = (HW + 255) / 256;
  project_kernel<C, HD><<<N * tiles, 256, 0, s>>>(feat, mlp, proj, HW, tiles);
  return hipGetLastError() == hipSuccess ? BTS_OK : BTS_E_LAUNCH;
}

template <int C, int HD>
static int run_bwd(const float* feat, const float* dproj, const float* mlp, int N, int HW, float* dfeat, float* d_mlp, hipStream_t s) {
  if (dfeat) {
    const int tiles = (HW + 255) / 256;
    project_bwd_feat_kernel<C, HD><<<N * tiles, 256, 0, s>>>(dproj, mlp, dfeat, HW, tiles);
    if (hipGetLastError() != hipSuccess) return BTS_E_LAUNCH;
  }
  if (d_mlp) {
    const int tiles64 = (HW + 63) / 64;
    const int tiles_per_slab = 32;  // 2048 pixels per work-group
    const int slabs = (tiles64 + tiles_per_slab - 1) / tiles_per_slab;
    project_bwd_weight_kernel<C, HD><<<N * slabs, 256, 0, s>>>(feat, dproj, d_mlp, HW, slabs, tiles_per_slab);
    if (hipGetLastError() != hipSuccess) return BTS_E_LAUNCH;
  }
  return BTS_OK;
}

int project_features_impl(int C, int HD, const float* feat, const float* mlp, int N, int HW, float* proj, hipStream_t s) {
  if (C == 64 && HD == 64) return run_fwd<64, 64>(feat, mlp, N, HW, proj, s);
  if (C == 32 && HD == 32) return run_fwd<32, 32>(feat, mlp, N, HW, proj, s);
  return BTS_E_UNSUPPORTED;
}

int project_features_bwd_impl(int C, int HD, const float* feat, const float* dproj, const float* mlp, int N, int HW, float* dfeat,
                              float* d_mlp, hipStream_t s) {
  if (C == 64 && HD == 64) return run_bwd<64, 64>(feat, dproj, mlp, N, HW, dfeat, d_mlp, s);
  if (C == 32 && HD == 32) return run_bwd<32, 32>(feat, dproj, mlp, N, HW, dfeat, d_mlp, s);
  return BTS_E_UNSUPPORTED;
}

}  // namespace bts
