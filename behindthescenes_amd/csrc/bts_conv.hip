// bts_conv.hip -- the Monodepth2 decoder's tail as hand-written CDNA4 kernels.
//
// With d_out = 64 the decoder's widths are clamped to >= 64 (models/common/backbones/monodepth2.py:189-206), so the layers that produce
// the renderer's scale-0 feature map are   upconv(0,0): ConvBlock 64 -> 64 @ H/2   ->  nearest x2  ->  upconv(0,1): ConvBlock 64 -> 64
// @ H  ->  dispconv(0): Conv3x3 64 -> 64 @ H   (monodepth2.py:211-239; ConvBlock = Conv3x3 + ELU, Conv3x3 = ReflectionPad2d(1) +
// 3 x 3 convolution, models/common/model/layers.py:11-40).  Through MIOpen the three cost 23.1 ms of a 47.5 ms exp_kitti_360.yaml
// step (profiles/r05c/md2_tail_probe.txt) -- and only 4.1 ms per full-size layer of that are the convolution kernels themselves
// (fp32 implicit GEMMs at 100-120 TFLOP/s: forward 1.18, data gradient 1.52, weight gradient 1.45 ms); the other 6 ms are the padded
// copy the reflection makes, layout changes around the channels-last kernels, ELU and its backward, the x2 upsampling and its backward
// as separate passes over 0.5 GB tensors.
//
// One operator, three kernels, nothing materialised in between:
//   y = [ELU]( conv3x3( reflect_pad1( [nearest x2]( x ) ), W ) + b ),   x (N, Hs, Ws, C) channels-last, y channels-last or NCHW
//   conv_fwd_bf_kernel    : implicit GEMM over the 9 taps; the reflection and the x2 upsampling are index arithmetic on the source pixel
//   conv_dgrad_bf_kernel  : the exact adjoint -- for an input pixel p the (output pixel q, tap t) pairs with reflect(q + t) = p: three
//                           per axis, a fourth for the two lines next to a border (the reflected line folds back onto them); with
//                           the x2 upsampling the wave accumulates the 2 x 2 children of a source pixel before it stores
//   conv_wgrad_bf_kernel  : dW[t] = sum_pixels x[reflect(p + t)] (x) dy[p], contraction over the pixels as the MFMA's k axis
// The first version of this file ran all three on v_mfma_f32_32x32x2_f32 (fp32 inputs, 157 TFLOP/s; kept as the -DBTS_CONV_FP32 A/B build,
// bts_conv_fp32.h) and won by the traffic and the launches that disappeared.  This one also leaves the fp32 matrix rate behind: every fp32
// operand is the EXACT sum of three bf16 numbers, six bf16 products reproduce the fp32 product to 2^-24 (see below), and the bf16 pipe
// is 16 x the fp32 one per instruction -- 0.375 of the matrix time with better sums than before (blocked accumulation).  bf16, not f16:
// activations and gradients have no bounded range to scale an f16 split on, and bf16 carries fp32's exponent.
#include "bts_bf16x3.h"

#include <cstring>

namespace bts {

void set_error(const char* fmt, const char* a = "", long b = 0, long c = 0, long d = 0);
int device_cu_count();
int transpose_launch(const float* src, float* dst, int N, int C, int H, int W, bool to_nhwc, hipStream_t s);

constexpr int kTapFloats = 64 * 64;          // one tap's weights

struct ConvParams {
  const float* x;     // source (N, Hs, Ws, 64) channels-last: the layer's input (FWD, WGRAD) or the output gradient dy' (DGRAD)
  const float* w;     // (64, 64, 3, 3) torch layout [co][ci][ty][tx]
  const float* bias;  // (64) or null (FWD)
  float* y;           // FWD: output (N, H, W, 64) or (N, 64, H, W); DGRAD: dx (N, Hs, Ws, 64)
  int N, H, W;        // the convolution's domain (output size of the layer = size of its padded-and-upsampled input)
  int up2;            // the layer's input is (N, H/2, W/2, 64), read through a nearest x2 upsampling
  int elu;            // FWD: ELU(alpha = 1) on the output
  int out_nchw;       // FWD: y is (N, 64, H, W)
  int tiles_per_row;  // tiles of 64 pixels along a row
  long n_tiles;       // N * rows * tiles_per_row (DGRAD with up2: rows = H / 2, a tile = 64 pixels of TWO adjacent rows)
};

__device__ __forceinline__ int reflect(int i, int L) { return i < 0 ? -i : (i >= L ? 2 * L - 2 - i : i); }

// ---------------------------------------------------------------------------------------------------------------------------------
// The same products on the bf16 matrix pipe, exactly.  An fp32 number is the sum of three bf16 numbers (8 + 8 + 8 significand bits:
// h = the top 16 bits, m = the top 16 bits of x - h, l = x - h - m; all three subtractions are exact), so
//     x w = xh wh + (xh wm + xm wh) + (xm wm + xh wl + xl wh) + [xm wl + xl wm + xl wl: below 2^-24 |x w|, dropped]
// six v_mfma_f32_32x32x16_bf16 (8 passes, k = 16) instead of eight v_mfma_f32_32x32x2_f32 (16 passes, k = 2 each): 0.375 of the matrix
// time, every product exact in the fp32 accumulator, no range to scale (bf16 has fp32's exponent).  Measured against fp64 the sums are
// CLOSER than the fp32-input MFMA's (blocked accumulation: tests/test_gpu_conv.py keeps its bars).
// What it costs: the split -- 5.5 VALU instructions per activation, done where the operand is formed (a lane's two float4 = 8
// consecutive channels of one pixel = one bf16 A fragment per term) -- and LDS: three terms x 2 bytes per weight are 221 KB for the
// layer, so a work-group keeps the weights of HALF the output channels (110.6 KB) and two work-groups of the same XCD walk the same
// tiles (the second read of a tile comes from that XCD's L2).
constexpr int kConvBfLds = 9 * 4 * 3 * 64 * 16;   // bytes: [tap][k-step of 16][term h, m, l][lane][8 bf16]
// weights -> LDS for one half of the columns: byte (((tap * 4 + s) * 3 + term) * 64 + lane) * 16 + 2 i holds term `term` of the weight
// at k = 16 s + 8 h + i, column `half` * 32 + col (lane = 32 h + col).  FWD: k = ci, column = co;  TRANSPOSED (data gradient): k = co,
// column = ci.  The threads walk w in memory order (coalesced), every thread splits one weight and drops three 2-byte pieces.
template <bool TRANSPOSED>
__device__ __forceinline__ void stage_conv_weights_bf(char* lds, const float* __restrict__ w, int half) {
  for (int e = threadIdx.x; e < 32 * 64 * 9; e += blockDim.x) {
    int co, ci, tap;
    if (TRANSPOSED) {   // co 0..63, ci in the half: runs of 32 x 9 contiguous floats
      co = e / (32 * 9);
      const int r = e - co * (32 * 9);
      ci = half * 32 + r / 9, tap = r - (r / 9) * 9;
    } else {            // co in the half: one contiguous block of 32 x 64 x 9 floats
      co = half * 32 + e / (64 * 9);
      const int r = e % (64 * 9);
      ci = r / 9, tap = r - ci * 9;
    }
    const float x = w[(co * 64 + ci) * 9 + tap];
    const float r1 = x - u2f(f2u(x) & 0xFFFF0000u);
    const float r2 = r1 - u2f(f2u(r1) & 0xFFFF0000u);
    const int k = TRANSPOSED ? co : ci, column = (TRANSPOSED ? ci : co) & 31;
    const int sidx = k >> 4, hh = (k >> 3) & 1, i = k & 7;
    unsigned short* dst = reinterpret_cast<unsigned short*>(lds + ((tap * 4 + sidx) * 3 * 64 + hh * 32 + column) * 16 + 2 * i);
    dst[0] = (unsigned short)(f2u(x) >> 16), dst[512] = (unsigned short)(f2u(r1) >> 16), dst[1024] = (unsigned short)(f2u(r2) >> 16);
  }
}

// the six products of one k-step on one accumulator, smallest terms first; W_IS_A: D[column of W][pixel] instead of D[pixel][column]
template <bool W_IS_A>
__device__ __forceinline__ void mfma6(f32x16& acc, const bf8& xh, const bf8& xm, const bf8& xl, const bf8& wh, const bf8& wm, const bf8& wl) {
  if (W_IS_A) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, xm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, xh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, wh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, wm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, wh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wh, acc, 0, 0, 0);
  }
}

// One source row piece (32 pixel slots x 64 k, the lane's eight float4) against the THREE taps of a kernel row: the split of the
// activations is done once and serves all three (acc3[tx] += X . W[ty][tx]); the shift by tx - 1 pixels happens on the accumulators,
// once per tile (conv_shift_*).  lds -> this lane's fragment of tap (ty, 0), k-step 0, term h.
template <bool W_IS_A>
__device__ __forceinline__ void conv_row3_bf(f32x16 (&acc3)[3], const char* wt, const float4 (&xa)[8]) {
#pragma unroll
  for (int sidx = 0; sidx < 4; ++sidx) {
    bf8 xh, xm, xl;
#if defined(BTS_CONV_ABL) && (BTS_CONV_ABL & 1)   // timing ablation: no split arithmetic
    xh = __builtin_bit_cast(bf8, xa[2 * sidx]), xm = __builtin_bit_cast(bf8, xa[2 * sidx + 1]), xl = xh;
#else
    split3_frag(xa[2 * sidx], xa[2 * sidx + 1], xh, xm, xl);
#endif
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
      const char* f = wt + (tx * 4 + sidx) * (3 * 1024);
      const bf8 wh = *reinterpret_cast<const bf8*>(f), wm = *reinterpret_cast<const bf8*>(f + 1024), wl = *reinterpret_cast<const bf8*>(f + 2048);
      mfma6<W_IS_A>(acc3[tx], xh, xm, xl, wh, wm, wl);
    }
  }
}

// How many 32-slot pixel tiles a wave's tile has: 2 (64 slots, 62 outputs; 8 waves per work-group, 256 VGPRs) or 1 (32 slots, 30 outputs;
// 12 waves per work-group = three per SIMD inside 168 VGPRs -- more waves to hide the loads and the split behind, 3 % more matrix work
// for the two halo slots).  The product's choice is the measured one (profiles/r05s); the other is an A/B build (-DBTS_CONV_PT=..).
#ifndef BTS_CONV_PT
#define BTS_CONV_PT 2
#endif
constexpr int kPT = BTS_CONV_PT;
constexpr int kConvSlots = 32 * kPT;
constexpr int kConvWaves = kPT == 2 ? 8 : 12;
// tile geometry of the bf16 kernels: a wave's tile = kConvSlots pixel SLOTS of one row, slot j <-> position x0 - 1 + j, x0 = kConvOut *
// (tile in row); the tile's outputs are the slots 1 .. kConvOut (their left / right neighbours are slots of the same tile).  The data
// gradient behind an x2 upsampling: position x0 - 2 + j, outputs 2 .. kConvSlots - 3 (kConvOutUp of them)
constexpr int kConvOut = kConvSlots - 2, kConvOutUp = kConvSlots - 4;

// which half of the columns and which stream of tiles a work-group takes: the two halves of a stream sit on the same XCD (blockIdx % 8)
struct ConvRole {
  int half, first, stride;   // wave tile index = first + k * stride
};
__device__ __forceinline__ ConvRole conv_role(int wave) {
  const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
  const int stream = (idx >> 1) * 8 + xcd;       // gridDim.x is a multiple of 16 (conv_grid_bf)
  return ConvRole{idx & 1, stream * kConvWaves + wave, (int)(gridDim.x >> 1) * kConvWaves};
}

__device__ __forceinline__ int reflect_slot(int pos, int L) {   // reflect; positions no output of the tile needs are clamped
  const int r = pos < 0 ? -pos : (pos >= L ? 2 * L - 2 - pos : pos);
  return min(max(r, 0), L - 1);
}

// value of slot - 1 (DIR = -1) or slot + 1 (DIR = +1) of an accumulator pair in the D[pixel slot][column] layout (rows = slots:
// tile row pt * 32 + 8 g + 4 h + e lives in register 4 g + e of lane half h): a neighbouring register, or -- across a group of four --
// the other lane half's
template <int DIR>
__device__ __forceinline__ void conv_shift_rows(const f32x16 (&a)[kPT], f32x16 (&out)[kPT], int h) {
#pragma unroll
  for (int pt = 0; pt < kPT; ++pt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (DIR < 0) {
        // half 1 needs half 0's (pt, 4 g + 3); half 0 needs half 1's register 4 (g - 1) + 3 of this tile, or (pt - 1, 15)
        const float from_h0 = a[pt][4 * g + 3];
        const float from_h1 = g > 0 ? a[pt][g > 0 ? 4 * g - 1 : 0] : (pt > 0 ? a[pt > 0 ? pt - 1 : 0][15] : 0.0f);
        const float got = __shfl_xor(h ? from_h1 : from_h0, 32, 64);
        out[pt][4 * g] = got;
#pragma unroll
        for (int e = 1; e < 4; ++e) out[pt][4 * g + e] = a[pt][4 * g + e - 1];
      } else {
        // half 0 needs half 1's (pt, 4 g); half 1 needs half 0's register 4 (g + 1) of this tile, or (pt + 1, 0)
        const float from_h1 = a[pt][4 * g];
        const float from_h0 = g < 3 ? a[pt][g < 3 ? 4 * g + 4 : 0] : (pt + 1 < kPT ? a[pt + 1 < kPT ? pt + 1 : 0][0] : 0.0f);
        const float got = __shfl_xor(h ? from_h1 : from_h0, 32, 64);
        out[pt][4 * g + 3] = got;
#pragma unroll
        for (int e = 0; e < 3; ++e) out[pt][4 * g + e] = a[pt][4 * g + e + 1];
      }
    }
}
// the same in the D[column][pixel slot] layout (slots across the 32 lanes of a half, tile after tile): the neighbouring lane's, or the
// neighbouring tile's end lane
template <int DIR>
__device__ __forceinline__ void conv_shift_lanes(const f32x16 (&a)[kPT], f32x16 (&out)[kPT], int lane) {
  const int col = lane & 31;
  const int nb = DIR < 0 ? (lane - 1) & 63 : (lane + 1) & 63;           // the neighbour inside the half (wrong for the half's end lane)
  const int wrap = DIR < 0 ? (lane + 31) & 63 : (lane - 31) & 63;       // the neighbouring tile's last / first lane of the same half
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int pt = 0; pt < kPT; ++pt) {
      const float n = __shfl(a[pt][r], nb, 64);
      if (DIR < 0) {
        if (pt > 0) {
          const float w = __shfl(a[pt > 0 ? pt - 1 : 0][r], wrap, 64);
          out[pt][r] = col == 0 ? w : n;
        } else {
          out[pt][r] = n;      // (slot 0 has no output: what lane col 0 gets here is never used)
        }
      } else {
        if (pt + 1 < kPT) {
          const float w = __shfl(a[pt + 1 < kPT ? pt + 1 : 0][r], wrap, 64);
          out[pt][r] = col == 31 ? w : n;
        } else {
          out[pt][r] = n;      // (the last slot has no output)
        }
      }
    }
}

// forward: one wave = kConvOut output pixels of a row x the 32 output channels of its work-group's half; per kernel row ONE set of loads and
// one split, three taps; the next (row, half tile)'s eight loads are in flight under the products of this one
template <bool OUT_NCHW>
__global__ __launch_bounds__(64 * kConvWaves) void conv_fwd_bf_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds_b[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, col = lane & 31;
  const ConvRole role = conv_role(wave);
  stage_conv_weights_bf<false>(lds_b, p.w, role.half);
  __syncthreads();
  const int Hs = p.up2 ? p.H >> 1 : p.H, Ws = p.up2 ? p.W >> 1 : p.W;
  const long rows = (long)p.H * p.tiles_per_row;
  const int c0 = role.half * 32;
  const char* const wlane = lds_b + lane * 16;
  for (long tile = role.first; tile < p.n_tiles; tile += role.stride) {
    const int img = (int)(tile / rows);
    const int rem = (int)(tile - (long)img * rows);
    const int y = rem / p.tiles_per_row, x0 = (rem - y * p.tiles_per_row) * kConvOut;
    const float* base = p.x + (long)img * Hs * Ws * 64;
    f32x16 acc[3][kPT];   // [tx][pt]
#pragma unroll
    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
      for (int pt = 0; pt < kPT; ++pt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tx][pt][r] = (tx != 1 || !p.bias) ? 0.0f : (OUT_NCHW ? p.bias[c0 + mfma_row(r, h)] : p.bias[c0 + col]);
    int sx[kPT];
#pragma unroll
    for (int pt = 0; pt < kPT; ++pt) {
      sx[pt] = reflect_slot(x0 - 1 + pt * 32 + col, p.W);
      if (p.up2) sx[pt] >>= 1;
    }
    float4 xa[2][8];
    auto load_piece = [&](int buf, int piece) {   // piece = kPT ty + pt
      const int ty = piece / kPT, pt = piece % kPT;
      int sy = reflect(y + ty - 1, p.H);
      if (p.up2) sy >>= 1;
      const float* src = base + (unsigned)((sy * Ws + sx[pt]) * 64 + 8 * h);
#pragma unroll
      for (int q = 0; q < 8; ++q) xa[buf][q] = *reinterpret_cast<const float4*>(src + 16 * (q >> 1) + 4 * (q & 1));
    };
    load_piece(0, 0);
#pragma unroll
    for (int piece = 0; piece < 3 * kPT; ++piece) {
#if !(defined(BTS_CONV_ABL) && (BTS_CONV_ABL & 2))   // (timing ablation 2: one piece's loads per tile)
      if (piece + 1 < 3 * kPT) load_piece((piece + 1) & 1, piece + 1);
#endif
      __builtin_amdgcn_sched_barrier(0);
      const int ty = piece / kPT, pt = piece % kPT;
      f32x16 a3[3] = {acc[0][pt], acc[1][pt], acc[2][pt]};
      conv_row3_bf<OUT_NCHW>(a3, wlane + ty * 3 * 4 * (3 * 1024), xa[piece & 1]);
      acc[0][pt] = a3[0], acc[1][pt] = a3[1], acc[2][pt] = a3[2];
      __builtin_amdgcn_sched_barrier(0);
    }
    // out[slot] = P0[slot - 1] + P1[slot] + P2[slot + 1]   (P_tx = the products of kernel column tx at the SOURCE slot)
    f32x16 lo[kPT], hi[kPT];
    if (OUT_NCHW) conv_shift_lanes<-1>(acc[0], lo, lane), conv_shift_lanes<+1>(acc[2], hi, lane);
    else conv_shift_rows<-1>(acc[0], lo, h), conv_shift_rows<+1>(acc[2], hi, h);
    float* out = p.y + (long)img * p.H * p.W * 64;
#pragma unroll
    for (int pt = 0; pt < kPT; ++pt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = (lo[pt][r] + acc[1][pt][r]) + hi[pt][r];
        if (p.elu) v = v > 0.0f ? v : expm1f(v);
#if defined(BTS_CONV_ABL) && (BTS_CONV_ABL & 4)   // timing ablation: no stores
        if (v != 1.2345e-30f) continue;
#endif
        const int slot = pt * 32 + (OUT_NCHW ? col : mfma_row(r, h));
        const int x = x0 - 1 + slot;
        const bool ok = slot >= 1 && slot <= kConvOut && x < p.W;
        if (OUT_NCHW) {
          if (ok) out[(unsigned)(((c0 + mfma_row(r, h)) * p.H + y) * p.W + x)] = v;
        } else {
          if (ok) out[(unsigned)((y * p.W + x) * 64 + c0 + col)] = v;
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// data gradient on the bf16 pipe, the same economy: Q_t[q] = W[t]^T dy[q] for the three taps of a kernel row from ONE split of the
// row piece of dy, accumulated over the (t_y, q_y) pairs of the output row (conv_dgrad_kernel's list), and
//     dx[x] = Q_-1[x + 1] + Q_0[x] + Q_+1[x - 1]  (+ Q_-1[0] for x == 1, + Q_+1[W - 1] for x == W - 2: the reflected columns)
// with dy read as zero outside the image -- shifts of the accumulators, once per tile.  Slots: position = x0 - off + slot; off = 1 and
// 62 outputs per tile, or, with the x2 upsampling in front of the layer (its 2 x 2 children are summed here), off = 2 and 60 outputs
// so that the two children of a source pixel are the slots (2 m, 2 m + 1): neighbouring registers of one lane.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * kConvWaves) void conv_dgrad_bf_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds_b[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, col = lane & 31;
  const ConvRole role = conv_role(wave);
  stage_conv_weights_bf<true>(lds_b, p.w, role.half);
  __syncthreads();
  const int out_rows = p.up2 ? p.H >> 1 : p.H;
  const long rows = (long)out_rows * p.tiles_per_row;
  const int c0 = role.half * 32;
  const int off = p.up2 ? 2 : 1, n_out = p.up2 ? kConvOutUp : kConvOut;
  const char* const wlane = lds_b + lane * 16;
  for (long tile = role.first; tile < p.n_tiles; tile += role.stride) {
    const int img = (int)(tile / rows);
    const int rem = (int)(tile - (long)img * rows);
    const int yt = rem / p.tiles_per_row, x0 = (rem - yt * p.tiles_per_row) * n_out;
    const float* base = p.x + (long)img * p.H * p.W * 64;   // dy' (N, H, W, 64)
    f32x16 acc[3][kPT];   // [tx + 1][pt]
#pragma unroll
    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
      for (int pt = 0; pt < kPT; ++pt) acc[tx][pt] = zero_acc();
    int qx[kPT];
    bool ok[kPT];
#pragma unroll
    for (int pt = 0; pt < kPT; ++pt) {
      const int pos = x0 - off + pt * 32 + col;
      ok[pt] = pos >= 0 && pos < p.W;
      qx[pt] = min(max(pos, 0), p.W - 1);
    }
    // the (t_y, q_y) pairs of the tile's output row(s), wave-uniform: up to 4 per row, two rows with up2
    int pair_ty[8], pair_qy[8], n_pairs = 0;
    for (int sub = 0; sub < (p.up2 ? 2 : 1); ++sub) {
      const int y = p.up2 ? 2 * yt + sub : yt;
      for (int ry = 0; ry < 5; ++ry) {
        int ty, qy;
        if (ry < 3) ty = ry - 1, qy = y - ty;
        else if (ry == 3) ty = -1, qy = (y == 1) ? 0 : -1;
        else ty = 1, qy = (y == p.H - 2) ? p.H - 1 : -1;
        if (qy < 0 || qy >= p.H) continue;
        pair_ty[n_pairs] = ty + 1, pair_qy[n_pairs] = qy, ++n_pairs;
      }
    }
    float4 xa[2][8];
    auto load_piece = [&](int buf, int piece) {   // piece = kPT pair + pt
      const int qy = pair_qy[piece / kPT], pt = piece % kPT;
      const float* src = base + (unsigned)((qy * p.W + qx[pt]) * 64 + 8 * h);
      const bool on = ok[pt];
#pragma unroll
      for (int q = 0; q < 8; ++q) xa[buf][q] = on ? *reinterpret_cast<const float4*>(src + 16 * (q >> 1) + 4 * (q & 1)) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    };
    // (the pair list is short and its length wave-uniform: a rolled loop over TWO pieces at a time, so that the two buffers and the
    // accumulators of a piece are compile-time choices: pieces 2 j and 2 j + 1 are the two tiles of pair j, or -- with one tile per wave --
    // pairs 2 j and 2 j + 1)
    const int n_pieces = kPT * n_pairs;
    load_piece(0, 0);
    for (int piece = 0; piece < n_pieces; piece += 2) {
      if (piece + 1 < n_pieces) load_piece(1, piece + 1);
      __builtin_amdgcn_sched_barrier(0);
      {
        f32x16 a3[3] = {acc[0][0], acc[1][0], acc[2][0]};
        conv_row3_bf<false>(a3, wlane + pair_ty[piece / kPT] * 3 * 4 * (3 * 1024), xa[0]);
        acc[0][0] = a3[0], acc[1][0] = a3[1], acc[2][0] = a3[2];
      }
      __builtin_amdgcn_sched_barrier(0);
      if (piece + 1 < n_pieces) {
        if (piece + 2 < n_pieces) load_piece(0, piece + 2);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 a3[3] = {acc[0][kPT - 1], acc[1][kPT - 1], acc[2][kPT - 1]};
        conv_row3_bf<false>(a3, wlane + pair_ty[(piece + 1) / kPT] * 3 * 4 * (3 * 1024), xa[1]);
        acc[0][kPT - 1] = a3[0], acc[1][kPT - 1] = a3[1], acc[2][kPT - 1] = a3[2];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // dx[slot] = Q_-1[slot + 1] + Q_0[slot] + Q_+1[slot - 1] (+ the reflected columns)
    f32x16 up[kPT], dn[kPT];
    conv_shift_rows<+1>(acc[0], up, h), conv_shift_rows<-1>(acc[2], dn, h);
    f32x16 v[kPT];
#pragma unroll
    for (int pt = 0; pt < kPT; ++pt)
#pragma unroll
      for (int r = 0; r < 16; ++r) v[pt][r] = (up[pt][r] + acc[1][pt][r]) + dn[pt][r];
    const int s1 = 1 - x0 + off, s2 = p.W - 2 - x0 + off;   // the slots of x == 1 and x == W - 2 (wave-uniform; may lie outside this tile)
    if (s1 >= off && s1 < off + n_out) {
      f32x16 e[kPT];
      conv_shift_rows<-1>(acc[0], e, h);
#pragma unroll
      for (int pt = 0; pt < kPT; ++pt)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[pt][r] += (pt * 32 + mfma_row(r, h) == s1) ? e[pt][r] : 0.0f;
    }
    if (s2 >= off && s2 < off + n_out) {
      f32x16 e[kPT];
      conv_shift_rows<+1>(acc[2], e, h);
#pragma unroll
      for (int pt = 0; pt < kPT; ++pt)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[pt][r] += (pt * 32 + mfma_row(r, h) == s2) ? e[pt][r] : 0.0f;
    }
    if (p.up2) {
      const int Ws = p.W >> 1;
      float* out = p.y + ((long)img * out_rows + yt) * Ws * 64;
#pragma unroll
      for (int pt = 0; pt < kPT; ++pt)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {   // slots (2 m, 2 m + 1): the two children of one source pixel along x
          const int slot = pt * 32 + mfma_row(r, h);
          const int x = x0 - off + slot;
          if (slot >= off && slot < off + n_out && x < p.W) out[(unsigned)((x >> 1) * 64 + c0 + col)] = v[pt][r] + v[pt][r + 1];
        }
    } else {
      float* out = p.y + ((long)img * p.H + yt) * p.W * 64;
#pragma unroll
      for (int pt = 0; pt < kPT; ++pt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int slot = pt * 32 + mfma_row(r, h);
          const int x = x0 - off + slot;
          if (slot >= off && slot < off + n_out && x < p.W) out[(unsigned)(x * 64 + c0 + col)] = v[pt][r];
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// weight gradient: dW[co][ci][t] = sum over all pixels p of dy'[p][co] x[reflect(p + t)][ci];  db[co] = sum_p dy'[p][co].
// D[co][ci] = A[co][k = pixel] . B[k = pixel][ci]: both operands are 128-byte row pieces of the channels-last tensors (a lane half = 32
// consecutive channels of one pixel; the two halves = the two pixels of a k-step).  A work-group = 8 waves = 2 pixel streams x the 4
// (co half, ci half) quadrants; a wave keeps its quadrant of all 9 taps in registers (9 accumulator tiles) over the whole launch and
// writes it once: partial sums per (work-group, stream) into the workspace, a second small kernel adds them up -- no atomics, the same
// bits on every run.
// ---------------------------------------------------------------------------------------------------------------------------------
struct WgradParams {
  const float* x;     // layer input (N, Hs, Ws, 64)
  const float* dy;    // dy' (N, H, W, 64)
  float* part;        // (gridDim.x * 2, 9 * 64 * 64 + 64) partial sums: [tap][co][ci], then db
  int N, H, W, up2, tiles_per_row;
  long n_tiles;
};
constexpr int kWgradPart = 9 * kTapFloats + 64;

// ---------------------------------------------------------------------------------------------------------------------------------
// weight gradient on the bf16 pipe.  k = 16 consecutive pixels of a row per matrix instruction: lane (h, channel) holds the 8 pixels
// 8 h .. 8 h + 7 of ITS channel -- 8 dword loads whose 32 lanes are 32 consecutive channels of one pixel (one 128-byte line) for dy, and
// 10 for x: the pixels -1 .. 8, from which the fragments of the three taps of a kernel row are three different pairings of ONE split
// (tap tx pairs the values i + tx, i + tx + 1).  A work-group = 12 waves = 3 kernel rows x the 4 (co half, ci half) quadrants, three
// accumulator tiles per wave (the fp32 kernel's 9 per wave are 144 registers; with the operands of a k-step, three terms each, and
// the next k-step's loads in flight this stays under 168: three waves per SIMD).  No LDS.  Partial sums per work-group into the
// workspace, conv_wgrad_reduce_kernel adds them up -- no atomics, the same bits on every run.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(768) void conv_wgrad_bf_kernel(const WgradParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, col = lane & 31;
  const int ty = wave >> 2, quad = wave & 3, ct = quad >> 1, cit = quad & 1;
  const int Hs = p.up2 ? p.H >> 1 : p.H, Ws = p.up2 ? p.W >> 1 : p.W;
  const int nks = (p.W + 15) >> 4;                 // k-steps per row
  const long n_rows = (long)p.N * p.H;
  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) acc[t] = zero_acc();
  float db = 0.0f;
  // the work-group's k-steps in order: rows blockIdx.x, + gridDim.x, ...; the loads of step f + 1 are issued before the products of step f
  long row = blockIdx.x;
  int ks = 0;
  float an[8], bn[10];
  auto load_step = [&](long r, int k) {
    const int img = (int)(r / p.H), y = (int)(r - (long)img * p.H);
    int sy = reflect(y + ty - 1, p.H);
    if (p.up2) sy >>= 1;
    const float* dyrow = p.dy + ((long)img * p.H + y) * p.W * 64 + ct * 32 + col;
    const float* xrow = p.x + ((long)img * Hs + sy) * Ws * 64 + cit * 32 + col;
    const int xa0 = 16 * k + 8 * h;                // this lane half's first pixel
    if (16 * k >= 16 && 16 * k + 32 <= p.W) {      // interior step (wave-uniform): no reflection, no ragged end -- constant offsets
      const float* ap = dyrow + (unsigned)(xa0 * 64);
#pragma unroll
      for (int i = 0; i < 8; ++i) an[i] = ap[i * 64];
      if (p.up2) {
        const float* bp = xrow + (unsigned)((xa0 >> 1) * 64);      // pixel xa0 + i - 1 -> source (xa0 + i - 1) >> 1 = xa0 / 2 + ((i - 1) >> 1)
#pragma unroll
        for (int i = 0; i < 10; ++i) bn[i] = bp[((i - 1) >> 1) * 64];
      } else {
        const float* bp = xrow + (unsigned)((xa0 - 1) * 64);
#pragma unroll
        for (int i = 0; i < 10; ++i) bn[i] = bp[i * 64];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int x = xa0 + i;
        an[i] = x < p.W ? dyrow[(unsigned)(x * 64)] : 0.0f;   // (a pixel beyond a ragged row end contributes nothing)
      }
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        int sx = reflect_slot(xa0 + i - 1, p.W);
        if (p.up2) sx >>= 1;
        bn[i] = xrow[(unsigned)(sx * 64)];
      }
    }
  };
  if (row < n_rows) load_step(row, 0);
  while (row < n_rows) {
    float a[8], b[10];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = an[i];
#pragma unroll
    for (int i = 0; i < 10; ++i) b[i] = bn[i];
    int nk = ks + 1;
    long nrow = row;
    if (nk == nks) nk = 0, nrow += gridDim.x;
    if (nrow < n_rows) load_step(nrow, nk);
    __builtin_amdgcn_sched_barrier(0);
    WSplit sa[8], sb[10];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      db += a[i];
      sa[i] = wsplit(a[i]);
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) sb[i] = wsplit(b[i]);
    bf8 ah, am, al;
    wfrags(sa, 0, ah, am, al);
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
      bf8 bh, bm, bl;
      wfrags(sb, tx, bh, bm, bl);
      // D[co][ci] += dy^T x: A = dy (rows = co), B = x (columns = ci); smallest terms first
      acc[tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[tx], 0, 0, 0);
      acc[tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[tx], 0, 0, 0);
      acc[tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[tx], 0, 0, 0);
      acc[tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[tx], 0, 0, 0);
      acc[tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[tx], 0, 0, 0);
      acc[tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[tx], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    ks = nk, row = nrow;
  }
  float* part = p.part + (long)blockIdx.x * kWgradPart;
#pragma unroll
  for (int tx = 0; tx < 3; ++tx)
#pragma unroll
    for (int r = 0; r < 16; ++r) part[(unsigned)((ty * 3 + tx) * kTapFloats + (ct * 32 + mfma_row(r, h)) * 64 + cit * 32 + col)] = acc[tx][r];
  if (ty == 1 && cit == 0) {
    db += __shfl_xor(db, 32, 64);     // the two pixel groups of every k-step
    if (h == 0) part[9 * kTapFloats + ct * 32 + col] = db;
  }
}

// dW (64, 64, 3, 3) [co][ci][t] and db (64) = sums of the partials; written (not accumulated)
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ part, int n_part, float* __restrict__ d_w, float* __restrict__ d_b) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= kWgradPart) return;
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  int k = 0;
  for (; k + 4 <= n_part; k += 4) {
    s0 += part[(long)k * kWgradPart + i], s1 += part[(long)(k + 1) * kWgradPart + i];
    s2 += part[(long)(k + 2) * kWgradPart + i], s3 += part[(long)(k + 3) * kWgradPart + i];
  }
  for (; k < n_part; ++k) s0 += part[(long)k * kWgradPart + i];
  const float s = (s0 + s1) + (s2 + s3);
  if (i < 9 * kTapFloats) {
    const int t = i / kTapFloats, co = (i % kTapFloats) / 64, ci = i % 64;
    if (d_w) d_w[(co * 64 + ci) * 9 + t] = s;
  } else if (d_b) {
    d_b[i - 9 * kTapFloats] = s;
  }
}

// dy' = dy * elu'(y) from the layer's OUTPUT y (ELU, alpha = 1: elu' = 1 for y > 0, y + 1 otherwise -- torch's backward of the in-place
// ELU of ConvBlock, layers.py:22-31); dy may be channels-last or NCHW (`dy_nchw`: the gradient of an NCHW output), dy' is channels-last
__global__ __launch_bounds__(256) void elu_bwd_kernel(const float4* __restrict__ dy, const float4* __restrict__ y, float4* __restrict__ out, long n4) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 g = dy[i], v = y[i];
    float4 o;
    o.x = g.x * (v.x > 0.0f ? 1.0f : v.x + 1.0f), o.y = g.y * (v.y > 0.0f ? 1.0f : v.y + 1.0f);
    o.z = g.z * (v.z > 0.0f ? 1.0f : v.z + 1.0f), o.w = g.w * (v.w > 0.0f ? 1.0f : v.w + 1.0f);
    out[i] = o;
  }
}

#ifdef BTS_CONV_FP32
#include "bts_conv_fp32.h"
#endif

static int conv_grid() { return device_cu_count(); }   // one persistent work-group of 8 waves per CU (the weights fill its LDS)
// the bf16 kernels: pairs of work-groups (the two halves of the columns) on the same XCD -> a multiple of 16, two work-groups per `want`ed one
static int conv_grid_bf(long want_wg) {
  long g = 2 * want_wg;
  const long cap = device_cu_count() / 16 * 16;
  if (g > cap) g = cap;
  g = (g + 15) / 16 * 16;
  return (int)(g < 16 ? 16 : g);
}

static bool conv_ok(const BtsConv3x3* c, const char* who) {
  if (!c || !c->x || !c->weight || c->N <= 0 || c->H < 4 || c->W < 4 || c->C != 64 || (c->up2 && ((c->H | c->W) & 1))) {
    set_error("%s: NULL pointer, C != 64, a frame below 4 x 4, or an odd size with up2 (N=%ld H=%ld W=%ld)", who, c ? c->N : 0, c ? c->H : 0, c ? c->W : 0);
    return false;
  }
  if ((long)c->H * c->W * 64 > 0x7FFFFFFFL) {
    set_error("%s: tensor too large for 32-bit offsets inside an image (H=%ld W=%ld)", who, c->H, c->W);
    return false;
  }
  return true;
}

int conv3x3_fwd_impl(const BtsConv3x3* c, hipStream_t s) {
  ConvParams p;
  memset(&p, 0, sizeof(p));
  p.x = c->x, p.w = c->weight, p.bias = c->bias, p.y = c->y;
  p.N = c->N, p.H = c->H, p.W = c->W, p.up2 = c->up2, p.elu = c->elu, p.out_nchw = c->out_nchw;
  p.tiles_per_row = (c->W + 63) / 64;
  p.n_tiles = (long)c->N * c->H * p.tiles_per_row;
#ifndef BTS_CONV_FP32
  p.tiles_per_row = (c->W + kConvOut - 1) / kConvOut;
  p.n_tiles = (long)c->N * c->H * p.tiles_per_row;
  const int grid = conv_grid_bf((p.n_tiles + kConvWaves - 1) / kConvWaves);
  // (the attribute is per device: set on every launch, like the render kernels' launchers do -- a host-side table write)
  if (c->out_nchw) {
    (void)hipFuncSetAttribute((const void*)conv_fwd_bf_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kConvBfLds);
    conv_fwd_bf_kernel<true><<<grid, 64 * kConvWaves, kConvBfLds, s>>>(p);
  } else {
    (void)hipFuncSetAttribute((const void*)conv_fwd_bf_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kConvBfLds);
    conv_fwd_bf_kernel<false><<<grid, 64 * kConvWaves, kConvBfLds, s>>>(p);
  }
#else   // A/B build: the fp32-input MFMA kernels of the first version
  const long want = (p.n_tiles + 7) / 8;
  const int grid = (int)(want < conv_grid() ? want : conv_grid());
  const size_t lds = sizeof(float) * kConvLds;
  if (c->out_nchw) {
    (void)hipFuncSetAttribute((const void*)conv_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    conv_fwd_kernel<true><<<grid, 512, lds, s>>>(p);
  } else {
    (void)hipFuncSetAttribute((const void*)conv_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    conv_fwd_kernel<false><<<grid, 512, lds, s>>>(p);
  }
#endif
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: convolution forward launch failed (%ld)", hipGetErrorString(e), (long)e);
    return BTS_E_LAUNCH;
  }
  return BTS_OK;
}

size_t conv3x3_bwd_workspace_impl(const BtsConv3x3* c) {
  const size_t dyp = sizeof(float) * (size_t)c->N * c->H * c->W * 64;            // dy' channels-last
  const size_t part = sizeof(float) * (size_t)conv_grid() * 2 * kWgradPart;     // weight-gradient partial sums
  return ((dyp + 255) & ~(size_t)255) + part;
}

int conv3x3_bwd_impl(const BtsConv3x3* c, const float* g_y, void* workspace, size_t ws_bytes, float* d_x, float* d_weight, float* d_bias, hipStream_t s) {
  if (ws_bytes < conv3x3_bwd_workspace_impl(c) || !workspace) {
    set_error("%s: workspace too small (%ld bytes needed)", "bts_conv3x3_bwd", (long)conv3x3_bwd_workspace_impl(c));
    return BTS_E_WORKSPACE;
  }
  float* dyp = static_cast<float*>(workspace);
  const size_t dyp_bytes = (sizeof(float) * (size_t)c->N * c->H * c->W * 64 + 255) & ~(size_t)255;
  float* part = reinterpret_cast<float*>(static_cast<char*>(workspace) + dyp_bytes);
  const long n = (long)c->N * c->H * c->W * 64;
  const float* dy = g_y;
  if (c->out_nchw) {          // the gradient of an NCHW output (the renderer's d_feat) -> channels-last
    if (c->elu) {
      set_error("%s: ELU with an NCHW output is not a layer of the decoder", "bts_conv3x3_bwd");
      return BTS_E_UNSUPPORTED;
    }
    if (int rc = transpose_launch(g_y, dyp, c->N, 64, c->H, c->W, true, s)) return rc;
    dy = dyp;
  } else if (c->elu) {
    const long n4 = n / 4;
    const long want = (n4 + 255) / 256;
    elu_bwd_kernel<<<(int)(want < 4096 ? want : 4096), 256, 0, s>>>(reinterpret_cast<const float4*>(g_y), reinterpret_cast<const float4*>(c->y),
                                                                   reinterpret_cast<float4*>(dyp), n4);
    dy = dyp;
  }
  const int tpr = (c->W + 63) / 64;
#ifdef BTS_CONV_FP32
  const size_t lds = sizeof(float) * kConvLds;
#endif
  if (d_x) {
    ConvParams p;
    memset(&p, 0, sizeof(p));
    p.x = dy, p.w = c->weight, p.y = d_x, p.N = c->N, p.H = c->H, p.W = c->W, p.up2 = c->up2, p.tiles_per_row = tpr;
#ifndef BTS_CONV_FP32
    const int n_out = c->up2 ? kConvOutUp : kConvOut;
    p.tiles_per_row = (c->W + n_out - 1) / n_out;
    p.n_tiles = (long)c->N * (c->up2 ? c->H / 2 : c->H) * p.tiles_per_row;
    (void)hipFuncSetAttribute((const void*)conv_dgrad_bf_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kConvBfLds);
    conv_dgrad_bf_kernel<<<conv_grid_bf((p.n_tiles + kConvWaves - 1) / kConvWaves), 64 * kConvWaves, kConvBfLds, s>>>(p);
#else
    p.n_tiles = (long)c->N * (c->up2 ? c->H / 2 : c->H) * tpr;
    const long want = (p.n_tiles + 7) / 8;
    (void)hipFuncSetAttribute((const void*)conv_dgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    conv_dgrad_kernel<<<(int)(want < conv_grid() ? want : conv_grid()), 512, lds, s>>>(p);
#endif
  }
  if (d_weight || d_bias) {
    WgradParams q;
    memset(&q, 0, sizeof(q));
    q.x = c->x, q.dy = dy, q.part = part, q.N = c->N, q.H = c->H, q.W = c->W, q.up2 = c->up2, q.tiles_per_row = tpr;
    q.n_tiles = (long)c->N * c->H * tpr;
#ifndef BTS_CONV_FP32
    const long n_rows = (long)c->N * c->H;
    const int grid = (int)(n_rows < conv_grid() ? n_rows : conv_grid());
    conv_wgrad_bf_kernel<<<grid, 768, 0, s>>>(q);
    conv_wgrad_reduce_kernel<<<(kWgradPart + 255) / 256, 256, 0, s>>>(part, grid, d_weight, d_bias);
#else
    const long want = (q.n_tiles + 1) / 2;
    const int grid = (int)(want < conv_grid() ? want : conv_grid());
    conv_wgrad_kernel<<<grid, 512, 0, s>>>(q);
    conv_wgrad_reduce_kernel<<<(kWgradPart + 255) / 256, 256, 0, s>>>(part, grid * 2, d_weight, d_bias);
#endif
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: convolution backward launch failed (%ld)", hipGetErrorString(e), (long)e);
    return BTS_E_LAUNCH;
  }
  return BTS_OK;
}

}  // namespace bts

using namespace bts;

extern "C" {

int bts_conv3x3_fwd(const BtsConv3x3* c, void* stream) {
  if (!conv_ok(c, "bts_conv3x3_fwd")) return BTS_E_INVALID;
  if (!c->y) {
    set_error("%s: NULL output", "bts_conv3x3_fwd");
    return BTS_E_INVALID;
  }
  return conv3x3_fwd_impl(c, (hipStream_t)stream);
}

size_t bts_conv3x3_bwd_workspace(const BtsConv3x3* c) {
  if (!c || c->N <= 0 || c->H <= 0 || c->W <= 0) return 0;
  return conv3x3_bwd_workspace_impl(c);
}

int bts_conv3x3_bwd(const BtsConv3x3* c, const float* g_y, void* workspace, size_t workspace_bytes, float* d_x, float* d_weight, float* d_bias,
                    void* stream) {
  if (!conv_ok(c, "bts_conv3x3_bwd")) return BTS_E_INVALID;
  if (!g_y || (c->elu && !c->y)) {
    set_error("%s: NULL output gradient, or an ELU layer without its output y", "bts_conv3x3_bwd");
    return BTS_E_INVALID;
  }
  return conv3x3_bwd_impl(c, g_y, workspace, workspace_bytes, d_x, d_weight, d_bias, (hipStream_t)stream);
}

}  // extern "C"
