// bts_conv_fp32.h -- the FIRST version of the decoder-tail kernels: the same three kernels on v_mfma_f32_32x32x2_f32 (fp32 inputs, 157
// TFLOP/s), weights of all 64 output channels in LDS (147 KB), one work-group of 8 waves per CU.  Compiled only with -DBTS_CONV_FP32 (an
// A/B build: `python -m behindthescenes_amd.build --tag convfp32 -DBTS_CONV_FP32`); the product runs the bf16 three-term kernels of
// bts_conv.hip.  Measured at the decoder tail's size (16 x 192 x 640): forward 1.27 ms, data gradient 1.3 - 1.5, weight gradient
// 1.9 - 2.1 (profiles/r05y), against 0.80 / 0.71 / 0.82 for the bf16 kernels (profiles/r05o).
// (included by bts_conv.hip INSIDE namespace bts, behind its parameter structs)
#pragma once

constexpr int kConvLds = 9 * kTapFloats;     // floats: 147 456 bytes

// weights -> LDS: slot (((tap * 8 + q) * 2 + h) * 64 + col) * 4 + j holds Wk[tap][k = 8 q + 4 h + j][col];
//   FWD / WGRAD-free form: k = ci, col = co  (y[co] += W[co][ci] x[ci]);  DGRAD: k = co, col = ci (dx[ci] += W[co][ci] dy[co])
template <bool TRANSPOSED>
__device__ __forceinline__ void stage_conv_weights(float* lds, const float* __restrict__ w) {
  for (int i = threadIdx.x; i < kConvLds; i += blockDim.x) {
    const int j = i & 3, col = (i >> 2) & 63, h = (i >> 8) & 1, q = (i >> 9) & 7, tap = i >> 12;
    const int k = 8 * q + 4 * h + j;
    const int co = TRANSPOSED ? k : col, ci = TRANSPOSED ? col : k;
    lds[i] = w[(co * 64 + ci) * 9 + tap];
  }
}

// one pass of a tile: acc[pt][ct] += X-fragment (64 pixels x 64 k) . Wk[tap] (64 k x 64 cols); `src` = per-lane element offsets of the
// two pixel tiles' source pixels (+ 4 h already applied), `ok` = per-lane validity (a masked lane contributes zeros)
template <bool OUT_NCHW>
__device__ __forceinline__ void conv_pass(f32x16 (&acc)[2][2], const float* lds, const float* __restrict__ base, const unsigned (&src)[2], const bool (&ok)[2],
                                          int tap, int h, int col) {
  float4 xa[2][8];
#pragma unroll
  for (int pt = 0; pt < 2; ++pt)
#pragma unroll
    for (int q = 0; q < 8; ++q) xa[pt][q] = ok[pt] ? *reinterpret_cast<const float4*>(base + src[pt] + 8 * q) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  __builtin_amdgcn_sched_barrier(0);   // all sixteen loads in flight before the first MFMA (the scheduler otherwise sinks each to its use)
  const float* wt = lds + tap * kTapFloats + h * 256 + col * 4;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float4 w0 = *reinterpret_cast<const float4*>(wt + q * 512);
    const float4 w1 = *reinterpret_cast<const float4*>(wt + q * 512 + 128);
    const float wv[2][4] = {{w0.x, w0.y, w0.z, w0.w}, {w1.x, w1.y, w1.z, w1.w}};
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const float xv[4] = {xa[pt][q].x, xa[pt][q].y, xa[pt][q].z, xa[pt][q].w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[pt][ct] = OUT_NCHW ? mfma(wv[ct][j], xv[j], acc[pt][ct]) : mfma(xv[j], wv[ct][j], acc[pt][ct]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// forward: one wave = 64 consecutive pixels of an output row x all 64 output channels; 9 passes (taps)
// ---------------------------------------------------------------------------------------------------------------------------------
template <bool OUT_NCHW>
__global__ __launch_bounds__(512) void conv_fwd_kernel(const ConvParams p) {
  extern __shared__ float lds[];
  stage_conv_weights<false>(lds, p.w);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, col = lane & 31;
  const int Hs = p.up2 ? p.H >> 1 : p.H, Ws = p.up2 ? p.W >> 1 : p.W;
  const long rows = (long)p.H * p.tiles_per_row;
  for (long tile = (long)blockIdx.x * 8 + wave; tile < p.n_tiles; tile += (long)gridDim.x * 8) {
    const int img = (int)(tile / rows);
    const int rem = (int)(tile - (long)img * rows);
    const int y = rem / p.tiles_per_row, x0 = (rem - y * p.tiles_per_row) * 64;
    const float* base = p.x + (long)img * Hs * Ws * 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        // the accumulators are born from the bias: D[pixel][co] (channels across the lanes) or D[co][pixel] (channels in the registers)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[pt][ct][r] = !p.bias ? 0.0f : (OUT_NCHW ? p.bias[ct * 32 + mfma_row(r, h)] : p.bias[ct * 32 + col]);
      }
    const int xl[2] = {min(x0 + col, p.W - 1), min(x0 + 32 + col, p.W - 1)};   // (lanes beyond a ragged row end repeat its last pixel; not stored)
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
      int sy = reflect(y + ty - 1, p.H);
      if (p.up2) sy >>= 1;
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        unsigned src[2];
        const bool ok[2] = {true, true};
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
          int sx = reflect(xl[pt] + tx - 1, p.W);
          if (p.up2) sx >>= 1;
          src[pt] = (unsigned)((sy * Ws + sx) * 64 + 4 * h);
        }
        conv_pass<OUT_NCHW>(acc, lds, base, src, ok, ty * 3 + tx, h, col);
      }
    }
    // epilogue: ELU, store.  NHWC: row r of the tile = pixel, the 32 lanes of a half = 32 consecutive channels (128-byte pieces);
    // NCHW: row r = channel, the lanes = 32 consecutive pixels of that channel's row
    float* out = p.y + (long)img * p.H * p.W * 64;
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[pt][ct][r];
          if (p.elu) v = v > 0.0f ? v : expm1f(v);
          if (OUT_NCHW) {
            const int x = x0 + pt * 32 + col;
            if (x < p.W) out[(unsigned)(((ct * 32 + mfma_row(r, h)) * p.H + y) * p.W + x)] = v;
          } else {
            const int x = x0 + pt * 32 + mfma_row(r, h);
            if (x < p.W) out[(unsigned)((y * p.W + x) * 64 + ct * 32 + col)] = v;
          }
        }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// data gradient: dx[p] = sum over the (q, t) with reflect(q + t) = p of W[t]^T dy[q].  Per axis (length L, coordinate c): q = c - t for
// t = -1, 0, +1 where that lies inside, plus (q = 0, t = -1) for c == 1 and (q = L - 1, t = +1) for c == L - 2 -- the padded lines -1
// and L are copies of lines 1 and L - 2, so what the convolution read there flows back onto those.  Rows are wave-uniform (a tile is one
// row); along the row the two extra pairs concern ONE lane each and run as passes of their own in the tiles that hold x = 1 / x = W - 2.
// With the x2 upsampling in front of the layer a tile is 64 pixels of rows 2 ys and 2 ys + 1: both rows accumulate into the same
// registers, horizontal neighbours are summed in registers (rows 2 m, 2 m + 1 of the MFMA tile sit in one lane), and the wave stores the
// gradient of 32 source pixels.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void conv_dgrad_kernel(const ConvParams p) {
  extern __shared__ float lds[];
  stage_conv_weights<true>(lds, p.w);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, col = lane & 31;
  const int out_rows = p.up2 ? p.H >> 1 : p.H;
  const long rows = (long)out_rows * p.tiles_per_row;
  for (long tile = (long)blockIdx.x * 8 + wave; tile < p.n_tiles; tile += (long)gridDim.x * 8) {
    const int img = (int)(tile / rows);
    const int rem = (int)(tile - (long)img * rows);
    const int yt = rem / p.tiles_per_row, x0 = (rem - yt * p.tiles_per_row) * 64;
    const float* base = p.x + (long)img * p.H * p.W * 64;   // dy' (N, H, W, 64)
    f32x16 acc[2][2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) acc[pt][ct] = zero_acc();
    const int xq[2] = {x0 + col, x0 + 32 + col};
    for (int sub = 0; sub < (p.up2 ? 2 : 1); ++sub) {
      const int y = p.up2 ? 2 * yt + sub : yt;
      for (int ry = 0; ry < 5; ++ry) {     // 0..2: t_y = ry - 1, q_y = y - t_y; 3: (q_y = 0, t_y = -1) for y == 1; 4: (q_y = H - 1, t_y = +1) for y == H - 2
        int ty, qy;
        if (ry < 3) ty = ry - 1, qy = y - ty;
        else if (ry == 3) ty = -1, qy = (y == 1) ? 0 : -1;
        else ty = 1, qy = (y == p.H - 2) ? p.H - 1 : -1;
        if (qy < 0 || qy >= p.H) continue;
        for (int rx = 0; rx < 5; ++rx) {
          int tx;
          unsigned src[2];
          bool ok[2];
          if (rx < 3) {
            tx = rx - 1;
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
              const int qx = xq[pt] - tx;
              ok[pt] = qx >= 0 && qx < p.W && xq[pt] < p.W;
              src[pt] = (unsigned)((qy * p.W + min(max(qx, 0), p.W - 1)) * 64 + 4 * h);
            }
          } else {
            const int xs = rx == 3 ? 1 : p.W - 2, qx = rx == 3 ? 0 : p.W - 1;
            tx = rx == 3 ? -1 : 1;
            if (xs < x0 || xs >= x0 + 64) continue;   // wave-uniform: this tile does not hold the column next to the border
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) ok[pt] = xq[pt] == xs, src[pt] = (unsigned)((qy * p.W + qx) * 64 + 4 * h);
          }
          conv_pass<false>(acc, lds, base, src, ok, (ty + 1) * 3 + (tx + 1), h, col);
        }
      }
    }
    if (p.up2) {
      const int Ws = p.W >> 1;
      float* out = p.y + ((long)img * out_rows + yt) * Ws * 64;
#pragma unroll
      for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {   // tile rows mfma_row(r, h), mfma_row(r, h) + 1: the two children of one source pixel along x
            const int x = x0 + pt * 32 + mfma_row(r, h);
            if (x < p.W) out[(unsigned)((x >> 1) * 64 + ct * 32 + col)] = acc[pt][ct][r] + acc[pt][ct][r + 1];
          }
    } else {
      float* out = p.y + ((long)img * p.H + yt) * p.W * 64;
#pragma unroll
      for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int x = x0 + pt * 32 + mfma_row(r, h);
            if (x < p.W) out[(unsigned)(x * 64 + ct * 32 + col)] = acc[pt][ct][r];
          }
    }
  }
}

__global__ __launch_bounds__(512) void conv_wgrad_kernel(const WgradParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, col = lane & 31;
  const int quad = wave & 3, stream = wave >> 2, ct = quad >> 1, cit = quad & 1;
  const int Hs = p.up2 ? p.H >> 1 : p.H, Ws = p.up2 ? p.W >> 1 : p.W;
  const long rows = (long)p.H * p.tiles_per_row;
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = zero_acc();
  float db = 0.0f;
  for (long tile = (long)blockIdx.x * 2 + stream; tile < p.n_tiles; tile += (long)gridDim.x * 2) {
    const int img = (int)(tile / rows);
    const int rem = (int)(tile - (long)img * rows);
    const int y = rem / p.tiles_per_row, x0 = (rem - y * p.tiles_per_row) * 64;
    const float* xb = p.x + (long)img * Hs * Ws * 64 + cit * 32 + col;
    const float* dyb = p.dy + ((long)img * p.H + y) * p.W * 64 + ct * 32 + col;
    int sy[3];
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
      sy[ty] = reflect(y + ty - 1, p.H);
      if (p.up2) sy[ty] >>= 1;
    }
    // rounds of two k-steps (four pixels: 20 loads); round r + 1's loads are issued BEFORE round r's MFMAs, into the other buffer, so the
    // 18 x 64 cycles of matrix work of a round run under the next round's memory latency
    float a[2][2], b[2][2][9];
    auto load_round = [&](int buf, int s) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int x = x0 + 2 * (s + u) + h;
        const bool in = x < p.W;
        const int xc = min(x, p.W - 1);
        a[buf][u] = in ? dyb[(unsigned)(xc * 64)] : 0.0f;    // (a pixel beyond a ragged row end contributes nothing)
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
          int sx = reflect(xc + tx - 1, p.W);
          if (p.up2) sx >>= 1;
#pragma unroll
          for (int ty = 0; ty < 3; ++ty) b[buf][u][ty * 3 + tx] = xb[(unsigned)((sy[ty] * Ws + sx) * 64)];
        }
      }
    };
    auto mfma_round = [&](int buf) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        db += a[buf][u];
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = mfma(a[buf][u], b[buf][u][t], acc[t]);
      }
    };
    load_round(0, 0);
#pragma unroll 1
    for (int s = 0; s < 32; s += 4) {
      load_round(1, s + 2);
      __builtin_amdgcn_sched_barrier(0);
      mfma_round(0);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 4 < 32) load_round(0, s + 4);
      __builtin_amdgcn_sched_barrier(0);
      mfma_round(1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float* part = p.part + ((long)blockIdx.x * 2 + stream) * kWgradPart;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) part[(unsigned)(t * kTapFloats + (ct * 32 + mfma_row(r, h)) * 64 + cit * 32 + col)] = acc[t][r];
  if (cit == 0) {
    db += __shfl_xor(db, 32, 64);     // the two pixels of every k-step
    if (h == 0) part[9 * kTapFloats + ct * 32 + col] = db;
  }
}


