// Host side of bts_render_bwd (path choice, workspace) + the ROUND-1 lane = ray backward, kept for A/B in the probe build only
// (-DBTS_PROBE, env BTS_BWD_V1): the product serves every shape with the lane = sample passes of bts_bwd_rows.hip (plain MLP,
// K <= 64: gate bits) and bts_bwd_blocks.hip (ResnetBlockFC layers / long rays: rows).
//
// Round-1 kernel: backward of the fused renderer (projected-feature path) for gfx950.
//
// One lane = one ray, walked BACK TO FRONT so that the suffix sum the compositing gradient needs
//     g_alpha_k = g_w_k * T_k - (sum_{m>k} g_w_m w_m) / (1 - alpha_k + 1e-10)
// is a running register.  The forward saved two floats per sample (pre-softplus s_k and the transmittance T_k); everything else
// is recomputed: projection, taps, PE, h = bilinear(G) + W_pe . pe (+ blocks) on MFMA exactly as in the forward.  Then per sample
//     g_s   = g_alpha * |delta| * exp(-|delta| sigma) * sigmoid(s)
//     g_h   = relu'(h) * w_out * g_s                                   (kept in the MFMA C layout: rows hidden, columns points)
//     dG   += bilinear-weights * g_h     -> float atomics on the 4 taps (skipped for inactive hidden units)
//     dW_pe+= g_h . pe^T                 -> MFMA with the contraction over the wave's 64 points; both operands are transposed
//                                           through per-wave LDS tiles ([point][hidden], [point][pe input]); 4 persistent
//                                           32x32 accumulator tiles per wave, flushed once per work-group
//     dw_out += relu(h) g_s, db_out += g_s  -> per-lane partial sums, reduced across lanes once at the end.
// The feature-map / w_in[:, :C] gradients follow from dG in bts_prep.hip (per-pixel GEMMs).
//
// What torch.autograd would do for nerf.py:283-299 + models_bts.py:266-338 + resnetfc.py:132-184 of the reference.
#include "bts_bwd.h"
#include <cstdlib>

namespace bts {

#ifdef BTS_PROBE
template <int HD, int NB>
struct BwdLds {
  static constexpr int PE_ROWS = kPeDim + 1;            // 40
  static constexpr int LDG = HD + 1;                    // g_h tile leading dim  [64 points][HD] (+1: conflict-free column access)
  static constexpr int LDX = PE_ROWS + 1;               // pe tile leading dim   [64 points][40]
  static constexpr int W_PE = 0;                        // [40][HD] k-major (forward A operand)
  static constexpr int W_OUT = W_PE + PE_ROWS * HD;     // [HD]
  static constexpr int EMPTY = W_OUT + HD;              // [HD] projected empty feature
  static constexpr int D_EMPTY = EMPTY + HD;            // [HD] gradient accumulator for it
  static constexpr int D_WPE = D_EMPTY + HD;            // [40][HD] work-group accumulator for dW_pe (k-major like W_PE)
  static constexpr int D_WOUT = D_WPE + PE_ROWS * HD;   // [HD] + 1 (db_out)
  // ResnetBlockFC layers (RE10K): per block the forward operands (k-major: w0t [in][out], b0, w1t, b1 -- the layout hidden_layer
  // expects), the row-major weights for the backward products (w0 [out][in], w1) and the work-group gradient accumulators
  static constexpr int BLK_F = D_WOUT + HD + 1;
  static constexpr int BLK_F_STRIDE = 2 * HD * HD + 2 * HD;
  static constexpr int BLK_R = BLK_F + NB * BLK_F_STRIDE;          // per block: w0 [out][in], w1 [out][in]
  static constexpr int BLK_R_STRIDE = 2 * HD * HD;
  static constexpr int D_BLK = BLK_R + NB * BLK_R_STRIDE;          // per block: dw0 [out][in], db0 [HD], dw1 [out][in], db1 [HD]
  static constexpr int D_BLK_STRIDE = 2 * HD * HD + 2 * HD;
  static constexpr int TILES = D_BLK + NB * D_BLK_STRIDE;          // per wave: g tile, pe tile (, activation tile when NB > 0)
  static constexpr int TAP = 64 * LDG + 64 * LDX + (NB > 0 ? 64 * LDG : 0);  // per wave: [64 points][4 texel indices, 4 weights]
  static constexpr int TILE_STRIDE = ((TAP + 64 * 8 + 3) & ~3) + 4;
  static constexpr int TOTAL = TILES + 4 * TILE_STRIDE;
};

// dW[i][j] += sum over the wave's 64 points of A[p][i] * B[p][j], both operands staged as [point][channel] LDS tiles (k-step s pairs
// points s and s + 32); i, j < 32: one 32x32 accumulator tile
__device__ __forceinline__ void point_contraction(f32x16& dw, const float* a_tile, int lda, const float* b_tile, int ldb, int lane) {
  const int h = lane >> 5, col = lane & 31;
#pragma unroll 4
  for (int s = 0; s < 32; ++s) {
    const int pnt = s + 32 * h;
    dw = mfma(a_tile[pnt * lda + col], b_tile[pnt * ldb + col], dw);
  }
}

template <int C, int HD, int NB, int NVMAX>
__global__ __launch_bounds__(256, 1) void render_bwd_kernel(const BwdParams bp) {
  static_assert(NB == 0 || HD == 32, "ResnetBlockFC backward is built for d_hidden = 32 (RE10K config)");
  constexpr int HT = HD / 32;
  using L = BwdLds<HD, NB>;
  constexpr int D_IN = C + kPeDim;
  const FwdParams& p = bp.f;
  extern __shared__ float lds[];
  const MlpLayout ml{D_IN, HD, NB};

  // ---- stage: PE rows of w_in (k-major), w_out, projected empty feature; zero the work-group gradient accumulators
  for (int i = threadIdx.x; i < L::PE_ROWS * HD; i += blockDim.x) {
    const int k = i / HD, hid = i % HD;
    const int src = kernel_to_ref_input<C>(k + C);
    lds[L::W_PE + i] = src >= 0 ? p.mlp[ml.w_in() + hid * D_IN + src] : p.mlp[ml.b_in() + hid];
    lds[L::D_WPE + i] = 0.0f;
  }
  for (int hid = threadIdx.x; hid < HD; hid += blockDim.x) {
    lds[L::W_OUT + hid] = p.mlp[ml.w_out() + hid];
    float a = 0.0f;
    if (p.empty_feature)
      for (int c = 0; c < C; ++c) a = __builtin_fmaf(p.mlp[ml.w_in() + hid * D_IN + c], p.empty_feature[c], a);
    lds[L::EMPTY + hid] = a;
    lds[L::D_EMPTY + hid] = 0.0f;
    lds[L::D_WOUT + hid] = 0.0f;
  }
  if (threadIdx.x == 0) lds[L::D_WOUT + HD] = 0.0f;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    float* f = lds + L::BLK_F + b * L::BLK_F_STRIDE;
    float* r = lds + L::BLK_R + b * L::BLK_R_STRIDE;
    float* d = lds + L::D_BLK + b * L::D_BLK_STRIDE;
    for (int i = threadIdx.x; i < HD * HD; i += blockDim.x) {
      const int k = i / HD, o = i % HD;
      f[i] = p.mlp[ml.blk_w0(b) + o * HD + k];                      // k-major (transposed) for the forward recompute
      f[HD * HD + HD + i] = p.mlp[ml.blk_w1(b) + o * HD + k];
      r[i] = p.mlp[ml.blk_w0(b) + i];                               // row-major [out][in] for W^T . g
      r[HD * HD + i] = p.mlp[ml.blk_w1(b) + i];
    }
    for (int i = threadIdx.x; i < HD; i += blockDim.x) {
      f[HD * HD + i] = p.mlp[ml.blk_b0(b) + i];
      f[2 * HD * HD + HD + i] = p.mlp[ml.blk_b1(b) + i];
    }
    for (int i = threadIdx.x; i < L::D_BLK_STRIDE; i += blockDim.x) d[i] = 0.0f;
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int h = lane >> 5, col = lane & 31;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int sample = wg / p.tiles_per_sample;
  const int tile = wg - sample * p.tiles_per_sample;
  const int Bp = p.Bp, K = p.K;
  const int r_raw = tile * 256 + wave * 64 + lane;
  const bool active = r_raw < Bp;
  const int r = active ? r_raw : Bp - 1;
  const long ray = (long)sample * Bp + r;
  const int lane_off = h * HD + col;
  const int H = p.H, W = p.W, nv = p.nv;
  float* gh_tile = lds + L::TILES + wave * L::TILE_STRIDE;  // [64][LDG]
  float* pe_tile = gh_tile + 64 * L::LDG;                   // [64][LDX]
  float* act_tile = pe_tile + 64 * L::LDX;                  // [64][LDG]  (NB > 0 only)
  float* tap_tile = gh_tile + ((L::TAP + 3) & ~3);          // [64][8], 16-byte aligned rows

  const Cam enc = load_cam(p.w2c_enc + sample * 16, p.K_enc + sample * 9);
  const float4* __restrict__ G = reinterpret_cast<const float4*>(p.proj) + (long)sample * H * W * (HD / 4);
  float* __restrict__ dG = bp.d_proj ? bp.d_proj + (long)sample * H * W * HD : nullptr;

  const float4 r0 = reinterpret_cast<const float4*>(p.rays)[ray * 2];
  const float4 r1 = reinterpret_cast<const float4*>(p.rays)[ray * 2 + 1];
  const float ox = r0.x, oy = r0.y, oz = r0.z, dx = r0.w, dy = r1.x, dz = r1.y;
  const float* zrow = p.z_samp + ray * K;
  const float* srow = p.sigma_raw + ray * K;
  const float* trow = p.trans + ray * K;

  float g_rgb[NVMAX * 3];
#pragma unroll
  for (int i = 0; i < NVMAX * 3; ++i) g_rgb[i] = (bp.g_rgb && active && i < nv * 3) ? bp.g_rgb[ray * nv * 3 + i] : 0.0f;
  const float g_depth = (bp.g_depth && active) ? bp.g_depth[ray] : 0.0f;
  // white background (nerf.py:301-304): rgb = sum_k w_k c_k + 1 - sum_k w_k, so every weight also receives -sum_channels g_rgb
  float g_bkgd = 0.0f;
  if (p.white_bkgd) {
#pragma unroll
    for (int i = 0; i < NVMAX * 3; ++i) g_bkgd -= g_rgb[i];
  }

  // persistent per-wave gradient state
  f32x16 dwpe[HT][2];  // dW_pe^T tiles: rows hidden (ht), columns pe input (2 tiles of 32, 40 used)
  float dw2[HT][16];   // per-lane partial of dw_out for the accumulator rows this lane holds
#pragma unroll
  for (int ht = 0; ht < HT; ++ht) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) dwpe[ht][kt] = zero_acc();
#pragma unroll
    for (int q = 0; q < 16; ++q) dw2[ht][q] = 0.0f;
  }
  // ResnetBlockFC gradients (HD = 32): one 32x32 tile per weight matrix, per-lane bias partials like dw2
  f32x16 dwb[NB > 0 ? NB : 1][2];
  float dbb[NB > 0 ? NB : 1][2][16];
#pragma unroll
  for (int b = 0; b < (NB > 0 ? NB : 1); ++b)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      dwb[b][j] = zero_acc();
#pragma unroll
      for (int q = 0; q < 16; ++q) dbb[b][j][q] = 0.0f;
    }
  float db2 = 0.0f;
  float S = 0.0f;  // sum_{m>k} g_w_m w_m
  float z_after = 0.0f;

  for (int k = K - 1; k >= 0; --k) {
    const float z = zrow[k];
    const float s_raw = srow[k];
    const float T = trow[k];
    const float px = ox + z * dx, py = oy + z * dy, pz = oz + z * dz;

    // ---------------- colours of this sample -> g_w = sum_j g_rgb_j . c_kj + g_depth z_k (+ g_weights_k)
    float g_w = g_depth * z + g_bkgd;
    if (bp.g_weights && active) g_w += bp.g_weights[ray * K + k];
    if (p.rgb_samps) {
      // the forward's per-sample colours, when the caller kept them (training asks for rgb_samps anyway): 12 B per view instead of a
      // projection, four taps and a blend per view
      const float* cs = p.rgb_samps + (ray * K + k) * (long)(nv * 3);
#pragma unroll
      for (int j = 0; j < NVMAX; ++j)
        if (j < nv) g_w += g_rgb[3 * j] * cs[3 * j] + g_rgb[3 * j + 1] * cs[3 * j + 1] + g_rgb[3 * j + 2] * cs[3 * j + 2];
    } else
#pragma unroll
    for (int j = 0; j < NVMAX; ++j) {
      if (j < nv) {
        const Cam cj = load_cam(p.w2c_r + ((long)sample * nv + j) * 16, p.K_r + ((long)sample * nv + j) * 9);
        const Proj pc = project<false>(cj, px, py, pz);
        const Taps tc = make_taps(pc.x, pc.y, H, W);
        const float4* img = reinterpret_cast<const float4*>(p.imgs) + ((long)sample * nv + j) * H * W;
        const float4 a = img[tc.o00], b = img[tc.o01], cc = img[tc.o10], d = img[tc.o11];
        const float c0 = ((a.x * tc.w00 + b.x * tc.w01) + cc.x * tc.w10) + d.x * tc.w11;
        const float c1 = ((a.y * tc.w00 + b.y * tc.w01) + cc.y * tc.w10) + d.y * tc.w11;
        const float c2 = ((a.z * tc.w00 + b.z * tc.w01) + cc.z * tc.w10) + d.z * tc.w11;
        g_w += g_rgb[3 * j] * c0 + g_rgb[3 * j + 1] * c1 + g_rgb[3 * j + 2] * c2;
      }
    }

    // ---------------- encoder view
    const Proj pe = p.code_mode == 1 ? project<true>(enc, px, py, pz) : project<false>(enc, px, py, pz);
    const Taps tp = make_taps(pe.x, pe.y, H, W);
    float v3[3];
    v3[0] = pe.x, v3[1] = pe.y;
    v3[2] = depth_code(pe, p.code_mode == 1, p.inv_z != 0, p.inv_dmax, p.inv_range, p.d_min, p.range);
    const bool use_empty = (p.learn_empty != 0) & pe.invalid;

    // ---------------- compositing gradient (nerf.py:283-299)
    float sigma = softplus(s_raw);
    const bool dead = (p.empty_empty != 0) & pe.invalid;  // sigma forced to 0: no gradient
    if (dead) sigma = 0.0f;
    const bool last = (k == K - 1);
    const float delta = last ? 1e10f : (z_after - z);
    const float ex = transmittance(delta, sigma);
    const bool capped = (p.hard_cap != 0) & last;
    const float alpha = capped ? 1.0f : 1.0f - ex;
    const float wgt = alpha * T;
    float g_alpha = g_w * T - S / (capped ? 1e-10f : ex + 1e-10f);  // 1 - alpha + 1e-10 with 1 - alpha = exp(-|delta| sigma) un-rounded
    if (bp.g_alphas && active) g_alpha += bp.g_alphas[ray * K + k];
    S = S + g_w * wgt;
    z_after = z;
    float g_s = 0.0f;
    if (!capped && !dead && active) g_s = g_alpha * fabsf(delta) * ex * (s_raw > 20.0f ? 1.0f : sigmoidf(s_raw));

    // ---------------- recompute h (forward, PROJ path)
    f32x16 acc[HT][2];
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
      GBuf ga, gb;
      gload<HD>(ga, G, o[0], 0, 4 * h);
      gather_seq<HD, 0>(acc, ga, gb, G, o, wq, h);
      if (p.learn_empty && __any(use_empty)) apply_empty<HD>(acc, emp, lds + L::EMPTY, h);
    }
    // PE inputs of this lane's point -> LDS tile row (for the dW_pe contraction) and through the MFMAs
    float* my_pe = pe_tile + lane * L::LDX;
    const float* wl = lds + L::W_PE + lane_off;
    my_pe[0] = v3[0], my_pe[1] = v3[1], my_pe[2] = v3[2], my_pe[3] = 1.0f;
    kstep<HD>(acc, wl, 0, v3[0], v3[1]);
    kstep<HD>(acc, wl + 2 * HD, 0, v3[2], 1.0f);
    wl += 4 * HD;
    {
      float sc[6], sn[6];
      pe_octave(sc, v3, p.freq_factor);
      float ff = p.freq_factor;
#pragma unroll 1
      for (int oct = 0; oct < kNumFreqs; ++oct) {
        ff = ff * 2.0f;
        if (oct + 1 < kNumFreqs) pe_octave(sn, v3, ff);
#pragma unroll
        for (int i = 0; i < 6; ++i) my_pe[4 + 6 * oct + i] = sc[i];
        kstep<HD>(acc, wl, 0, sc[0], sc[1]);
        kstep<HD>(acc, wl + 2 * HD, 0, sc[2], sc[3]);
        kstep<HD>(acc, wl + 4 * HD, 0, sc[4], sc[5]);
        wl += 6 * HD;
#pragma unroll
        for (int i = 0; i < 6; ++i) sc[i] = sn[i];
      }
    }

    // ---------------- ResnetBlockFC layers, forward recompute: keep every block's input h and its inner activation net
    f32x16 h_in[NB > 0 ? NB : 1][HT][2], net_a[NB > 0 ? NB : 1][HT][2];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float* base = lds + L::BLK_F + b * L::BLK_F_STRIDE;
#pragma unroll
      for (int ot = 0; ot < HT; ++ot)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) h_in[b][ot][pt] = acc[ot][pt];
#pragma unroll
      for (int ot = 0; ot < HT; ++ot)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float bias = base[HD * HD + ot * 32 + mfma_row(q, 0) + 4 * h];
          net_a[b][ot][0][q] = bias, net_a[b][ot][1][q] = bias;
        }
      hidden_layer<HD>(net_a[b], acc, base, lane);
#pragma unroll
      for (int ot = 0; ot < HT; ++ot)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float bias = base[2 * HD * HD + HD + ot * 32 + mfma_row(q, 0) + 4 * h];
          acc[ot][0][q] += bias, acc[ot][1][q] += bias;
        }
      hidden_layer<HD>(acc, net_a[b], base + HD * HD + HD, lane);
    }

    // ---------------- g_h (of the LAST layer's output) in the C layout; dw_out / db_out partials
    float gs_t[2];
    {
      unsigned t0, t1;
      bcast_tiles(__float_as_uint(g_s), t0, t1);
      gs_t[0] = __uint_as_float(t0), gs_t[1] = __uint_as_float(t1);
    }
    db2 += g_s;
#pragma unroll
    for (int ht = 0; ht < HT; ++ht)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int hid = ht * 32 + mfma_row(q, 0) + 4 * h;
        const float w2 = lds[L::W_OUT + hid];
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
          const float hv = acc[ht][pt][q];
          dw2[ht][q] = __builtin_fmaf(fmaxf(hv, 0.0f), gs_t[pt], dw2[ht][q]);
          acc[ht][pt][q] = hv > 0.0f ? w2 * gs_t[pt] : 0.0f;   // acc now holds g_h
        }
      }

    // ---------------- back through the blocks (resnetfc.py:53-62): h' = h + fc_1(relu(fc_0(relu(h))))
    //   g_dx = g_h';  dW1 += g_dx (x) relu(net);  g_net = relu'(net) . (W1^T g_dx);  dW0 += g_net (x) relu(h);  g_h = g_h' + relu'(h) . (W0^T g_net)
#pragma unroll
    for (int b = NB - 1; b >= 0; --b) {
      const float* rw = lds + L::BLK_R + b * L::BLK_R_STRIDE;
      if (bp.d_mlp) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) {
            const int hid = mfma_row(q, 0) + 4 * h;
            gh_tile[(pt * 32 + col) * L::LDG + hid] = acc[0][pt][q];
            act_tile[(pt * 32 + col) * L::LDG + hid] = fmaxf(net_a[b][0][pt][q], 0.0f);
            dbb[b][1][q] += acc[0][pt][q];
          }
        point_contraction(dwb[b][1], gh_tile, L::LDG, act_tile, L::LDG, lane);   // dW1[out][in]
      }
      f32x16 g_net[HT][2];
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) g_net[0][pt] = zero_acc();
      hidden_layer<HD, false>(g_net, acc, rw + HD * HD, lane);                   // W1^T . g_dx
#pragma unroll
      for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) g_net[0][pt][q] = net_a[b][0][pt][q] > 0.0f ? g_net[0][pt][q] : 0.0f;
      if (bp.d_mlp) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) {
            const int hid = mfma_row(q, 0) + 4 * h;
            gh_tile[(pt * 32 + col) * L::LDG + hid] = g_net[0][pt][q];
            act_tile[(pt * 32 + col) * L::LDG + hid] = fmaxf(h_in[b][0][pt][q], 0.0f);
            dbb[b][0][q] += g_net[0][pt][q];
          }
        point_contraction(dwb[b][0], gh_tile, L::LDG, act_tile, L::LDG, lane);   // dW0[out][in]
      }
      f32x16 g_rh[HT][2];
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) g_rh[0][pt] = zero_acc();
      hidden_layer<HD, false>(g_rh, g_net, rw, lane);                            // W0^T . g_net
#pragma unroll
      for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) acc[0][pt][q] += h_in[b][0][pt][q] > 0.0f ? g_rh[0][pt][q] : 0.0f;
    }

    // ---------------- transposed copy of g_h (of lin_in's output) for the dW_pe MFMA
#pragma unroll
    for (int ht = 0; ht < HT; ++ht)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int hid = ht * 32 + mfma_row(q, 0) + 4 * h;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) gh_tile[(pt * 32 + col) * L::LDG + hid] = acc[ht][pt][q];
      }

    // ---------------- dG += w_tap * g_h on the four taps.  lane = CHANNEL: for every point p of the wave the 64 lanes add its 64
    // hidden gradients to the 256 contiguous bytes of a texel row -- 2 cache-line requests per atomic instruction instead of 32
    // (with lane = point, one dword per line: 27 of the 28 ms of the first backward went into those L2 atomics).  g_h comes from
    // the [point][hidden] tile that also feeds the dW_pe contraction; taps and weights of every point from a small LDS table.
    if (dG && bp.gh_ws) {
      // two-pass form: the rows go to the workspace ([group of 64 rays][k][ray][channel], 256-byte coalesced stores); the scatter
      // kernel below merges them per texel in LDS before anything reaches the L2 atomic units
      if (p.learn_empty) tap_tile[lane * 8 + 4] = use_empty ? -1.0f : 1.0f;
      constexpr int PPI = 64 / HD;
      const int ch = lane % HD;
      const int hid_l = proj_hidden_of_storage(ch);
      const long grp = (long)wg * 4 + wave;
      float* wrow = bp.gh_ws + ((grp * K + k) * 64) * HD;
#pragma unroll 4
      for (int pnt0 = 0; pnt0 < 64; pnt0 += PPI) {
        const int pnt = pnt0 + lane / HD;
        float gv = gh_tile[pnt * L::LDG + hid_l];
        if (p.learn_empty && tap_tile[pnt * 8 + 4] < 0.0f) {   // the point took the (projected) empty feature
          if (gv != 0.0f) atomicAdd(&lds[L::D_EMPTY + hid_l], gv);
          gv = 0.0f;
        }
        wrow[pnt * HD + ch] = gv;
      }
    } else if (dG) {
      {
        float* row = tap_tile + lane * 8;
        reinterpret_cast<int4*>(row)[0] = make_int4(tp.o00, tp.o01, tp.o10, tp.o11);
        reinterpret_cast<float4*>(row)[1] = use_empty ? make_float4(-1.0f, 0.0f, 0.0f, 0.0f) : make_float4(tp.w00, tp.w01, tp.w10, tp.w11);
      }
      constexpr int PPI = 64 / HD;                      // points per atomic instruction (2 when d_hidden = 32)
      const int ch = lane % HD;                          // storage channel of this lane
      const int hid_l = proj_hidden_of_storage(ch);      // ... which holds this hidden unit
#pragma unroll 2
      for (int pnt0 = 0; pnt0 < 64; pnt0 += PPI) {
        const int pnt = pnt0 + lane / HD;
        const float gv = gh_tile[pnt * L::LDG + hid_l];
        const int4 oo = reinterpret_cast<const int4*>(tap_tile + pnt * 8)[0];
        const float4 ww = reinterpret_cast<const float4*>(tap_tile + pnt * 8)[1];
        if (gv != 0.0f) {
          if (ww.x < 0.0f) {   // learn_empty: the point took the (projected) empty feature
            atomicAdd(&lds[L::D_EMPTY + hid_l], gv);
          } else {
            atomic_add_f32(dG + (long)oo.x * HD + ch, ww.x * gv);
            atomic_add_f32(dG + (long)oo.y * HD + ch, ww.y * gv);
            atomic_add_f32(dG + (long)oo.z * HD + ch, ww.z * gv);
            atomic_add_f32(dG + (long)oo.w * HD + ch, ww.w * gv);
          }
        }
      }
    }

    // ---------------- dW_pe^T[hid][kin] += sum_points g_h[hid][p] * pe[p][kin]   (k-step s pairs points s and s + 32)
    if (bp.d_mlp) {
#pragma unroll 4
      for (int s = 0; s < 32; ++s) {
        const int pnt = s + 32 * h;
        float a[HT], b[2];
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) a[ht] = gh_tile[pnt * L::LDG + ht * 32 + col];
        b[0] = pe_tile[pnt * L::LDX + col];
        b[1] = col < L::PE_ROWS - 32 ? pe_tile[pnt * L::LDX + 32 + col] : 0.0f;
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) {
          dwpe[ht][0] = mfma(a[ht], b[0], dwpe[ht][0]);
          dwpe[ht][1] = mfma(a[ht], b[1], dwpe[ht][1]);
        }
      }
    }
  }

  // ---------------- flush: per-wave registers -> work-group LDS accumulators -> one global atomic per parameter
  if (bp.d_mlp) {
#pragma unroll
    for (int ht = 0; ht < HT; ++ht) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int hid = ht * 32 + mfma_row(q, h), kin = kt * 32 + col;
          if (kin < L::PE_ROWS) atomicAdd(&lds[L::D_WPE + kin * HD + hid], dwpe[ht][kt][q]);
        }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        // reduce the 32 point-columns of each lane half, then one LDS atomic per hidden unit
        float v = dw2[ht][q];
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (col == 0) atomicAdd(&lds[L::D_WOUT + ht * 32 + mfma_row(q, h)], v);
      }
    }
    float v = db2;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) atomicAdd(&lds[L::D_WOUT + HD], v);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float* d = lds + L::D_BLK + b * L::D_BLK_STRIDE;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float* dw = d + j * (HD * HD + HD);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          atomicAdd(&dw[mfma_row(q, h) * HD + col], dwb[b][j][q]);   // D[i = out][j = in]
          float bv = dbb[b][j][q];
#pragma unroll
          for (int off = 16; off >= 1; off >>= 1) bv += __shfl_xor(bv, off, 64);
          if (col == 0) atomicAdd(&dw[HD * HD + mfma_row(q, h)], bv);
        }
      }
    }
  }
  __syncthreads();
  if (bp.d_mlp) {
    for (int i = threadIdx.x; i < L::PE_ROWS * HD; i += blockDim.x) {
      const int k = i / HD, hid = i % HD;
      const int src = kernel_to_ref_input<C>(k + C);
      const float v = lds[L::D_WPE + i];
      if (v != 0.0f) atomic_add_f32(bp.d_mlp + (src >= 0 ? ml.w_in() + hid * D_IN + src : ml.b_in() + hid), v);
    }
    for (int i = threadIdx.x; i <= HD; i += blockDim.x) {
      const float v = lds[L::D_WOUT + i];
      if (v != 0.0f) atomic_add_f32(bp.d_mlp + (i < HD ? ml.w_out() + i : ml.b_out()), v);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float* d = lds + L::D_BLK + b * L::D_BLK_STRIDE;
      for (int i = threadIdx.x; i < L::D_BLK_STRIDE; i += blockDim.x) {
        const float v = d[i];
        if (v != 0.0f) atomic_add_f32(bp.d_mlp + ml.blk(b) + i, v);   // packed order: w0, b0, w1, b1 = the LDS order
      }
    }
  }
  if (bp.d_empty_proj) {
    for (int i = threadIdx.x; i < HD; i += blockDim.x) {
      const float v = lds[L::D_EMPTY + i];
      if (v != 0.0f) atomic_add_f32(bp.d_empty_proj + i, v);
    }
  }
}

// ---- dG scatter pass.  One wave per group of 64 rays (one 8x8 patch under PatchRaySampler).  Neighbouring rays and consecutive
// samples of a patch land on the same few texels of G: the 4 taps x 64 rays of one step cover ~9x9 texels, and the footprint drifts by
// about a pixel per step.  The wave therefore keeps a CW x CH texel window of dG rows in LDS (slot = (y mod CH, x mod CW): a texel
// keeps its slot while the window slides), adds tap contributions there with ds_add, and only rows LEAVING the window go to global
// memory as one 256-byte row of float atomics.  ~5-10 % of the tap updates remain as L2 atomics (measured by simulation on the
// KITTI-360 training geometry and on the GPU).  A step whose footprint does not fit the window falls back to direct row atomics.
// lane = channel throughout; the taps of the step are computed lane = ray and broadcast with v_readlane.
struct ScatterParams {
  FwdParams f;
  const float* gh_ws;
  float* d_proj;
  int groups_per_sample;
  int mode;   // probe builds only (-DBTS_PROBE, env BTS_SCATTER_MODE): 1 no LDS adds, 2 never move the window, 4 no workspace loads,
              // 8 skip non-fitting steps -- how the pass was taken apart in profiles/README.md; always 0 in the product
};

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
  return __builtin_amdgcn_readfirstlane(v);
}

#ifdef BTS_PROBE
#define BTS_SCATTER_ABL(bit) ((sp.mode & (bit)) != 0)
#else
#define BTS_SCATTER_ABL(bit) false
#endif

template <int HD>
__global__ __launch_bounds__(64) void scatter_dg_kernel(const ScatterParams sp) {
#ifndef BTS_SCATTER_RB
#define BTS_SCATTER_RB 64
#endif
  constexpr int CW = 12, CH = 12, RB = BTS_SCATTER_RB;  // window of texels (9x9 footprint of a patch + 3 of drift); g_h rows per register block
  __shared__ float cache[(CW * CH + 1) * HD];   // + one scratch row: the target of clamped (duplicate, zero-weight) taps
  const FwdParams& p = sp.f;
  const int lane = threadIdx.x;
  const int grp = blockIdx.x;
  const int sample = grp / sp.groups_per_sample;
  const int g_in = grp - sample * sp.groups_per_sample;
  const int Bp = p.Bp, K = p.K, H = p.H, W = p.W;
  const int r_raw = g_in * 64 + lane;
  const int r = r_raw < Bp ? r_raw : Bp - 1;
  const long ray = (long)sample * Bp + r;
  const bool chan = HD == 64 || lane < HD;
  const int ch = lane % HD;
  for (int i = lane; i < (CW * CH + 1) * HD; i += 64) cache[i] = 0.0f;
  const Cam enc = load_cam(p.w2c_enc + sample * 16, p.K_enc + sample * 9);
  const float4 r0 = reinterpret_cast<const float4*>(p.rays)[ray * 2];
  const float4 r1 = reinterpret_cast<const float4*>(p.rays)[ray * 2 + 1];
  const float* zrow = p.z_samp + ray * K;
  float* __restrict__ dG = sp.d_proj + (long)sample * H * W * HD;
  const float* __restrict__ ws = sp.gh_ws + (long)grp * K * 64 * HD;
  int wx = 0, wy = 0;   // window origin (uniform)

  // evict every slot whose texel lies outside the window at (nwx, nwy)
  auto flush = [&](int nwx, int nwy, bool all) {
    const int wxm = ((wx % CW) + CW) % CW, wym = ((wy % CH) + CH) % CH;   // slot column / row of the window origin
    for (int sy = 0; sy < CH; ++sy) {
      const int ty = wy + sy - wym + (sy < wym ? CH : 0);
      const bool row_out = all || ty < nwy || ty >= nwy + CH;
      for (int sx = 0; sx < CW; ++sx) {
        const int tx = wx + sx - wxm + (sx < wxm ? CW : 0);
        if (row_out || tx < nwx || tx >= nwx + CW) {
          if (chan) {
            float* c = &cache[(sy * CW + sx) * HD + ch];
            const float v = *c;
            if (v != 0.0f) {
              atomic_add_f32(dG + ((long)ty * W + tx) * HD + ch, v);
              *c = 0.0f;
            }
          }
        }
      }
    }
  };

  float cur[RB], nxt[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) cur[i] = chan ? ws[((long)(K - 1) * 64 + i) * HD + ch] : 0.0f;

  for (int k = K - 1; k >= 0; --k) {
    const float z = zrow[k];
    const Proj pe = project<false>(enc, r0.x + z * r0.w, r0.y + z * r1.x, r0.z + z * r1.y);
    int x0, y0, x1, y1;
    const Taps tp = make_taps_xy(pe.x, pe.y, H, W, x0, y0, x1, y1);
    const int mnx = wave_min(x0), mxx = -wave_min(-x1), mny = wave_min(y0), mxy = -wave_min(-y1);
    const bool fits = ((mxx - mnx < CW) && (mxy - mny < CH)) || BTS_SCATTER_ABL(2);
    if (fits && !BTS_SCATTER_ABL(2) && (mnx < wx || mxx >= wx + CW || mny < wy || mxy >= wy + CH)) {
      const int nwx = mnx - (CW - (mxx - mnx + 1)) / 2, nwy = mny - (CH - (mxy - mny + 1)) / 2;
      flush(nwx, nwy, false);
      wx = nwx, wy = nwy;
    }
    // LDS slot (float index of the row) of each tap of this lane's point.  A clamped tap (x1 == x0 or y1 == y0 at the far border:
    // weight exactly 0) would alias its neighbour's slot inside one read-modify-write group; it goes to the scratch row instead.
    const int rya = (int)((unsigned)y0 % CH) * CW, ryb = (int)((unsigned)y1 % CH) * CW;
    const int cxa = (int)((unsigned)x0 % CW), cxb = (int)((unsigned)x1 % CW);
    const bool ddx = x1 != x0, ddy = y1 != y0;
    const int s00 = (rya + cxa) * HD;
    const int s01 = ddx ? (rya + cxb) * HD : CW * CH * HD;
    const int s10 = ddy ? (ryb + cxa) * HD : CW * CH * HD;
    const int s11 = (ddx && ddy) ? (ryb + cxb) * HD : CW * CH * HD;
    const float* wk = ws + (long)k * 64 * HD;
#pragma unroll 1
    for (int b = 0; b < 64 / RB; ++b) {
      // prefetch the next block of rows (next step's first block after the last one of this step)
      {
        const bool more = b + 1 < 64 / RB || k > 0;
        const float* nb = b + 1 < 64 / RB ? wk + (long)(b + 1) * RB * HD : wk - (long)64 * HD;
#pragma unroll
        for (int i = 0; i < RB; ++i) nxt[i] = (chan && more && !BTS_SCATTER_ABL(4)) ? nb[i * HD + ch] : 0.0f;
      }
      auto bc_i = [&](int v, int pnt) { return __builtin_amdgcn_readlane(v, pnt); };
      auto bc_f = [&](float v, int pnt) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), pnt)); };
      // NOTE: every v_readlane below sits in wave-uniform control flow.  Reading a lane that is inactive at the readlane is
      // undefined in LLVM's model (the producer may be sunk into the divergent region): with d_hidden = 32 an `if (chan)` around
      // this block made lanes >= 32 inactive and points 32..63 picked up stale registers.  Idle lanes aim at the scratch row.
      if (fits) {
        if (!BTS_SCATTER_ABL(1)) {
          // wave-private read-modify-write (LDS operations of one wave execute in order).  ds_add_f32 would be one instruction
          // per tap but runs at ~100 cycles per wave instruction on gfx950 (measured); plain loads and stores do not.
#pragma unroll
          for (int i = 0; i < RB; ++i) {
            const int pnt = b * RB + i;
            const float gv = cur[i];
            float* c00 = &cache[(chan ? bc_i(s00, pnt) : CW * CH * HD) + ch];
            float* c01 = &cache[(chan ? bc_i(s01, pnt) : CW * CH * HD) + ch];
            float* c10 = &cache[(chan ? bc_i(s10, pnt) : CW * CH * HD) + ch];
            float* c11 = &cache[(chan ? bc_i(s11, pnt) : CW * CH * HD) + ch];
            const float a00 = *c00, a01 = *c01, a10 = *c10, a11 = *c11;
            *c00 = a00 + bc_f(tp.w00, pnt) * gv;
            *c01 = a01 + bc_f(tp.w01, pnt) * gv;
            *c10 = a10 + bc_f(tp.w10, pnt) * gv;
            *c11 = a11 + bc_f(tp.w11, pnt) * gv;
          }
        }
      } else if (!BTS_SCATTER_ABL(8)) {
        // rare (a footprint wider than the window: rays nearly through the encoder's centre): every tap a row of L2 atomics.
        // The rows are re-read from the workspace so that the register block is never indexed dynamically.
#pragma unroll 1
        for (int pnt = b * RB; pnt < (b + 1) * RB; ++pnt) {
          const float gv = chan ? wk[pnt * HD + ch] : 0.0f;
          const long ya = (long)bc_i(y0, pnt) * W, yb = (long)bc_i(y1, pnt) * W;
          const int xa = bc_i(x0, pnt), xb = bc_i(x1, pnt);
          const float w00 = bc_f(tp.w00, pnt), w01 = bc_f(tp.w01, pnt), w10 = bc_f(tp.w10, pnt), w11 = bc_f(tp.w11, pnt);
          if (gv != 0.0f) {
            atomic_add_f32(dG + (ya + xa) * HD + ch, w00 * gv);
            atomic_add_f32(dG + (ya + xb) * HD + ch, w01 * gv);
            atomic_add_f32(dG + (yb + xa) * HD + ch, w10 * gv);
            atomic_add_f32(dG + (yb + xb) * HD + ch, w11 * gv);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < RB; ++i) cur[i] = nxt[i];
    }
  }
  flush(0, 0, true);
}

template <int C, int HD, int NB>
static int launch_bwd(const BwdParams& bp, int grid, hipStream_t s) {
  using L = BwdLds<HD, NB>;
  const size_t shmem = L::TOTAL * sizeof(float);
  auto go = [&](auto kern) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    kern<<<grid, 256, shmem, s>>>(bp);
  };
  if (bp.f.nv <= 1) go(render_bwd_kernel<C, HD, NB, 1>);
  else if (bp.f.nv <= 2) go(render_bwd_kernel<C, HD, NB, 2>);
  else if (bp.f.nv <= 4) go(render_bwd_kernel<C, HD, NB, 4>);
  else go(render_bwd_kernel<C, HD, NB, 8>);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: backward kernel launch failed (%ld)", hipGetErrorString(e), (long)e);
    return BTS_E_LAUNCH;
  }
  return BTS_OK;
}

#endif  // BTS_PROBE

FwdParams make_params(const BtsFieldCfg* cfg, const BtsFieldTensors* t);
int render_grid(const FwdParams& p);
int render_chunk_log2(int grid, long groups);
int launch_bwd_rows(const BwdParams& bp, int C, int HD, int n, int grid, hipStream_t s);
int launch_bwd_blocks(const BwdParams& bp, float* u0_ws, int C, int HD, int NB, int n, int grid, hipStream_t s);

// plain MLP and at most one wave of samples per ray: the gate-bit passes of bts_bwd_rows.hip; ResnetBlockFC layers (RE10K) and K > 64:
// the row passes of bts_bwd_blocks.hip
static bool bits_path(const BtsFieldCfg* cfg, const BtsRenderArgs* a) { return cfg->n_blocks == 0 && a->K <= 64; }

// workspace = what the passes hand each other per sample.  Gate-bit path: g_s (one float) + the relu gates as bits, once per sample
// and once per channel -- 20 bytes at d_hidden 64.  Row path: the gradient row at lin_in's output (4 d_hidden bytes) + g_s.
// (Round 1: 256-byte g_h rows per sample AND lane = ray; probe build only.)
// + pass C's slot copies of dW_pe (bts_bwd.h: kFlushSlots x 40 x d_hidden floats, 80 KB at d_hidden 64), behind the per-sample part
static size_t flush_bytes(const BtsFieldCfg* cfg) { return sizeof(float) * kFlushSlots * kFlushRows * (size_t)cfg->d_hidden; }
static size_t align16(size_t b) { return (b + 15) & ~(size_t)15; }
size_t render_bwd_workspace_impl(const BtsFieldCfg* cfg, const BtsRenderArgs* a) {
  const size_t rays = (size_t)cfg->n * (size_t)a->rays_per_sample;
  // gate-bit path: g_s + the per-sample masks, rounded up to 8 bytes (the per-channel masks behind them are read as 64-bit words)
  const size_t bits = align16(((rays * (size_t)a->K * (1 + (size_t)cfg->d_hidden / 32) + 1) & ~(size_t)1) * sizeof(float) +
                              rays * 2 * (size_t)cfg->d_hidden * sizeof(float)) + flush_bytes(cfg);
  const size_t rows = align16(rays * (size_t)a->K * ((size_t)cfg->d_hidden + 1) * sizeof(float)) + flush_bytes(cfg);
#ifdef BTS_PROBE   // any path may serve the call (BTS_BWD_V1)
  const size_t groups = (size_t)cfg->n * ((a->rays_per_sample + 255) / 256) * 4;
  const size_t v1 = groups * (size_t)a->K * 64 * (size_t)cfg->d_hidden * sizeof(float);
  return v1 > rows ? (v1 > bits ? v1 : bits) : (rows > bits ? rows : bits);
#else
  return bits_path(cfg, a) ? bits : rows;
#endif
}

#ifdef BTS_PROBE
template <int HD>
static int launch_scatter(const BwdParams& bp, int n, hipStream_t s) {
  ScatterParams sp;
  sp.f = bp.f, sp.gh_ws = bp.gh_ws, sp.d_proj = bp.d_proj, sp.groups_per_sample = bp.f.tiles_per_sample * 4;
  sp.mode = 0;
  if (const char* e = getenv("BTS_SCATTER_MODE")) sp.mode = atoi(e);
  scatter_dg_kernel<HD><<<n * sp.groups_per_sample, 64, 0, s>>>(sp);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: dG scatter kernel launch failed (%ld)", hipGetErrorString(e), (long)e);
    return BTS_E_LAUNCH;
  }
  return BTS_OK;
}
#endif

// where pass C's slot copies sit inside a workspace (the tail of either layout), for a caller that zeroes them itself
void render_bwd_flush_region(const BtsFieldCfg* cfg, const BtsRenderArgs* a, void* workspace, float** ptr, size_t* bytes) {
  const size_t total = render_bwd_workspace_impl(cfg, a);
  *bytes = flush_bytes(cfg);
  *ptr = reinterpret_cast<float*>(static_cast<char*>(workspace) + (total - *bytes));
}

int render_bwd_impl(const BtsFieldCfg* cfg, const BtsFieldTensors* t, const BtsRenderArgs* a, const BtsRenderGrads* g, void* workspace,
                    size_t, hipStream_t s, bool flush_clean) {
  BwdParams bp;
#ifdef BTS_PROBE
  // the probe build sizes the workspace as the maximum of three layouts, so render_bwd_flush_region's "tail of the workspace" is not where
  // the layout taken below puts pass C's slot copies: never trust a caller-side clear here (the passes zero the slots themselves)
  flush_clean = false;
#endif
  bp.flush_clean = flush_clean;
  bp.f = make_params(cfg, t);
  bp.f.rays = a->rays, bp.f.z_samp = a->z_samp;
  bp.f.Bp = a->rays_per_sample, bp.f.K = a->K, bp.f.hard_cap = a->hard_alpha_cap, bp.f.white_bkgd = a->white_bkgd;
  bp.f.sigma_raw = a->sigma_raw, bp.f.trans = a->trans, bp.f.sigma_noise = a->sigma_noise;
  bp.f.rgb_samps = a->rgb_samps;   // optional INPUT here: the forward's per-sample colours (else they are recomputed)
  bp.f.tiles_per_sample = (a->rays_per_sample + 255) / 256;
  bp.g_rgb = g->g_rgb, bp.g_depth = g->g_depth, bp.g_weights = g->g_weights, bp.g_alphas = g->g_alphas;
  bp.d_proj = g->d_proj_nhwc, bp.d_mlp = g->d_mlp_params, bp.d_empty_proj = g->d_empty_proj;
  bp.gh_ws = nullptr, bp.gs_ws = nullptr, bp.mask_ws = nullptr, bp.pmask_ws = nullptr, bp.flush_ws = nullptr;
#ifdef BTS_TICKS
  bp.ticks = nullptr;
  if (const char* e = getenv("BTS_DBG_PTR")) bp.ticks = (unsigned long long*)strtoull(e, nullptr, 0);   // diagnostic build only
#endif
  bp.tiles = bp.d_proj ? g->d_proj_tiles : nullptr;
  bp.tiles_per_img = (int)((((long)(cfg->H >> cfg->feat_shift) * (cfg->W >> cfg->feat_shift)) + 63) / 64);
  bp.tile_tw = tile_cols(cfg->H >> cfg->feat_shift, cfg->W >> cfg->feat_shift, cfg->tile_blocks);
#ifdef BTS_PROBE
  static const bool direct = getenv("BTS_BWD_DIRECT_ATOMICS") != nullptr;   // A/B (probe build): round-1 kernel, every tap update an L2 atomic
  static const bool v1 = getenv("BTS_BWD_V1") != nullptr || direct;         // A/B (probe build): the round-1 lane = ray pass for every shape
  static const bool rows_always = getenv("BTS_BWD_ROWS") != nullptr;        // A/B (probe build): the row passes for every shape
  if (v1 && !bp.f.fs && !bp.tiles) {   // (the round-1 pass knows full-size maps only, and no tile flags)
    bp.gh_ws = (bp.d_proj && !direct) ? static_cast<float*>(workspace) : nullptr;
    const int grid = bp.f.tiles_per_sample * cfg->n;
    int rc = BTS_E_UNSUPPORTED;
    if (cfg->C == 64 && cfg->d_hidden == 64 && cfg->n_blocks == 0) rc = launch_bwd<64, 64, 0>(bp, grid, s);
    else if (cfg->C == 32 && cfg->d_hidden == 32 && cfg->n_blocks == 0) rc = launch_bwd<32, 32, 0>(bp, grid, s);
    else if (cfg->C == 32 && cfg->d_hidden == 32 && cfg->n_blocks == 1) rc = launch_bwd<32, 32, 1>(bp, grid, s);
    if (rc == BTS_OK && bp.gh_ws) rc = cfg->d_hidden == 64 ? launch_scatter<64>(bp, cfg->n, s) : launch_scatter<32>(bp, cfg->n, s);
    return rc;
  }
#else
  constexpr bool rows_always = false;
#endif
  bp.f.lpr = 64, bp.f.groups = (long)cfg->n * a->rays_per_sample;
  if (bp.f.groups > 0x7FF00000L) {
    set_error("%s: too many rays in one call (%ld)", "bts_render_bwd", bp.f.groups);
    return BTS_E_UNSUPPORTED;
  }
  // the RE10K model at its own sample count (32 < K <= 48, exp_re10k.yaml: 48): four rays in three wave iterations (rowsb_kernel<PK>)
  const bool pack48 = !(bits_path(cfg, a) && !rows_always) && cfg->C == 32 && cfg->d_hidden == 32 && cfg->n_blocks == 1 && a->K > 32 &&
                      a->K <= 48 && a->rays_per_sample % 4 == 0;
  if (pack48) bp.f.lpr = 48, bp.f.groups /= 4;
  const int grid = render_grid(bp.f);
  bp.f.chunk_log2 = render_chunk_log2(grid, bp.f.groups);
  const size_t samples = (size_t)cfg->n * a->rays_per_sample * a->K;
  int rc;
  if (bits_path(cfg, a) && !rows_always) {
    bp.gs_ws = static_cast<float*>(workspace);
    bp.mask_ws = reinterpret_cast<unsigned*>(bp.gs_ws + samples);
    const size_t head = (samples * (1 + (size_t)cfg->d_hidden / 32) + 1) & ~(size_t)1;   // dwords, even: the 64-bit masks stay 8-byte aligned
    bp.pmask_ws = reinterpret_cast<uint2*>(bp.gs_ws + head);
    bp.flush_ws = reinterpret_cast<float*>(static_cast<char*>(workspace) +
                                           align16((head + (size_t)cfg->n * a->rays_per_sample * 2 * (size_t)cfg->d_hidden) * sizeof(float)));
    rc = launch_bwd_rows(bp, cfg->C, cfg->d_hidden, cfg->n, grid, s);
  } else {
    float* u0_ws = static_cast<float*>(workspace);          // (rays, K, d_hidden): rows first, they are read as 16-byte pieces
    bp.gs_ws = u0_ws + samples * (size_t)cfg->d_hidden;     // (rays, K)
    bp.flush_ws = reinterpret_cast<float*>(static_cast<char*>(workspace) + align16(samples * ((size_t)cfg->d_hidden + 1) * sizeof(float)));
    rc = launch_bwd_blocks(bp, u0_ws, cfg->C, cfg->d_hidden, cfg->n_blocks, cfg->n, grid, s);
  }
  if (rc != BTS_E_UNSUPPORTED) return rc;
  set_error("%s: unsupported MLP shape C=%ld d_hidden=%ld n_blocks=%ld", "bts_render_bwd", cfg->C, cfg->d_hidden, cfg->n_blocks);
  return rc;
}

}  // namespace bts
