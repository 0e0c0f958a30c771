// bts_loss.hip -- the photometric loss on the renderer's patch outputs, forward and backward in one pass.
//
// Reference: ReconstructionLoss.__call__ (models/bts/model/loss.py:83-293) with criterion "l1+ssim" (compute_errors_l1ssim :10-18,
// SSIM models/common/model/layers.py:79-150: 3x3 Gaussian window, zero padding, comp_mode), minimum over the render views, the
// invalid policies strict / weight_guided / none, and edge_aware_smoothness (:21-40).  The reference runs ~60 small kernels and
// nine .item() synchronisations per step for this; everything here is local to one patch, so one wave takes one patch
// (lane = pixel, ph*pw <= 64) and produces the patch's partial sums AND d loss / d rgb, d loss / d depth directly:
//   rgb term:  L_q = keep_q * min_v e_v(q),   e_v = 0.85 mean_c ssim_c + 0.15 mean_c |x_c - y_c|
//   ssim_c(q) = clamp(1 - n/d, 0, 1) / 2,  n = (2 mu_x mu_y + c1)(2 s_xy + c2),  d = (mu_x^2 + mu_y^2 + c1)(s_xx + s_yy + c2)
//   with mu = G * x, s_xx = G * x^2 - mu_x^2, s_xy = G * xy - mu_x mu_y over the zero-padded 3x3 neighbourhood.
// Backward of the SSIM term: every pixel q turns its upstream gradient into three coefficients (of d mu_x, d G*x^2, d G*xy);
// pixel p then gathers them from its neighbours:  g_x[p] = sum_q G(q-p) (c_mu(q) + 2 x_p c_xx(q) + y_p c_xy(q)).
// 3x3 neighbourhoods go through zero-bordered LDS planes private to the wave (LDS operations of one wave execute in order: no barrier).
#include <hip/hip_runtime.h>

#include "../../include/bts_render.h"

namespace bts {

void set_error(const char* fmt, const char* what, long a = 0, long b = 0, long c = 0);

struct LossParams {
  const float* rgb;      // (B, nv, 3)
  const float* depth;    // (B)
  const float* weights;  // (B, K)      (invalid policy weight_guided)
  const float* invalid;  // (B, K, nv)  (policies strict / weight_guided)
  const float* invalid_wsum;  // (B, nv) or null: sum_k w_k invalid_k,v  } from the renderer's epilogue; replace weights / invalid
  const float* invalid_any;   // (B, nv) or null: max_k invalid_k,v      }
  const float* rgb_gt;   // (B, 3)
  float* parts;          // (patches, 4): sum rgb term, sum smoothness term, invalid rays, 0
  float* g_rgb;          // (B, nv, 3) or null
  float* g_depth;        // (B) or null
  int n_patches, ph, pw, nv, K, policy;  // policy 0 none, 1 strict, 2 weight_guided
  float s_rgb, s_eas;    // d loss / d (sum rgb term), d loss / d (sum smoothness term)
  int has_eas;
};

constexpr int kPlane = 208;  // floats per LDS plane: (ph + 2) * (pw + 2) <= 3 * 66

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Lanes of a wave exchange pixels through LDS planes.  The hardware executes a wave's LDS instructions in order, but the COMPILER
// sees one thread storing to plane[ctr] and loading plane[ctr +- 1]: provably different addresses, so it may hoist the loads above
// the store (and the next store above these loads).  This fence pins the order at wave scope; it costs no instruction.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Gaussian-weighted sum of the 3x3 neighbourhood of `centre` in a zero-bordered plane (layers.py:92-101 window)
__device__ __forceinline__ float gauss9(const float* pl, int centre, int ld) {
  constexpr float a = 0.0947f, b = 0.1183f, c = 0.1478f;
  const float* r0 = pl + centre - ld;
  const float* r1 = pl + centre;
  const float* r2 = pl + centre + ld;
  float s = a * r0[-1];
  s += b * r0[0], s += a * r0[1];
  s += b * r1[-1], s += c * r1[0], s += b * r1[1];
  s += a * r2[-1], s += b * r2[0], s += a * r2[1];
  return s;
}

struct SsimStats {
  float mu_x, gxx, gxy;
};

__global__ __launch_bounds__(256) void photometric_loss_kernel(const LossParams p) {
  // per wave: Y planes (3 channels), X plane, products X^2 and XY, and three coefficient planes for the backward gather
  __shared__ float lds[4][9][kPlane];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int patch = blockIdx.x * 4 + wave;
  if (patch >= p.n_patches) return;
  float(*pl)[kPlane] = lds[wave];
  for (int i = lane; i < 9 * kPlane; i += 64) pl[0][i] = 0.0f;
  wave_lds_fence();
  const int ph = p.ph, pw = p.pw, nv = p.nv, K = p.K, area = ph * pw;
  const int ld = pw + 2;
  const bool act = lane < area;
  const int ly = act ? lane / pw : 0, lx = act ? lane - ly * pw : 0;
  const int ctr = (ly + 1) * ld + lx + 1;
  const long ray = (long)patch * area + (act ? lane : 0);

  // ---- invalid ray?  (loss.py:100-118)
  bool invalid = false;
  if (p.policy != 0) {
    bool all_v = true;
    for (int v = 0; v < nv; ++v) {
      if (p.policy == 2 && p.invalid_wsum) {          // the renderer's epilogue already summed over the samples
        all_v = all_v && (p.invalid_wsum[ray * nv + v] > 0.9f);
      } else if (p.policy == 1 && p.invalid_any) {
        all_v = all_v && (p.invalid_any[ray * nv + v] > 0.5f);
      } else if (p.policy == 2) {
        float s = 0.0f;
        for (int k = 0; k < K; ++k) s += p.invalid[(ray * K + k) * nv + v] * p.weights[ray * K + k];
        all_v = all_v && (s > 0.9f);
      } else {
        bool any_k = false;
        for (int k = 0; k < K; ++k) any_k = any_k || (p.invalid[(ray * K + k) * nv + v] > 0.5f);
        all_v = all_v && any_k;
      }
    }
    invalid = all_v;
  }
  const float keep = (act && !invalid) ? 1.0f : 0.0f;

  // ---- ground truth: planes Y_c, and the per-channel statistics that do not depend on the view
  float y[3], mu_y[3], gyy[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    y[c] = act ? p.rgb_gt[ray * 3 + c] : 0.0f;
    if (act) pl[c][ctr] = y[c];
  }
  wave_lds_fence();
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    mu_y[c] = gauss9(pl[c], ctr, ld);
    if (act) pl[3][ctr] = y[c] * y[c];
    wave_lds_fence();
    gyy[c] = gauss9(pl[3], ctr, ld);
    wave_lds_fence();
  }

  constexpr float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
  // statistics of (view v, channel c) at this pixel; planes 3 / 4 / 5 hold x, x^2, x*y
  auto stats = [&](float x, int c) {
    if (act) pl[3][ctr] = x, pl[4][ctr] = x * x, pl[5][ctr] = x * y[c];
    wave_lds_fence();
    SsimStats s;
    s.mu_x = gauss9(pl[3], ctr, ld);
    s.gxx = gauss9(pl[4], ctr, ld);
    s.gxy = gauss9(pl[5], ctr, ld);
    wave_lds_fence();
    return s;
  };

  // ---- pass 1: e_v, the minimum over the views (loss.py:152-153)
  float e_min = 0.0f;
  int v_star = 0;
  for (int v = 0; v < nv; ++v) {
    float ss = 0.0f, l1 = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float x = act ? p.rgb[(ray * nv + v) * 3 + c] : 0.0f;
      const SsimStats s = stats(x, c);
      const float mxx = s.mu_x * s.mu_x, myy = mu_y[c] * mu_y[c], mxy = s.mu_x * mu_y[c];
      const float sx = s.gxx - mxx, sy = gyy[c] - myy, sxy = s.gxy - mxy;
      const float nn = (2.0f * mxy + c1) * (2.0f * sxy + c2);
      const float dd = (mxx + myy + c1) * (sx + sy + c2);
      ss += fminf(fmaxf(1.0f - nn / dd, 0.0f), 1.0f) / 2.0f;
      l1 += fabsf(x - y[c]);
    }
    const float e = 0.85f * (ss / 3.0f) + 0.15f * (l1 / 3.0f);
    if (v == 0 || e < e_min) e_min = e, v_star = v;
  }
  const float L = e_min * keep;          // policy none: keep = 1 on every pixel
  const float up = keep * p.s_rgb;       // d loss / d e_{v*}(this pixel)

  // ---- pass 2: gradient with respect to the rendered colours
  if (p.g_rgb) {
    for (int v = 0; v < nv; ++v) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float x = act ? p.rgb[(ray * nv + v) * 3 + c] : 0.0f;
        const SsimStats s = stats(x, c);
        float k_mu = 0.0f, k_xx = 0.0f, k_xy = 0.0f, g_l1 = 0.0f;
        if (act && v == v_star && up != 0.0f) {
          const float mxx = s.mu_x * s.mu_x, myy = mu_y[c] * mu_y[c], mxy = s.mu_x * mu_y[c];
          const float sx = s.gxx - mxx, sy = gyy[c] - myy, sxy = s.gxy - mxy;
          const float A1 = 2.0f * mxy + c1, A2 = 2.0f * sxy + c2, B1 = mxx + myy + c1, B2 = sx + sy + c2;
          const float nn = A1 * A2, dd = B1 * B2;
          const float t = 1.0f - nn / dd;
          if (t >= 0.0f && t <= 1.0f) {   // torch.clamp passes the gradient on the closed interval
            // d(ssim) = -1/2 d(n/d);  d(n/d) = dn/d - n dd/d^2
            const float gs = up * (0.85f / 3.0f) * (-0.5f);
            const float dn_mu = 2.0f * mu_y[c] * (A2 - A1), dn_xy = 2.0f * A1;
            const float dd_mu = 2.0f * s.mu_x * (B2 - B1), dd_xx = B1;
            const float inv_d = 1.0f / dd, n_d2 = nn * inv_d * inv_d;
            k_mu = gs * (dn_mu * inv_d - n_d2 * dd_mu);
            k_xy = gs * (dn_xy * inv_d);
            k_xx = gs * (-n_d2 * dd_xx);
          }
          const float df = x - y[c];
          g_l1 = up * (0.15f / 3.0f) * (df > 0.0f ? 1.0f : (df < 0.0f ? -1.0f : 0.0f));
        }
        if (act) pl[6][ctr] = k_mu, pl[7][ctr] = k_xx, pl[8][ctr] = k_xy;
        wave_lds_fence();
        const float t_mu = gauss9(pl[6], ctr, ld), t_xx = gauss9(pl[7], ctr, ld), t_xy = gauss9(pl[8], ctr, ld);
        wave_lds_fence();
        if (act) p.g_rgb[(ray * nv + v) * 3 + c] = t_mu + 2.0f * x * t_xx + y[c] * t_xy + g_l1;
      }
    }
  }

  // ---- edge-aware smoothness of the normalised inverse depth (loss.py:21-40), masked by the invalid rays (:262-266)
  float eas = 0.0f;
  if (p.has_eas) {
    const float dep = act ? p.depth[ray] : 1.0f;
    const float dcl = fminf(fmaxf(dep, 1e-3f), 80.0f);
    const float d = act ? 1.0f / dcl : 0.0f;
    const float m = wave_sum(d) / (float)area;
    const float dn = d / m;
    const float keep_e = keep;
    // neighbours to the right (lane + 1) and below (lane + pw)
    const float dn_r = __shfl_down(dn, 1, 64), dn_b = __shfl_down(dn, pw, 64);
    float idx = 0.0f, idy = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      idx += fabsf(y[c] - __shfl_down(y[c], 1, 64));
      idy += fabsf(y[c] - __shfl_down(y[c], pw, 64));
    }
    const bool has_r = act && lx + 1 < pw, has_b = act && ly + 1 < ph;
    const float wx = has_r ? expf(-(idx / 3.0f)) : 0.0f, wy = has_b ? expf(-(idy / 3.0f)) : 0.0f;
    const float ex = dn - dn_r, ey = dn - dn_b;
    eas = (fabsf(ex) * wx + fabsf(ey) * wy) * keep_e;
    if (p.g_depth) {
      // u_p = d (sum eas) / d dn_p: this pixel's own edges and the edges of its left / upper neighbour
      const float sgx = (ex > 0.0f ? 1.0f : (ex < 0.0f ? -1.0f : 0.0f)) * wx * keep_e;
      const float sgy = (ey > 0.0f ? 1.0f : (ey < 0.0f ? -1.0f : 0.0f)) * wy * keep_e;
      const float from_l = __shfl_up(sgx, 1, 64), from_u = __shfl_up(sgy, pw, 64);
      float u = sgx + sgy;
      if (act && lx > 0) u -= from_l;
      if (act && ly > 0) u -= from_u;
      if (!act) u = 0.0f;
      // dn_p = d_p / m, m = mean d:  d/d d_j = u_j / m - (sum_p u_p d_p) / (m^2 N)
      const float sud = wave_sum(u * d);
      const float gd = u / m - sud / (m * m * (float)area);
      const float g_dep = (dep >= 1e-3f && dep <= 80.0f) ? -gd / (dcl * dcl) : 0.0f;
      if (act) p.g_depth[ray] = g_dep * p.s_eas;
    }
  } else if (p.g_depth && act) {
    p.g_depth[ray] = 0.0f;
  }

  const float s_rgb = wave_sum(L), s_eas = wave_sum(eas), s_inv = wave_sum((act && invalid) ? 1.0f : 0.0f);
  if (lane == 0) {
    float4 o = make_float4(s_rgb, s_eas, s_inv, 0.0f);
    reinterpret_cast<float4*>(p.parts)[patch] = o;
  }
}

int photometric_loss_impl(const BtsLossArgs* a, hipStream_t s) {
  LossParams p;
  p.rgb = a->rgb, p.depth = a->depth, p.weights = a->weights, p.invalid = a->invalid, p.rgb_gt = a->rgb_gt;
  p.invalid_wsum = a->invalid_wsum, p.invalid_any = a->invalid_any;
  p.parts = a->parts, p.g_rgb = a->g_rgb, p.g_depth = a->g_depth;
  p.n_patches = a->n_patches, p.ph = a->patch_h, p.pw = a->patch_w, p.nv = a->nv, p.K = a->K, p.policy = a->invalid_policy;
  p.s_rgb = a->scale_rgb, p.s_eas = a->scale_eas, p.has_eas = a->edge_aware_smoothness;
  if (p.n_patches == 0) return BTS_OK;
  const int grid = (p.n_patches + 3) / 4;
  photometric_loss_kernel<<<grid, 256, 0, s>>>(p);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: loss kernel launch failed (%ld)", hipGetErrorString(e), (long)e);
    return BTS_E_LAUNCH;
  }
  return BTS_OK;
}

}  // namespace bts
