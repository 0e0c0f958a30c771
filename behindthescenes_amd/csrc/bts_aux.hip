// Small HBM-bound kernels at the edges of the render path: layout changes at the hand-off from the PyTorch encoder,
// ray generation (util.py:113-149, 244-273), stratified depth sampling (nerf.py:103-123) and distance->z
// (projection_operations.py:4-16).  All are coalesced streaming kernels; none does arithmetic worth an MFMA.
#include "bts_common.h"
#include <cstdint>

namespace bts {

void set_error(const char* fmt, const char* a = "", long b = 0, long c = 0, long d = 0);

// ----------------------------------------------------------------------------------------------------------------
// (N, C, HW) <-> (N, HW, C) through a 64x65 LDS tile: both the read and the write are 256-byte coalesced rows.
// ----------------------------------------------------------------------------------------------------------------
template <bool TO_NHWC>
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int HW) {
  __shared__ float tile[64][65];
  const int n = blockIdx.z;
  const int c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  const float* s = src + (long)n * C * HW;
  float* d = dst + (long)n * C * HW;
  if (TO_NHWC) {
    // read rows = channels, cols = pixels
    for (int i = ty; i < 64; i += 4) {
      const int c = c0 + i, p = p0 + tx;
      if (c < C && p < HW) tile[i][tx] = s[(long)c * HW + p];
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
      const int p = p0 + i, c = c0 + tx;
      if (c < C && p < HW) d[(long)p * C + c] = tile[tx][i];
    }
  } else {
    for (int i = ty; i < 64; i += 4) {
      const int p = p0 + i, c = c0 + tx;
      if (c < C && p < HW) tile[i][tx] = s[(long)p * C + c];
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
      const int c = c0 + i, p = p0 + tx;
      if (c < C && p < HW) d[(long)c * HW + p] = tile[tx][i];
    }
  }
}

int transpose_launch(const float* src, float* dst, int N, int C, int H, int W, bool to_nhwc, hipStream_t s) {
  const int HW = H * W;
  dim3 grid((HW + 63) / 64, (C + 63) / 64, N);
  if (to_nhwc) transpose_kernel<true><<<grid, 256, 0, s>>>(src, dst, C, HW);
  else transpose_kernel<false><<<grid, 256, 0, s>>>(src, dst, C, HW);
  return hipGetLastError() == hipSuccess ? BTS_OK : BTS_E_LAUNCH;
}

// (N, 3, H, W) -> (N, H, W, 4) rgb0, x*scale + shift  (encode's images * .5 + .5, models_bts.py:82).  With a frame table (nv > 0) the
// source is (n, v, 3, H, W) and output frame (smp, j) reads input frame (smp, ids[j]): BTSNet.encode's `images[:, ids_render]` without
// the copy (bts_train_step_fwd)
struct FrameIds {
  int nv, v, ids[BTS_MAX_VIEWS];
};
__device__ __forceinline__ void pack_rgb_body(const float* __restrict__ src, float4* __restrict__ dst, long HW, long total, float scale, float shift,
                                              const FrameIds& f, long block, long n_blocks) {
  // four pixels per thread and trip, 256 apart (every load and every store of a wave stays one contiguous piece): twelve independent
  // loads in flight per thread instead of three dependent round trips per pixel -- the pass is a pure stream (98 MB in, 129 MB out at
  // exp_kitti_360.yaml's batch).  (Four CONSECUTIVE pixels per thread -- float4 loads -- was tried first: its 16-byte stores are 64 bytes
  // apart across the lanes, four partial writes per 64-byte piece, and the pass did not get faster: profiles/r06n.)
  long i0 = block * 1024L + threadIdx.x;
  for (; i0 + 768 < total; i0 += n_blocks * 1024) {
    float v[4][3];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long i = i0 + 256 * k;
      long n = i / HW;
      const long p = i - n * HW;
      if (f.nv > 0) n = (n / f.nv) * f.v + f.ids[n % f.nv];
      const float* s = src + n * 3 * HW + p;
      v[k][0] = s[0], v[k][1] = s[HW], v[k][2] = s[2 * HW];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[i0 + 256 * k] = make_float4(v[k][0] * scale + shift, v[k][1] * scale + shift, v[k][2] * scale + shift, 0.0f);
  }
  // the ragged end of this thread's last trip
  for (long i = i0; i < total; i += 256) {
    if (i >= i0 + 1024) break;
    long n = i / HW;
    const long p = i - n * HW;
    if (f.nv > 0) n = (n / f.nv) * f.v + f.ids[n % f.nv];
    const float* s = src + n * 3 * HW + p;
    dst[i] = make_float4(s[0] * scale + shift, s[HW] * scale + shift, s[2 * HW] * scale + shift, 0.0f);
  }
}
__global__ __launch_bounds__(256) void pack_rgb_kernel(const float* __restrict__ src, float4* __restrict__ dst, long HW, long total,
                                                       float scale, float shift, const FrameIds f) {
  pack_rgb_body(src, dst, HW, total, scale, shift, f, blockIdx.x, gridDim.x);
}

int pack_rgb_launch(const float* src, float* dst, int N, int H, int W, float scale, float shift, hipStream_t s) {
  const long HW = (long)H * W, total = HW * N;
  const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  FrameIds f;
  f.nv = 0, f.v = 0;
  pack_rgb_kernel<<<grid, 256, 0, s>>>(src, reinterpret_cast<float4*>(dst), HW, total, scale, shift, f);
  return hipGetLastError() == hipSuccess ? BTS_OK : BTS_E_LAUNCH;
}

// frames (n, v, 3, H, W) -> (n, nv, H, W, 4): frame ids[j] of every batch element
int pack_rgb_views_launch(const float* src, float* dst, int n, int v, int nv, const int* ids, int H, int W, float scale, float shift, hipStream_t s) {
  const long HW = (long)H * W, total = HW * n * nv;
  if (total == 0) return BTS_OK;
  const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  FrameIds f;
  f.nv = nv, f.v = v;
  for (int j = 0; j < BTS_MAX_VIEWS; ++j) f.ids[j] = j < nv ? ids[j] : 0;
  pack_rgb_kernel<<<grid, 256, 0, s>>>(src, reinterpret_cast<float4*>(dst), HW, total, scale, shift, f);
  return hipGetLastError() == hipSuccess ? BTS_OK : BTS_E_LAUNCH;
}

// the ray of pixel (x, y) of a view: [c2w translation, R @ unproj(pixel), near, far]  (util.py:113-149, 244-273)
__device__ __forceinline__ void ray_of_pixel(const float* __restrict__ P, const float* __restrict__ Kp, int H, int W, int x, int y,
                                             float z_near, float z_far, int norm_dir, float4& a, float4& b) {
  const float fx = Kp[0], fy = Kp[4], cx = Kp[2], cy = Kp[5];
  const float gx = W > 1 ? linspace_at(-1.0f, 1.0f, W, x) : -1.0f;
  const float gy = H > 1 ? linspace_at(-1.0f, 1.0f, H, y) : -1.0f;
  float d0 = (gx - cx) / fx, d1 = (gy - cy) / fy, d2 = 1.0f;
  if (norm_dir) {
    const float nrm = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    d0 = d0 / nrm, d1 = d1 / nrm, d2 = d2 / nrm;
  }
  float w[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float t = P[4 * r + 0] * d0;
    t = __builtin_fmaf(P[4 * r + 1], d1, t);
    t = __builtin_fmaf(P[4 * r + 2], d2, t);
    w[r] = t;
  }
  a = make_float4(P[3], P[7], P[11], w[0]);
  b = make_float4(w[1], w[2], z_near, z_far);
}

// gen_rays: rays (V, H, W, 8)
__global__ __launch_bounds__(256) void gen_rays_kernel(const float* __restrict__ poses, const float* __restrict__ projs, int V, int H, int W,
                                                       float z_near, float z_far, int norm_dir, float4* __restrict__ rays) {
  const long total = (long)V * H * W;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int v = (int)(i / ((long)H * W));
    const int rem = (int)(i - (long)v * H * W);
    const int y = rem / W, x = rem - y * W;
    float4 a, b;
    ray_of_pixel(poses + v * 16, projs + v * 9, H, W, x, y, z_near, z_far, norm_dir, a, b);
    rays[2 * i] = a;
    rays[2 * i + 1] = b;
  }
}

int gen_rays_launch(const float* poses, const float* projs, int V, int H, int W, float zn, float zf, int norm_dir, float* rays,
                    hipStream_t s) {
  const long total = (long)V * H * W;
  const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  gen_rays_kernel<<<grid, 256, 0, s>>>(poses, projs, V, H, W, zn, zf, norm_dir, reinterpret_cast<float4*>(rays));
  return hipGetLastError() == hipSuccess ? BTS_OK : BTS_E_LAUNCH;
}

// patch_rays: PatchRaySampler.sample on device (ray_sampler.py:125-162).  Per sample P patches (view, y0, x0) of ph x pw pixels:
// the rays of exactly those pixels (never the (n, v, H, W, 8) ray volume the reference builds and then slices) and, when frames are
// given, their ground-truth colours gathered from the NCHW frames.  Output order = the reference's: patch-major, then row, then column.
// With a view table (m.n > 0) the patch's view index names an entry of it (the reference samples from images[:, ids_loss]); the colours
// leave as x * gt_scale + gt_shift (mul, then add: the trainer's images * .5 + .5 applied to the gathered pixels only).
struct ViewTable {
  int n, ids[BTS_MAX_LOSS_VIEWS];
};
__device__ __forceinline__ void patch_rays_body(const float* __restrict__ poses, const float* __restrict__ projs, const float* __restrict__ images,
                                                const int* __restrict__ pv, const int* __restrict__ py, const int* __restrict__ px, int n, int v, int c, int H,
                                                int W, int P, int ph, int pw, float z_near, float z_far, int norm_dir, float4* __restrict__ rays,
                                                float* __restrict__ gt, const ViewTable& m, float gt_scale, float gt_shift, long block, long n_blocks) {
  const int per = P * ph * pw;
  const long total = (long)n * per;
  for (long i = block * 256L + threadIdx.x; i < total; i += n_blocks * 256) {
    const int smp = (int)(i / per);
    int rem = (int)(i - (long)smp * per);
    const int patch = rem / (ph * pw);
    rem -= patch * ph * pw;
    const int dy = rem / pw, dx = rem - dy * pw;
    int view = pv[smp * P + patch];
    if (m.n > 0) view = m.ids[min(max(view, 0), m.n - 1)];
    const int y = py[smp * P + patch] + dy, x = px[smp * P + patch] + dx;
    float4 a, b;
    ray_of_pixel(poses + ((long)smp * v + view) * 16, projs + ((long)smp * v + view) * 9, H, W, x, y, z_near, z_far, norm_dir, a, b);
    rays[2 * i] = a;
    rays[2 * i + 1] = b;
    if (gt) {
      const float* img = images + (((long)smp * v + view) * c) * H * W + (long)y * W + x;
      for (int ch = 0; ch < c; ++ch) gt[i * c + ch] = img[(long)ch * H * W] * gt_scale + gt_shift;
    }
  }
}
__global__ __launch_bounds__(256) void patch_rays_kernel(const float* __restrict__ poses, const float* __restrict__ projs,
                                                         const float* __restrict__ images, const int* __restrict__ pv,
                                                         const int* __restrict__ py, const int* __restrict__ px, int n, int v, int c, int H,
                                                         int W, int P, int ph, int pw, float z_near, float z_far, int norm_dir,
                                                         float4* __restrict__ rays, float* __restrict__ gt, const ViewTable m, float gt_scale,
                                                         float gt_shift) {
  patch_rays_body(poses, projs, images, pv, py, px, n, v, c, H, W, P, ph, pw, z_near, z_far, norm_dir, rays, gt, m, gt_scale, gt_shift, blockIdx.x, gridDim.x);
}

int patch_rays_views_launch(const float* poses, const float* projs, const float* images, const int* pv, const int* py, const int* px, int n, int v,
                            int c, int H, int W, int P, int ph, int pw, float zn, float zf, int norm_dir, float* rays, float* gt, int n_ids,
                            const int* ids, float gt_scale, float gt_shift, hipStream_t s) {
  const long total = (long)n * P * ph * pw;
  if (total == 0) return BTS_OK;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  ViewTable m;
  m.n = n_ids;
  for (int j = 0; j < BTS_MAX_LOSS_VIEWS; ++j) m.ids[j] = j < n_ids ? ids[j] : 0;
  patch_rays_kernel<<<grid, 256, 0, s>>>(poses, projs, images, pv, py, px, n, v, c, H, W, P, ph, pw, zn, zf, norm_dir,
                                         reinterpret_cast<float4*>(rays), images ? gt : nullptr, m, gt_scale, gt_shift);
  return hipGetLastError() == hipSuccess ? BTS_OK : BTS_E_LAUNCH;
}

int patch_rays_launch(const float* poses, const float* projs, const float* images, const int* pv, const int* py, const int* px, int n, int v,
                      int c, int H, int W, int P, int ph, int pw, float zn, float zf, int norm_dir, float* rays, float* gt, hipStream_t s) {
  return patch_rays_views_launch(poses, projs, images, pv, py, px, n, v, c, H, W, P, ph, pw, zn, zf, norm_dir, rays, gt, 0, nullptr, 1.0f, 0.0f, s);
}

// sample_coarse: s = linspace(0, 1-1/K, K)[k] + u/K ; z = near(1-s) + far s   or   1/((1/near)(1-s) + (1/far) s)
__global__ __launch_bounds__(256) void sample_coarse_kernel(const float* __restrict__ rays, const float* __restrict__ u, long B, int K, int lindisp,
                                                            float* __restrict__ z) {
  const long total = B * K;
  const float step = 1.0f / (float)K;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / K;
    const int k = (int)(i - b * K);
    const float near = rays[b * 8 + 6], far = rays[b * 8 + 7];
    z[i] = coarse_depth(u[i], coarse_base(K, k), step, near, far, lindisp != 0);   // the render kernel's own routine (BtsRenderArgs.jitter)
  }
}

int sample_coarse_launch(const float* rays, const float* u, long B, int K, int lindisp, float* z, hipStream_t s) {
  const long total = B * K;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  sample_coarse_kernel<<<grid, 256, 0, s>>>(rays, u, B, K, lindisp, z);
  return hipGetLastError() == hipSuccess ? BTS_OK : BTS_E_LAUNCH;
}

// distance_to_z: z = depth * cam.z / |cam|,  cam = inv_K @ [gx, gy, 1]
__global__ __launch_bounds__(256) void distance_to_z_kernel(const float* __restrict__ depths, const float* __restrict__ invK, int N, int H, int W,
                                                            float* __restrict__ out) {
  const long total = (long)N * H * W;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int n = (int)(i / ((long)H * W));
    const int rem = (int)(i - (long)n * H * W);
    const int y = rem / W, x = rem - y * W;
    const float* M = invK + n * 9;
    const float gx = W > 1 ? linspace_at(-1.0f, 1.0f, W, x) : -1.0f;
    const float gy = H > 1 ? linspace_at(-1.0f, 1.0f, H, y) : -1.0f;
    float c[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float a = M[3 * r + 0] * gx;
      a = __builtin_fmaf(M[3 * r + 1], gy, a);
      a = __builtin_fmaf(M[3 * r + 2], 1.0f, a);
      c[r] = a;
    }
    const float nrm = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    out[i] = depths[i] * (c[2] / nrm);
  }
}

int distance_to_z_launch(const float* depths, const float* invK, int N, int H, int W, float* out, hipStream_t s) {
  const long total = (long)N * H * W;
  const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  distance_to_z_kernel<<<grid, 256, 0, s>>>(depths, invK, N, H, W, out);
  return hipGetLastError() == hipSuccess ? BTS_OK : BTS_E_LAUNCH;
}

// ----------------------------------------------------------------------------------------------------------------
// Batched inverse of 3x3 / 4x4 matrices (torch.inverse at models_bts.py:71 on the c2w poses and at
// projection_operations.py:9 on the intrinsics): one thread per matrix, Gauss-Jordan with partial pivoting in fp64, rounded once to
// fp32 -- within 1 ulp of any correctly working fp32 LU, and no hipSOLVER call / host synchronisation on the render path.
// ----------------------------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void invert_small_dev(const float* __restrict__ src, float* __restrict__ dst) {
  double a[D][2 * D];
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c < D; ++c) a[r][c] = (double)src[r * D + c], a[r][D + c] = r == c ? 1.0 : 0.0;
#pragma unroll
  for (int col = 0; col < D; ++col) {
    int piv = col;
    double best = fabs(a[col][col]);
#pragma unroll
    for (int r = col + 1; r < D; ++r)
      if (fabs(a[r][col]) > best) best = fabs(a[r][col]), piv = r;
#pragma unroll
    for (int r = col + 1; r < D; ++r)
      if (r == piv) {
#pragma unroll
        for (int c = 0; c < 2 * D; ++c) {
          const double t = a[col][c];
          a[col][c] = a[r][c], a[r][c] = t;
        }
      }
    const double inv = 1.0 / a[col][col];  // singular input -> inf / nan, like torch.inverse's garbage-or-error; callers pass rigid poses
#pragma unroll
    for (int c = 0; c < 2 * D; ++c) a[col][c] *= inv;
#pragma unroll
    for (int r = 0; r < D; ++r)
      if (r != col) {
        const double f = a[r][col];
#pragma unroll
        for (int c = 0; c < 2 * D; ++c) a[r][c] -= f * a[col][c];
      }
  }
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c < D; ++c) dst[r * D + c] = (float)a[r][D + c];
}

template <int D>
__global__ __launch_bounds__(64) void invert_small_kernel(const float* __restrict__ src, float* __restrict__ dst, int N) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= N) return;
  invert_small_dev<D>(src + (long)i * D * D, dst + (long)i * D * D);
}

// The camera part of BTSNet.encode (models_bts.py:71-97, 120-136) for a training step: world -> camera of the encoder frame and of the
// nv render frames (the inverse above on exactly those poses) and their intrinsics, gathered into the contiguous blocks the render
// kernels read: cams = [K_enc (n, 9) | w2c_enc (n, 16) | K_r (n, nv, 9) | w2c_r (n, nv, 16)].  One thread per (batch element, frame).
struct CamIds {
  int nv, v, id_enc, ids[BTS_MAX_VIEWS];
};
__device__ __forceinline__ void camera_prep_body(const float* __restrict__ Ks, const float* __restrict__ poses, int n, const CamIds& f,
                                                 float* __restrict__ cams, int i) {
  if (i >= n * (1 + f.nv)) return;
  const int smp = i / (1 + f.nv), j = i - smp * (1 + f.nv);
  const int frame = j == 0 ? f.id_enc : f.ids[j - 1];
  float* K_enc = cams;
  float* w2c_enc = K_enc + (long)n * 9;
  float* K_r = w2c_enc + (long)n * 16;
  float* w2c_r = K_r + (long)n * f.nv * 9;
  float* Kd = j == 0 ? K_enc + smp * 9 : K_r + ((long)smp * f.nv + (j - 1)) * 9;
  float* Pd = j == 0 ? w2c_enc + smp * 16 : w2c_r + ((long)smp * f.nv + (j - 1)) * 16;
  const float* Ksrc = Ks + ((long)smp * f.v + frame) * 9;
#pragma unroll
  for (int e = 0; e < 9; ++e) Kd[e] = Ksrc[e];
  invert_small_dev<4>(poses + ((long)smp * f.v + frame) * 16, Pd);
}
__global__ __launch_bounds__(64) void camera_prep_kernel(const float* __restrict__ Ks, const float* __restrict__ poses, int n, const CamIds f,
                                                        float* __restrict__ cams) {
  camera_prep_body(Ks, poses, n, f, cams, blockIdx.x * 64 + threadIdx.x);
}

// ---- the hand-over of a training step in ONE launch (bts_train_step_fwd): cameras, rgb0 packing of the render frames, patch rays + colours
// and the zero fill of every scale's tile flags are independent of each other; as four launches on one queue each paid its own dispatch
// gap on an otherwise idle GPU (the step after them is 0.2 ms at exp_kitti_raw.yaml's shapes).  Work-group ranges take the roles.
struct HandoverParams {
  const float* Ks;
  const float* poses;
  const float* images;
  const int *pv, *py, *px;
  float* cams;
  float4* imgs;
  float4* rays;
  float* gt;
  unsigned char* zero[BTS_MAX_SCALES];
  long zero_bytes[BTS_MAX_SCALES];
  CamIds cam;
  FrameIds frames;
  ViewTable loss;
  int n, v, H, W, P, ph, pw, n_zero;
  float z_near, z_far, scale, shift;
  int b_cam, b_pack, b_patch, b_zero;    // work-groups per role, in this order
};
__global__ __launch_bounds__(256) void handover_kernel(const HandoverParams p) {
  int b = blockIdx.x;
  if (b < p.b_cam) {
    camera_prep_body(p.Ks, p.poses, p.n, p.cam, p.cams, b * 256 + threadIdx.x);
    return;
  }
  b -= p.b_cam;
  if (b < p.b_pack) {
    const long HW = (long)p.H * p.W;
    pack_rgb_body(p.images, p.imgs, HW, HW * p.n * p.frames.nv, p.scale, p.shift, p.frames, b, p.b_pack);
    return;
  }
  b -= p.b_pack;
  if (b < p.b_patch) {
    patch_rays_body(p.poses, p.Ks, p.images, p.pv, p.py, p.px, p.n, p.v, 3, p.H, p.W, p.P, p.ph, p.pw, p.z_near, p.z_far, 1, p.rays, p.gt, p.loss, p.scale,
                    p.shift, b, p.b_patch);
    return;
  }
  b -= p.b_patch;
  for (int r = 0; r < p.n_zero; ++r) {       // (flag arrays are multiples of 4 bytes only by accident: bytes at the ragged end)
    unsigned* w = reinterpret_cast<unsigned*>(p.zero[r]);
    const long words = p.zero_bytes[r] >> 2;
    for (long i = b * 256L + threadIdx.x; i < words; i += (long)p.b_zero * 256) w[i] = 0u;
    if (b == 0 && threadIdx.x < (p.zero_bytes[r] & 3)) p.zero[r][(words << 2) + threadIdx.x] = 0;
  }
}

int handover_launch(const float* Ks, const float* poses, const float* images, const int* pv, const int* py, const int* px, int n, int v, int id_enc, int nv,
                    const int* ids_render, int n_loss, const int* ids_loss, int H, int W, int P, int ph, int pw, float z_near, float z_far, float scale,
                    float shift, float* cams, float* imgs, float* rays, float* gt, int n_zero, unsigned char* const* zero, const long* zero_bytes,
                    hipStream_t s) {
  HandoverParams p;
  p.Ks = Ks, p.poses = poses, p.images = images, p.pv = pv, p.py = py, p.px = px, p.cams = cams;
  p.imgs = reinterpret_cast<float4*>(imgs), p.rays = reinterpret_cast<float4*>(rays), p.gt = gt;
  p.cam.nv = nv, p.cam.v = v, p.cam.id_enc = id_enc, p.frames.nv = nv, p.frames.v = v, p.loss.n = n_loss;
  for (int j = 0; j < BTS_MAX_VIEWS; ++j) p.cam.ids[j] = p.frames.ids[j] = j < nv ? ids_render[j] : 0;
  for (int j = 0; j < BTS_MAX_LOSS_VIEWS; ++j) p.loss.ids[j] = j < n_loss ? ids_loss[j] : 0;
  p.n = n, p.v = v, p.H = H, p.W = W, p.P = P, p.ph = ph, p.pw = pw, p.n_zero = n_zero;
  long zmax = 0;
  for (int r = 0; r < BTS_MAX_SCALES; ++r) {
    p.zero[r] = r < n_zero ? zero[r] : nullptr, p.zero_bytes[r] = r < n_zero ? zero_bytes[r] : 0;
    if (p.zero_bytes[r] > zmax) zmax = p.zero_bytes[r];
  }
  p.z_near = z_near, p.z_far = z_far, p.scale = scale, p.shift = shift;
  const long px_total = (long)n * nv * H * W, ray_total = (long)n * P * ph * pw;
  p.b_cam = (n * (1 + nv) + 255) / 256;
  p.b_pack = (int)((px_total + 255) / 256 < 2048 ? (px_total + 255) / 256 : 2048);
  p.b_patch = (int)((ray_total + 255) / 256 < 1024 ? (ray_total + 255) / 256 : 1024);
  p.b_zero = n_zero ? (int)((zmax / 4 + 255) / 256 < 64 ? (zmax / 4 + 255) / 256 + 1 : 64) : 0;
  handover_kernel<<<p.b_cam + p.b_pack + p.b_patch + p.b_zero, 256, 0, s>>>(p);
  return hipGetLastError() == hipSuccess ? BTS_OK : BTS_E_LAUNCH;
}

int camera_prep_launch(const float* Ks, const float* poses, int n, int v, int id_enc, int nv, const int* ids, float* cams, hipStream_t s) {
  CamIds f;
  f.nv = nv, f.v = v, f.id_enc = id_enc;
  for (int j = 0; j < BTS_MAX_VIEWS; ++j) f.ids[j] = j < nv ? ids[j] : 0;
  const int total = n * (1 + nv);
  camera_prep_kernel<<<(total + 63) / 64, 64, 0, s>>>(Ks, poses, n, f, cams);
  return hipGetLastError() == hipSuccess ? BTS_OK : BTS_E_LAUNCH;
}

int invert_small_launch(const float* src, float* dst, int N, int dim, hipStream_t s) {
  const int grid = (N + 63) / 64;
  if (dim == 3) invert_small_kernel<3><<<grid, 64, 0, s>>>(src, dst, N);
  else if (dim == 4) invert_small_kernel<4><<<grid, 64, 0, s>>>(src, dst, N);
  else return BTS_E_UNSUPPORTED;
  return hipGetLastError() == hipSuccess ? BTS_OK : BTS_E_LAUNCH;
}

}  // namespace bts
