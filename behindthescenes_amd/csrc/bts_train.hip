// bts_train_step_fwd / bts_train_step_bwd (ABI 7): a training step's share of the renderer enqueued from ONE call per direction.
//
// Nothing here computes anything new: the forward is camera hand-over -> rgb0 packing of the render frames -> patch rays, then per
// scale tile flags -> projection of the flagged tiles -> the fused render kernel (lean outputs + the backward's saved state) -> the
// photometric loss pass, and one reduction of the per-patch sums; the backward is, per scale, the three render passes and the tile
// projection backward.  The kernels, their launch geometry and their arguments are those of the entry-by-entry path
// (bts_render_fwd, bts_photometric_loss, ...: the *_impl functions those entry points call) -- what goes away is the HOST work
// between them: ~20 ctypes calls, ~40 allocator round trips, the autograd nodes and ~25 small torch kernels per exp_kitti_raw.yaml
// step, which took longer to issue (1.1 ms) than the GPU needed to run the step's kernels (0.5 ms).
// Reference: BTSWrapper.forward (models/bts/trainer.py:208-259) + the criterion call of utils/base_trainer.py:287-297.
#include "bts_common.h"

#include <cstring>

namespace bts {

void set_error(const char* fmt, const char* a = "", long b = 0, long c = 0, long d = 0);
bool shape_supported(int C, int HD, int NB);

int camera_prep_launch(const float* Ks, const float* poses, int n, int v, int id_enc, int nv, const int* ids, float* cams, hipStream_t s);
int pack_rgb_views_launch(const float* src, float* dst, int n, int v, int nv, const int* ids, int H, int W, float scale, float shift, hipStream_t s);
int patch_rays_views_launch(const float* poses, const float* projs, const float* images, const int* pv, const int* py, const int* px, int n, int v,
                            int c, int H, int W, int P, int ph, int pw, float zn, float zf, int norm_dir, float* rays, float* gt, int n_ids,
                            const int* ids, float gt_scale, float gt_shift, hipStream_t s);
int photometric_loss_impl(const BtsLossArgs* a, hipStream_t s);
int project_features_impl(int C, int HD, const float* feat, const float* mlp, int N, int HW, float* proj, const unsigned char* tiles, hipStream_t s,
                          bool feat_cl = false, int Wm = 0,     // Wm: the map's width when `tiles` are 16 x 4 blocks (BtsFieldCfg.tile_blocks), 0 = runs of 64 texels
                          void* list_ws = nullptr, size_t list_ws_bytes = 0);   // scratch for the balanced (list-driven) form
int mark_tiles_impl(const float* rays, const float* z_samp, const float* jitter, const float* w2c_enc, const float* K_enc, long B, int Bp, int K, int lindisp,
                    int H, int W, int fs, unsigned char* tiles, hipStream_t s, int blocks);
int project_features_bwd_tiles_impl(int C, int HD, const float* feat, float* dproj, unsigned char* tiles, const float* mlp, int N, int HW, float* dfeat,
                                    float* d_mlp, int clear, hipStream_t s, bool feat_cl = false, int Wm = 0, void* list_ws = nullptr,
                                    size_t list_ws_bytes = 0);   // list_ws: scratch for the balanced (list-driven) form, project_bwd_list_bytes(N * tiles) bytes
int render_fwd_impl(const BtsFieldCfg* cfg, const BtsFieldTensors* t, const BtsRenderArgs* a, hipStream_t s);
size_t render_bwd_workspace_impl(const BtsFieldCfg* cfg, const BtsRenderArgs* a);
int render_bwd_impl(const BtsFieldCfg* cfg, const BtsFieldTensors* t, const BtsRenderArgs* a, const BtsRenderGrads* g, void* ws, size_t ws_bytes,
                    hipStream_t s, bool flush_clean);
void render_bwd_flush_region(const BtsFieldCfg* cfg, const BtsRenderArgs* a, void* workspace, float** ptr, size_t* bytes);
int handover_launch(const float* Ks, const float* poses, const float* images, const int* pv, const int* py, const int* px, int n, int v, int id_enc, int nv,
                    const int* ids_render, int n_loss, const int* ids_loss, int H, int W, int P, int ph, int pw, float z_near, float z_far, float scale,
                    float shift, float* cams, float* imgs, float* rays, float* gt, int n_zero, unsigned char* const* zero, const long* zero_bytes,
                    hipStream_t s);

// ---- loss_vals = loss_matrix . [per-scale sums of the loss pass' per-patch parts]: one work-group, lanes stride the patches
struct FinishParams {
  const float* parts[BTS_MAX_SCALES];   // (n_patches, 4) each
  int n_scales, n_patches;
  float M[9 * 3 * BTS_MAX_SCALES];
  float* out;   // (9)
};
__global__ __launch_bounds__(256) void loss_finish_kernel(const FinishParams p) {
  __shared__ float red[4][3 * BTS_MAX_SCALES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[3 * BTS_MAX_SCALES];
#pragma unroll
  for (int i = 0; i < 3 * BTS_MAX_SCALES; ++i) acc[i] = 0.0f;
#pragma unroll
  for (int s = 0; s < BTS_MAX_SCALES; ++s) {
    if (s >= p.n_scales) break;
    const float4* q = reinterpret_cast<const float4*>(p.parts[s]);
    for (int i = threadIdx.x; i < p.n_patches; i += 256) {
      const float4 v = q[i];
      acc[3 * s] += v.x, acc[3 * s + 1] += v.y, acc[3 * s + 2] += v.z;
    }
  }
#pragma unroll
  for (int i = 0; i < 3 * BTS_MAX_SCALES; ++i) {
    float v = acc[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 9) {
    const int cols = 3 * p.n_scales;
    float o = 0.0f;
    for (int c = 0; c < cols; ++c) o += p.M[threadIdx.x * cols + c] * (((red[0][c] + red[1][c]) + red[2][c]) + red[3][c]);
    p.out[threadIdx.x] = o;
  }
}

// ---- gs = g * (coefficient of the scale's sum in the loss * upstream gradient): what autograd does with the loss pass' gradients on the
// entry-by-entry path (`g_rgb * g[0]`, g = loss_matrix[8] * upstream: the same two roundings)
struct ScaleGradParams {
  const float* g_rgb[BTS_MAX_SCALES];
  const float* g_depth[BTS_MAX_SCALES];
  float* gs_rgb[BTS_MAX_SCALES];
  float* gs_depth[BTS_MAX_SCALES];
  float c_rgb[BTS_MAX_SCALES], c_eas[BTS_MAX_SCALES];
  const float* g_loss;   // device scalar or null (1)
  int n_scales;
  long n_rgb, n_depth;   // elements per scale
  // regions the backward accumulates into: the packed parameter gradient, the projected empty feature's, pass C's slot copies (float counts)
  float* zero[2 + BTS_MAX_SCALES];
  long zero_n[2 + BTS_MAX_SCALES];
  int n_zero;
};
__global__ __launch_bounds__(256) void scale_grads_kernel(const ScaleGradParams p) {
  const float up = p.g_loss ? *p.g_loss : 1.0f;
  const int s = blockIdx.y;
  const float a = p.c_rgb[s] * up, b = p.c_eas[s] * up;
  const float* gr = p.g_rgb[s];
  const float* gd = p.g_depth[s];
  float* orr = p.gs_rgb[s];
  float* od = p.gs_depth[s];
  for (long i = blockIdx.x * 256L + threadIdx.x; i < p.n_rgb; i += (long)gridDim.x * 256) orr[i] = gr[i] * a;
  if (gd && od)
    for (long i = blockIdx.x * 256L + threadIdx.x; i < p.n_depth; i += (long)gridDim.x * 256) od[i] = gd[i] * b;
  if (s == 0)      // (one launch instead of a fill per region: each was a dispatch gap of its own on the queue)
    for (int r = 0; r < p.n_zero; ++r)
      for (long i = blockIdx.x * 256L + threadIdx.x; i < p.zero_n[r]; i += (long)gridDim.x * 256) p.zero[r][i] = 0.0f;
}

// ---- learn_empty: the render backward leaves the gradient of the PROJECTED empty feature e_p = w_in[:, :C] . e in d_empty_proj (Hd);
// chain rule on parameter-sized tensors: d e[c] = sum_h w_in[h, c] d e_p[h],  d w_in[h, c] += d e_p[h] e[c]
__global__ __launch_bounds__(256) void empty_grad_kernel(const float* __restrict__ mlp, const float* __restrict__ empty, const float* __restrict__ d_eproj,
                                                       int C, int HD, int d_in, float* __restrict__ d_mlp, float* __restrict__ d_empty) {
  for (int i = threadIdx.x; i < HD * C; i += 256) {
    const int h = i / C, c = i - h * C;
    if (d_mlp) d_mlp[h * d_in + c] += d_eproj[h] * empty[c];
  }
  if (d_empty)
    for (int c = threadIdx.x; c < C; c += 256) {
      float a = 0.0f;
      for (int h = 0; h < HD; ++h) a += mlp[h * d_in + c] * d_eproj[h];
      d_empty[c] = a;
    }
}

// ---- independent scales on independent queues.  A "multiscale" step (trainer.py:220-242; exp_re10k.yaml) renders the same rays against four
// maps: four chains of small kernels (flag, project, render, loss | rows, scatter, dW_pe, reduce, projection backward) that share nothing
// but read-only inputs and the atomically accumulated parameter gradient.  One after the other on one queue every kernel pays its own
// ramp-up and tail (a persistent grid's last waves run alone) and ~45 launch gaps per step; on side queues the chains fill each other's
// tails.  The queues and their fork / join events are created once per host thread and device (handles only, no device memory) and
// are ordered INSIDE the caller's stream: the first kernel of a side chain waits for an event recorded on the caller's stream, the
// caller's stream waits for every chain's last kernel before the call's last kernel -- to the caller the call is still one stream-ordered
// unit.
struct SideQueues {
  hipStream_t q[BTS_MAX_SCALES - 1];
  hipEvent_t fork, first, join[BTS_MAX_SCALES - 1];
  bool ok;
};
static SideQueues* side_queues() {
  static thread_local SideQueues* per_dev[16] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!per_dev[dev]) {
    SideQueues* sq = new SideQueues;
    sq->ok = hipEventCreateWithFlags(&sq->fork, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&sq->first, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < BTS_MAX_SCALES - 1; ++i)
      sq->ok = sq->ok && hipStreamCreateWithFlags(&sq->q[i], hipStreamNonBlocking) == hipSuccess &&
               hipEventCreateWithFlags(&sq->join[i], hipEventDisableTiming) == hipSuccess;
    per_dev[dev] = sq;
  }
  return per_dev[dev]->ok ? per_dev[dev] : nullptr;
}

static long map_texels(const BtsTrainStep* st, int s) { return (long)(st->cfg.H >> st->scale[s].feat_shift) * (st->cfg.W >> st->scale[s].feat_shift); }

static int check_step(const BtsTrainStep* st, const char* who, bool bwd) {
  if (!st) {
    set_error("%s: NULL step", who);
    return BTS_E_INVALID;
  }
  const BtsFieldCfg& c = st->cfg;
  if (c.n <= 0 || c.H <= 0 || c.W <= 0 || st->v <= 0 || st->P <= 0 || st->ph <= 0 || st->pw <= 0 || st->K <= 0 || st->ph * st->pw > 64 ||
      st->ph > c.H || st->pw > c.W) {
    set_error("%s: non-positive size, or a patch of more than 64 pixels / larger than the frame (n=%ld, P=%ld, K=%ld)", who, c.n, st->P, st->K);
    return BTS_E_INVALID;
  }
  if (!shape_supported(c.C, c.d_hidden, c.n_blocks) || c.num_freqs != kNumFreqs || c.nv < 1 || c.nv > BTS_MAX_VIEWS) {
    set_error("%s: configuration outside the compiled envelope (C=%ld d_hidden=%ld n_blocks=%ld; needs num_freqs=6, 1 <= nv <= 8)", who, c.C, c.d_hidden,
              c.n_blocks);
    return BTS_E_UNSUPPORTED;
  }
  if (st->n_scales < 1 || st->n_scales > BTS_MAX_SCALES || st->n_loss < 1 || st->n_loss > BTS_MAX_LOSS_VIEWS) {
    set_error("%s: n_scales=%ld must be 1..4 and n_loss=%ld 1..16", who, st->n_scales, st->n_loss);
    return BTS_E_INVALID;
  }
  if (st->invalid_policy < 0 || st->invalid_policy > 2 || (c.code_mode != 0 && c.code_mode != 1) || c.enc_render_view < -1 || c.enc_render_view >= c.nv) {
    set_error("%s: invalid_policy=%ld / code_mode=%ld / enc_render_view=%ld out of range", who, st->invalid_policy, c.code_mode, c.enc_render_view);
    return BTS_E_INVALID;
  }
  bool ids_ok = st->id_encoder >= 0 && st->id_encoder < st->v;
  for (int j = 0; j < c.nv; ++j) ids_ok = ids_ok && st->ids_render[j] >= 0 && st->ids_render[j] < st->v;
  for (int j = 0; j < st->n_loss; ++j) ids_ok = ids_ok && st->ids_loss[j] >= 0 && st->ids_loss[j] < st->v;
  if (!ids_ok) {
    set_error("%s: a frame id outside [0, v=%ld)", who, st->v);
    return BTS_E_INVALID;
  }
  if (!st->images || !st->Ks || !st->poses_c2w || !st->patch_v || !st->patch_y || !st->patch_x || !st->mlp_params || !st->rays || !st->rgb_gt ||
      !st->loss_vals || !st->cams || !st->imgs_nhwc4 || (c.learn_empty && !st->empty_feature)) {
    set_error("%s: NULL input / output / scratch pointer", who);
    return BTS_E_INVALID;
  }
  for (int s = 0; s < st->n_scales; ++s) {
    const BtsTrainScale& q = st->scale[s];
    const int fs = q.feat_shift;
    if (fs < 0 || fs > 6 || (c.H & ((1 << fs) - 1)) || (c.W & ((1 << fs) - 1))) {
      set_error("%s: scale %ld: feat_shift=%ld needs 0..6 and H, W multiples of 2^feat_shift", who, s, fs);
      return BTS_E_INVALID;
    }
    if ((reinterpret_cast<uintptr_t>(q.sampled_tiles) & 3) != 0) {
      set_error("%s: scale %ld: sampled_tiles must be 4-byte aligned (the hand-over kernel clears the flags word-wise)", who, s);
      return BTS_E_INVALID;
    }
    if (!q.feat_nchw || !q.jitter || !q.rgb || !q.depth || !q.invalid_wsum || !q.invalid_any || !q.proj_nhwc || !q.sampled_tiles || !q.z_samp ||
        !q.sigma_raw || !q.trans || !q.rgb_samps || !q.loss_parts || !q.g_rgb || !q.g_depth) {
      set_error("%s: scale %ld: NULL pointer", who, s);
      return BTS_E_INVALID;
    }
    if (bwd && (!q.gs_rgb || !q.gs_depth || !q.d_proj_nhwc || !q.d_proj_tiles)) {
      set_error("%s: scale %ld: NULL backward pointer (gs_rgb, gs_depth, d_proj_nhwc, d_proj_tiles)", who, s);
      return BTS_E_INVALID;
    }
  }
  return BTS_OK;
}

// the per-scale views of the step as the single-call entry points take them
struct ScaleView {
  BtsFieldCfg cfg;
  BtsFieldTensors t;
  BtsRenderArgs a;
};
static ScaleView scale_view(const BtsTrainStep* st, int s) {
  ScaleView v;
  memset(&v, 0, sizeof(v));
  const BtsTrainScale& q = st->scale[s];
  const int n = st->cfg.n, nv = st->cfg.nv;
  v.cfg = st->cfg;
  v.cfg.feat_shift = q.feat_shift;
  // the scale's tile geometry: 16 x 4 blocks for a channels-last map, runs of 64 texels for an NCHW one (BtsFieldCfg.tile_blocks: what each
  // layout is faster with); the step's own flag arrays never leave the two calls, so the step decides for itself
  v.cfg.tile_blocks = q.feat_channels_last != 0;
  float* K_enc = st->cams;
  float* w2c_enc = K_enc + (long)n * 9;
  float* K_r = w2c_enc + (long)n * 16;
  float* w2c_r = K_r + (long)n * nv * 9;
  v.t.proj_nhwc = q.proj_nhwc, v.t.K_enc = K_enc, v.t.w2c_enc = w2c_enc, v.t.imgs_nhwc4 = st->imgs_nhwc4, v.t.K_r = K_r, v.t.w2c_r = w2c_r;
  v.t.empty_feature = st->cfg.learn_empty ? st->empty_feature : nullptr, v.t.mlp_params = st->mlp_params;
  v.a.rays_per_sample = st->P * st->ph * st->pw, v.a.K = st->K, v.a.hard_alpha_cap = st->hard_alpha_cap, v.a.white_bkgd = 0;
  v.a.rays = st->rays, v.a.lindisp = st->lindisp;
  return v;
}

int train_step_fwd_impl(const BtsTrainStep* st, hipStream_t main_stream) {
  hipStream_t stream = main_stream;
  const BtsFieldCfg& c = st->cfg;
  const int n = c.n, nv = c.nv, Bp = st->P * st->ph * st->pw;
  unsigned char* zero[BTS_MAX_SCALES];
  long zero_bytes[BTS_MAX_SCALES];
  for (int s = 0; s < st->n_scales; ++s) zero[s] = st->scale[s].sampled_tiles, zero_bytes[s] = (long)n * ((map_texels(st, s) + 63) / 64);
  int rc = handover_launch(st->Ks, st->poses_c2w, st->images, st->patch_v, st->patch_y, st->patch_x, n, st->v, st->id_encoder, nv, st->ids_render, st->n_loss,
                           st->ids_loss, c.H, c.W, st->P, st->ph, st->pw, st->z_near, st->z_far, st->img_scale, st->img_shift, st->cams, st->imgs_nhwc4,
                           st->rays, st->rgb_gt, st->n_scales, zero, zero_bytes, stream);
  if (rc) {
    set_error("%s: the hand-over kernel launch failed", "bts_train_step_fwd");
    return rc;
  }
  FinishParams fin;
  memset(&fin, 0, sizeof(fin));
  SideQueues* sq = (st->concurrent_scales && st->n_scales > 1) ? side_queues() : nullptr;
  if (sq) {
    if (hipEventRecord(sq->fork, main_stream) != hipSuccess) return BTS_E_LAUNCH;
    for (int s = 1; s < st->n_scales; ++s)
      if (hipStreamWaitEvent(sq->q[s - 1], sq->fork, 0) != hipSuccess) return BTS_E_LAUNCH;
  }
  // one slice of the (idle) backward workspace per scale for the projection's tile list
  const size_t fwd_ws = st->bwd_workspace ? (st->bwd_workspace_bytes / (size_t)st->n_scales) & ~(size_t)255 : 0;
  for (int s = 0; s < st->n_scales; ++s) {
    hipStream_t stream = (sq && s > 0) ? sq->q[s - 1] : main_stream;
    const BtsTrainScale& q = st->scale[s];
    ScaleView v = scale_view(st, s);
    const long texels = map_texels(st, s);
    rc = mark_tiles_impl(st->rays, nullptr, q.jitter, v.t.w2c_enc, v.t.K_enc, (long)n * Bp, Bp, st->K, st->lindisp, c.H, c.W, q.feat_shift, q.sampled_tiles,
                         stream, v.cfg.tile_blocks);
    if (!rc) rc = project_features_impl(c.C, c.d_hidden, q.feat_nchw, st->mlp_params, n, (int)texels, q.proj_nhwc, q.sampled_tiles, stream, q.feat_channels_last != 0,
                                        v.cfg.tile_blocks ? c.W >> q.feat_shift : 0,
                                        // the backward's workspace is idle during the forward: the scale's slice of it holds the tile list
                                        fwd_ws ? static_cast<char*>(st->bwd_workspace) + fwd_ws * (size_t)s : nullptr, fwd_ws);
    if (rc) {
      set_error("%s: projection launch failed at scale %ld", "bts_train_step_fwd", s);
      return rc;
    }
    v.a.jitter = q.jitter, v.a.z_samp_out = q.z_samp;
    v.a.rgb = q.rgb, v.a.depth = q.depth, v.a.rgb_samps = q.rgb_samps, v.a.sigma_raw = q.sigma_raw, v.a.trans = q.trans;
    v.a.invalid_wsum = q.invalid_wsum, v.a.invalid_any = q.invalid_any;
    rc = render_fwd_impl(&v.cfg, &v.t, &v.a, stream);
    if (rc) return rc;
    // the invalid-ray mask of EVERY scale's term comes from scale 0's render (loss.py:100-118 reads data["coarse"][0]): the side chains'
    // loss passes wait for it
    if (sq && s == 0 && hipEventRecord(sq->first, main_stream) != hipSuccess) return BTS_E_LAUNCH;
    if (sq && s > 0 && st->invalid_policy != 0 && hipStreamWaitEvent(stream, sq->first, 0) != hipSuccess) return BTS_E_LAUNCH;
    BtsLossArgs la;
    memset(&la, 0, sizeof(la));
    la.rgb = q.rgb, la.depth = st->edge_aware_smoothness ? q.depth : nullptr, la.rgb_gt = st->rgb_gt, la.parts = q.loss_parts;
    la.g_rgb = q.g_rgb, la.g_depth = st->edge_aware_smoothness ? q.g_depth : nullptr;
    la.n_patches = n * st->P, la.patch_h = st->ph, la.patch_w = st->pw, la.nv = nv, la.K = 0;
    la.invalid_policy = st->invalid_policy, la.edge_aware_smoothness = st->edge_aware_smoothness, la.scale_rgb = 1.0f, la.scale_eas = 1.0f;
    la.invalid_wsum = st->invalid_policy == 2 ? st->scale[0].invalid_wsum : nullptr;
    la.invalid_any = st->invalid_policy == 1 ? st->scale[0].invalid_any : nullptr;
    rc = photometric_loss_impl(&la, stream);
    if (rc) return rc;
    fin.parts[s] = q.loss_parts;
    if (sq && s > 0 && (hipEventRecord(sq->join[s - 1], stream) != hipSuccess || hipStreamWaitEvent(main_stream, sq->join[s - 1], 0) != hipSuccess))
      return BTS_E_LAUNCH;
  }
  stream = main_stream;
  fin.n_scales = st->n_scales, fin.n_patches = n * st->P, fin.out = st->loss_vals;
  memcpy(fin.M, st->loss_matrix, sizeof(float) * 9 * 3 * st->n_scales);
  loss_finish_kernel<<<1, 256, 0, stream>>>(fin);
  if (hipGetLastError() != hipSuccess) {
    set_error("%s: loss reduction launch failed", "bts_train_step_fwd");
    return BTS_E_LAUNCH;
  }
  return BTS_OK;
}

int train_step_bwd_impl(const BtsTrainStep* st, const float* g_loss, hipStream_t main_stream) {
  hipStream_t stream = main_stream;
  const BtsFieldCfg& c = st->cfg;
  const int n = c.n, nv = c.nv, Bp = st->P * st->ph * st->pw;
  const int cols = 3 * st->n_scales;
  const long n_params = MlpLayout{c.C + 3 + 6 * c.num_freqs, c.d_hidden, c.n_blocks}.total();
  ScaleGradParams sg;
  memset(&sg, 0, sizeof(sg));
  for (int s = 0; s < st->n_scales; ++s) {
    const BtsTrainScale& q = st->scale[s];
    sg.g_rgb[s] = q.g_rgb, sg.g_depth[s] = st->edge_aware_smoothness ? q.g_depth : nullptr, sg.gs_rgb[s] = q.gs_rgb, sg.gs_depth[s] = q.gs_depth;
    sg.c_rgb[s] = st->loss_matrix[8 * cols + 3 * s], sg.c_eas[s] = st->loss_matrix[8 * cols + 3 * s + 1];
  }
  sg.g_loss = g_loss, sg.n_scales = st->n_scales, sg.n_rgb = (long)n * Bp * nv * 3, sg.n_depth = (long)n * Bp;
  // one workspace slice per scale when the scales run side by side (the passes of a scale hand each other its contents)
  size_t ws_each = 0;
  ScaleView v0 = scale_view(st, 0);
  ws_each = (render_bwd_workspace_impl(&v0.cfg, &v0.a) + 255) & ~(size_t)255;
  if (ws_each > st->bwd_workspace_bytes || (ws_each && !st->bwd_workspace)) {
    set_error("%s: bwd_workspace too small (%ld bytes needed)", "bts_train_step_bwd", (long)ws_each);
    return BTS_E_WORKSPACE;
  }
  SideQueues* sq = (st->concurrent_scales && st->n_scales > 1 && st->bwd_workspace_bytes >= ws_each * (size_t)st->n_scales) ? side_queues() : nullptr;
  const bool want_empty = c.learn_empty && st->d_empty_proj && (st->d_mlp_params || st->d_empty_feature);
  if (st->d_mlp_params) sg.zero[sg.n_zero] = st->d_mlp_params, sg.zero_n[sg.n_zero++] = n_params;
  if (want_empty) sg.zero[sg.n_zero] = st->d_empty_proj, sg.zero_n[sg.n_zero++] = c.d_hidden;
  for (int s = 0; s < (sq ? st->n_scales : 1); ++s) {     // pass C's slot copies: the reduce kernel leaves them zero again, one fill serves every scale
    float* fp;
    size_t fb;
    render_bwd_flush_region(&v0.cfg, &v0.a, static_cast<char*>(st->bwd_workspace) + ws_each * (size_t)s, &fp, &fb);
    sg.zero[sg.n_zero] = fp, sg.zero_n[sg.n_zero++] = (long)(fb / sizeof(float));
  }
  const long want = (sg.n_rgb + 255) / 256;
  scale_grads_kernel<<<dim3((unsigned)(want < 1024 ? want : 1024), (unsigned)st->n_scales), 256, 0, stream>>>(sg);
  if (hipGetLastError() != hipSuccess) {
    set_error("%s: gradient scaling launch failed", "bts_train_step_bwd");
    return BTS_E_LAUNCH;
  }
  if (sq) {
    if (hipEventRecord(sq->fork, main_stream) != hipSuccess) return BTS_E_LAUNCH;
    for (int s = 1; s < st->n_scales; ++s)
      if (hipStreamWaitEvent(sq->q[s - 1], sq->fork, 0) != hipSuccess) return BTS_E_LAUNCH;
  }
  for (int s = 0; s < st->n_scales; ++s) {
    stream = (sq && s > 0) ? sq->q[s - 1] : main_stream;
    const BtsTrainScale& q = st->scale[s];
    ScaleView v = scale_view(st, s);
    v.a.z_samp = q.z_samp, v.a.sigma_raw = q.sigma_raw, v.a.trans = q.trans, v.a.rgb_samps = q.rgb_samps;
    BtsRenderGrads g;
    memset(&g, 0, sizeof(g));
    g.g_rgb = q.gs_rgb, g.g_depth = st->edge_aware_smoothness ? q.gs_depth : nullptr;
    const bool need_map = q.d_feat_nchw != nullptr || st->d_mlp_params != nullptr;
    g.d_proj_nhwc = need_map ? q.d_proj_nhwc : nullptr, g.d_proj_tiles = need_map ? q.d_proj_tiles : nullptr;
    g.d_mlp_params = st->d_mlp_params, g.d_empty_proj = want_empty ? st->d_empty_proj : nullptr;
    const size_t need = render_bwd_workspace_impl(&v.cfg, &v.a);
    void* ws = sq ? static_cast<void*>(static_cast<char*>(st->bwd_workspace) + ws_each * (size_t)s) : st->bwd_workspace;
    const size_t ws_bytes = sq ? ws_each : st->bwd_workspace_bytes;
    if (need > ws_bytes || (need && !st->bwd_workspace)) {
      set_error("%s: bwd_workspace too small (%ld bytes needed)", "bts_train_step_bwd", (long)need);
      return BTS_E_WORKSPACE;
    }
    int rc = render_bwd_impl(&v.cfg, &v.t, &v.a, &g, ws, ws_bytes, stream, true);
    if (rc) return rc;
    if (need_map) {
      float* slots;
      size_t slot_bytes;
      render_bwd_flush_region(&v.cfg, &v.a, ws, &slots, &slot_bytes);   // (pass C's slot copies at the END of the slice stay zero between the scales)
      rc = project_features_bwd_tiles_impl(c.C, c.d_hidden, q.feat_nchw, q.d_proj_nhwc, q.d_proj_tiles, st->mlp_params, n, (int)map_texels(st, s),
                                           q.d_feat_nchw, st->d_mlp_params, 1, stream, q.feat_channels_last != 0, v.cfg.tile_blocks ? c.W >> q.feat_shift : 0,
                                           // the scale's slice of the backward workspace is dead by now (everything bts_render_bwd parked there was
                                           // read by its own passes, enqueued above on this stream) except pass C's slot copies at its END
                                           ws, need > slot_bytes ? need - slot_bytes : 0);
      if (rc) {
        set_error("%s: projection backward launch failed at scale %ld", "bts_train_step_bwd", s);
        return rc;
      }
    }
    if (sq && s > 0 && (hipEventRecord(sq->join[s - 1], stream) != hipSuccess || hipStreamWaitEvent(main_stream, sq->join[s - 1], 0) != hipSuccess))
      return BTS_E_LAUNCH;
  }
  stream = main_stream;
  if (want_empty) {
    empty_grad_kernel<<<1, 256, 0, stream>>>(st->mlp_params, st->empty_feature, st->d_empty_proj, c.C, c.d_hidden, c.C + 3 + 6 * c.num_freqs, st->d_mlp_params,
                                             st->d_empty_feature);
    if (hipGetLastError() != hipSuccess) {
      set_error("%s: empty-feature gradient launch failed", "bts_train_step_bwd");
      return BTS_E_LAUNCH;
    }
  }
  return BTS_OK;
}

int gen_rays_launch(const float* poses, const float* projs, int V, int H, int W, float zn, float zf, int norm_dir, float* rays, hipStream_t s);
int distance_to_z_launch(const float* depths, const float* invK, int N, int H, int W, float* out, hipStream_t s);
int invert_small_launch(const float* src, float* dst, int N, int dim, hipStream_t s);
int project_features_plain(int C, int HD, const float* feat, const float* mlp, int N, int HW, float* proj, hipStream_t s, bool channels_last) {
  return project_features_impl(C, HD, feat, mlp, N, HW, proj, nullptr, s, channels_last);
}

int eval_frame_impl(const BtsEvalFrame* f, hipStream_t stream) {
  const BtsFieldCfg& c = f->cfg;
  const int n = c.n, nv = c.nv;
  int rc = camera_prep_launch(f->Ks, f->poses_c2w, n, f->v, f->id_encoder, nv, f->ids_render, f->cams, stream);
  if (!rc && nv) rc = pack_rgb_views_launch(f->images, f->imgs_nhwc4, n, f->v, nv, f->ids_render, c.H, c.W, f->img_scale, f->img_shift, stream);
  if (!rc) rc = project_features_plain(c.C, c.d_hidden, f->feat_nchw, f->mlp_params, n, c.H * c.W, f->proj_nhwc, stream, f->feat_channels_last != 0);
  if (!rc) rc = gen_rays_launch(f->poses_c2w, f->Ks, n * f->v, c.H, c.W, f->z_near, f->z_far, f->norm_dir, f->rays, stream);
  if (rc) {
    set_error("%s: a hand-over kernel launch failed", "bts_eval_frame");
    return rc;
  }
  BtsFieldCfg cfg = c;
  cfg.feat_shift = 0;
  BtsFieldTensors t;
  memset(&t, 0, sizeof(t));
  float* K_enc = f->cams;
  float* w2c_enc = K_enc + (long)n * 9;
  float* K_r = w2c_enc + (long)n * 16;
  float* w2c_r = K_r + (long)n * nv * 9;
  t.proj_nhwc = f->proj_nhwc, t.K_enc = K_enc, t.w2c_enc = w2c_enc, t.imgs_nhwc4 = f->imgs_nhwc4, t.K_r = K_r, t.w2c_r = w2c_r;
  t.empty_feature = c.learn_empty ? f->empty_feature : nullptr, t.mlp_params = f->mlp_params;
  BtsRenderArgs a;
  memset(&a, 0, sizeof(a));
  a.rays_per_sample = f->v * c.H * c.W, a.K = f->K, a.hard_alpha_cap = f->hard_alpha_cap, a.rays = f->rays, a.jitter = f->jitter, a.lindisp = f->lindisp;
  a.rgb = f->rgb, a.depth = f->depth, a.weights = f->weights, a.alphas = f->alphas, a.invalid = f->invalid;
  rc = render_fwd_impl(&cfg, &t, &a, stream);
  if (rc) return rc;
  if (f->depth_z) {
    rc = invert_small_launch(f->Ks, f->inv_K, n * f->v, 3, stream);
    if (!rc) rc = distance_to_z_launch(f->depth, f->inv_K, n * f->v, c.H, c.W, f->depth_z, stream);
    if (rc) {
      set_error("%s: distance_to_z launch failed", "bts_eval_frame");
      return rc;
    }
  }
  return BTS_OK;
}

}  // namespace bts

using namespace bts;

extern "C" {

int bts_eval_frame(const BtsEvalFrame* f, void* stream) {
  if (!f) {
    set_error("%s: NULL frame", "bts_eval_frame");
    return BTS_E_INVALID;
  }
  const BtsFieldCfg& c = f->cfg;
  if (c.n <= 0 || c.H <= 0 || c.W <= 0 || f->v <= 0 || f->K <= 0 || c.nv < 0 || c.nv > BTS_MAX_VIEWS) {
    set_error("%s: non-positive size or more than 8 render views (n=%ld, v=%ld, K=%ld)", "bts_eval_frame", c.n, f->v, f->K);
    return BTS_E_INVALID;
  }
  if (!shape_supported(c.C, c.d_hidden, c.n_blocks) || c.num_freqs != kNumFreqs) {
    set_error("%s: configuration outside the compiled envelope (C=%ld d_hidden=%ld n_blocks=%ld)", "bts_eval_frame", c.C, c.d_hidden, c.n_blocks);
    return BTS_E_UNSUPPORTED;
  }
  bool ids_ok = f->id_encoder >= 0 && f->id_encoder < f->v;
  for (int j = 0; j < c.nv; ++j) ids_ok = ids_ok && f->ids_render[j] >= 0 && f->ids_render[j] < f->v;
  if (!ids_ok || c.enc_render_view < -1 || c.enc_render_view >= c.nv) {
    set_error("%s: a frame id outside [0, v=%ld) or enc_render_view out of range", "bts_eval_frame", f->v);
    return BTS_E_INVALID;
  }
  if (!f->images || !f->Ks || !f->poses_c2w || !f->feat_nchw || !f->mlp_params || !f->jitter || !f->cams || !f->proj_nhwc || !f->rays || !f->rgb ||
      !f->depth || (c.nv > 0 && !f->imgs_nhwc4) || (f->depth_z && !f->inv_K) || (c.learn_empty && !f->empty_feature)) {
    set_error("%s: NULL input / output / scratch pointer", "bts_eval_frame");
    return BTS_E_INVALID;
  }
  if ((long)c.n * f->v * c.H * c.W > 0x7FF00000L) {
    set_error("%s: too many rays in one call (%ld)", "bts_eval_frame", (long)c.n * f->v * c.H * c.W);
    return BTS_E_UNSUPPORTED;
  }
  return eval_frame_impl(f, (hipStream_t)stream);
}

int bts_train_step_fwd(const BtsTrainStep* st, void* stream) {
  if (int rc = check_step(st, "bts_train_step_fwd", false)) return rc;
  return train_step_fwd_impl(st, (hipStream_t)stream);
}

int bts_train_step_bwd(const BtsTrainStep* st, const float* g_loss, void* stream) {
  if (int rc = check_step(st, "bts_train_step_bwd", true)) return rc;
  return train_step_bwd_impl(st, g_loss, (hipStream_t)stream);
}

}  // extern "C"
