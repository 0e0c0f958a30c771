// extern "C" entry points of libbts_render.so (declared in include/bts_render.h): argument validation and launch
// geometry only -- no allocation, no synchronisation, no exceptions.
#include "bts_common.h"

#include <cstdio>
#include <cstring>

namespace bts {
struct FwdParams;
void set_error(const char* fmt, const char* a = "", long b = 0, long c = 0, long d = 0);
const char* last_error();
bool shape_supported(int C, int HD, int NB);

int transpose_launch(const float* src, float* dst, int N, int C, int H, int W, bool to_nhwc, hipStream_t s);
int pack_rgb_launch(const float* src, float* dst, int N, int H, int W, float scale, float shift, hipStream_t s);
int patch_rays_launch(const float* poses, const float* projs, const float* images, const int* pv, const int* py, const int* px, int n, int v,
                      int c, int H, int W, int P, int ph, int pw, float zn, float zf, int norm_dir, float* rays, float* gt, hipStream_t s);
int photometric_loss_impl(const BtsLossArgs* a, hipStream_t s);
int gen_rays_launch(const float* poses, const float* projs, int V, int H, int W, float zn, float zf, int norm_dir, float* rays,
                    hipStream_t s);
int sample_coarse_launch(const float* rays, const float* u, long B, int K, int lindisp, float* z, hipStream_t s);
int distance_to_z_launch(const float* depths, const float* invK, int N, int H, int W, float* out, hipStream_t s);
int invert_small_launch(const float* src, float* dst, int N, int dim, hipStream_t s);

int project_features_impl(int C, int HD, const float* feat, const float* mlp, int N, int HW, float* proj, const unsigned char* tiles, hipStream_t s,
                          bool feat_cl = false, int Wm = 0,     // Wm: the map's width when `tiles` are 16 x 4 blocks (BtsFieldCfg.tile_blocks), 0 = runs of 64 texels
                          void* list_ws = nullptr, size_t list_ws_bytes = 0);   // scratch for the balanced (list-driven) form
int mark_tiles_impl(const float* rays, const float* z_samp, const float* jitter, const float* w2c_enc, const float* K_enc, long B, int Bp, int K, int lindisp,
                    int H, int W, int fs, unsigned char* tiles, hipStream_t s, int blocks);
int project_features_bwd_tiles_impl(int C, int HD, const float* feat, float* dproj, unsigned char* tiles, const float* mlp, int N, int HW, float* dfeat,
                                    float* d_mlp, int clear, hipStream_t s, bool feat_cl = false, int Wm = 0, void* list_ws = nullptr,
                                    size_t list_ws_bytes = 0);   // list_ws: scratch for the balanced (list-driven) form, project_bwd_list_bytes(N * tiles) bytes
int project_features_bwd_impl(int C, int HD, const float* feat, const float* dproj, const float* mlp, int N, int HW, float* dfeat,
                              float* d_mlp, hipStream_t s);
int render_fwd_impl(const BtsFieldCfg* cfg, const BtsFieldTensors* t, const BtsRenderArgs* a, hipStream_t s);
int field_query_impl(const BtsFieldCfg* cfg, const BtsFieldTensors* t, const float* xyz, int P, int only_density, float* rgb,
                     float* invalid, float* sigma, hipStream_t s);
int occupancy_profile_impl(const BtsFieldCfg* cfg, const BtsFieldTensors* t, const float* xyz, int Y, int cols, float threshold,
                           int only_density, float* profile, float* sigma, hipStream_t s);
size_t render_bwd_workspace_impl(const BtsFieldCfg* cfg, const BtsRenderArgs* a);
int render_bwd_impl(const BtsFieldCfg* cfg, const BtsFieldTensors* t, const BtsRenderArgs* a, const BtsRenderGrads* g, void* ws,
                    size_t ws_bytes, hipStream_t s, bool flush_clean = false);
}  // namespace bts

using namespace bts;

// feat_shift: 0 .. 6, H and W multiples of 2^feat_shift (the small map then IS the nearest-neighbour source of the H x W one)
static int check_shift(const BtsFieldCfg* cfg, const char* who) {
  const int fs = cfg->feat_shift;
  if (fs < 0 || fs > 6 || (cfg->H & ((1 << fs) - 1)) != 0 || (cfg->W & ((1 << fs) - 1)) != 0) {
    set_error("%s: feat_shift=%ld needs 0 <= feat_shift <= 6 and H=%ld, W=%ld multiples of 2^feat_shift", who, (long)fs, (long)cfg->H, (long)cfg->W);
    return BTS_E_INVALID;
  }
  return BTS_OK;
}
static long feat_texels(const BtsFieldCfg* cfg) { return (long)(cfg->H >> cfg->feat_shift) * (cfg->W >> cfg->feat_shift); }

static int check_cfg(const BtsFieldCfg* cfg, const BtsFieldTensors* t, bool need_imgs) {
  if (!cfg || !t) {
    set_error("%s: NULL cfg/tensors", "bts");
    return BTS_E_INVALID;
  }
  if (cfg->n <= 0 || cfg->H <= 0 || cfg->W <= 0 || cfg->nv < 0) {
    set_error("%s: non-positive size n=%ld H=%ld W=%ld", "bts", cfg->n, cfg->H, cfg->W);
    return BTS_E_INVALID;
  }
  if (!bts_supported(cfg)) {
    set_error("%s: configuration outside the compiled envelope (C=%ld d_hidden=%ld n_blocks=%ld; also needs num_freqs=6, nv<=8)",
              "bts", cfg->C, cfg->d_hidden, cfg->n_blocks);
    return BTS_E_UNSUPPORTED;
  }
  if (int rc = check_shift(cfg, "bts")) return rc;
  if (cfg->feat_shift && !t->proj_nhwc) {
    set_error("%s: feat_shift > 0 needs the projected map (proj_nhwc)", "bts");
    return BTS_E_INVALID;
  }
  if ((!t->feat_nhwc && !t->proj_nhwc) || !t->K_enc || !t->w2c_enc || !t->mlp_params) {
    set_error("%s: NULL field tensor", "bts");
    return BTS_E_INVALID;
  }
  if (need_imgs && cfg->nv > 0 && (!t->imgs_nhwc4 || !t->K_r || !t->w2c_r)) {
    set_error("%s: NULL colour-view tensor with nv=%ld", "bts", cfg->nv);
    return BTS_E_INVALID;
  }
  if (cfg->learn_empty && !t->empty_feature) {
    set_error("%s: learn_empty set but empty_feature is NULL", "bts");
    return BTS_E_INVALID;
  }
  if (cfg->enc_render_view < -1 || cfg->enc_render_view >= cfg->nv) {
    set_error("%s: enc_render_view=%ld must be -1 or a render view index below nv=%ld", "bts", cfg->enc_render_view, cfg->nv);
    return BTS_E_INVALID;
  }
  if (cfg->code_mode != 0 && cfg->code_mode != 1) {
    set_error("%s: unknown code_mode %ld", "bts", cfg->code_mode);
    return BTS_E_INVALID;
  }
  return BTS_OK;
}

extern "C" {

int bts_abi_version(void) { return BTS_ABI_VERSION; }
const char* bts_last_error(void) { return last_error(); }

int bts_supported(const BtsFieldCfg* cfg) {
  if (!cfg) return 0;
  return shape_supported(cfg->C, cfg->d_hidden, cfg->n_blocks) && cfg->num_freqs == kNumFreqs && cfg->nv <= BTS_MAX_VIEWS ? 1 : 0;
}

int64_t bts_mlp_param_count(const BtsFieldCfg* cfg) {
  if (!cfg) return -1;
  return MlpLayout{cfg->C + 3 + 6 * cfg->num_freqs, cfg->d_hidden, cfg->n_blocks}.total();
}

int bts_render_fwd(const BtsFieldCfg* cfg, const BtsFieldTensors* t, const BtsRenderArgs* a, void* stream) {
  int rc = check_cfg(cfg, t, true);
  if (rc) return rc;
  if (!a || !a->rays || (!a->z_samp && !a->jitter) || !a->rgb || !a->depth) {
    set_error("%s: NULL render argument (rays, z_samp or jitter, rgb and depth are required)", "bts_render_fwd");
    return BTS_E_INVALID;
  }
  if (a->rays_per_sample <= 0 || a->K <= 0) {
    set_error("%s: non-positive rays_per_sample=%ld K=%ld", "bts_render_fwd", a->rays_per_sample, a->K);
    return BTS_E_INVALID;
  }
  return render_fwd_impl(cfg, t, a, (hipStream_t)stream);
}

size_t bts_render_bwd_workspace(const BtsFieldCfg* cfg, const BtsRenderArgs* a) {
  if (!cfg || !a) return 0;
  return render_bwd_workspace_impl(cfg, a);
}

int bts_render_bwd(const BtsFieldCfg* cfg, const BtsFieldTensors* t, const BtsRenderArgs* a, const BtsRenderGrads* g,
                   void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_cfg(cfg, t, true);
  if (rc) return rc;
  if (!a || !g || !a->rays || !a->z_samp || !a->sigma_raw || !a->trans || !t->proj_nhwc) {
    set_error("%s: NULL argument (rays, z_samp, the forward's sigma_raw + trans and proj_nhwc are required)", "bts_render_bwd");
    return BTS_E_INVALID;
  }
  if (a->rays_per_sample <= 0 || a->K <= 0) {
    set_error("%s: non-positive rays_per_sample=%ld K=%ld", "bts_render_bwd", a->rays_per_sample, a->K);
    return BTS_E_INVALID;
  }
  if (workspace_bytes < render_bwd_workspace_impl(cfg, a) || (!workspace && render_bwd_workspace_impl(cfg, a) > 0)) {
    set_error("%s: workspace too small (%ld bytes needed)", "bts_render_bwd", (long)render_bwd_workspace_impl(cfg, a));
    return BTS_E_WORKSPACE;
  }
  return render_bwd_impl(cfg, t, a, g, workspace, workspace_bytes, (hipStream_t)stream);
}

int bts_project_features(const BtsFieldCfg* cfg, const float* feat_nchw, const float* mlp_params, int32_t N, float* proj_nhwc,
                         void* stream) {
  if (!cfg || !feat_nchw || !mlp_params || !proj_nhwc || N <= 0 || cfg->H <= 0 || cfg->W <= 0) {
    set_error("%s: NULL pointer or non-positive size", "bts_project_features");
    return BTS_E_INVALID;
  }
  if (!bts_supported(cfg)) {
    set_error("%s: configuration outside the compiled envelope (C=%ld d_hidden=%ld n_blocks=%ld)", "bts_project_features", cfg->C,
              cfg->d_hidden, cfg->n_blocks);
    return BTS_E_UNSUPPORTED;
  }
  if (int rc = check_shift(cfg, "bts_project_features")) return rc;
  int rc = project_features_impl(cfg->C, cfg->d_hidden, feat_nchw, mlp_params, N, (int)feat_texels(cfg), proj_nhwc, nullptr, (hipStream_t)stream);
  if (rc) set_error("%s: kernel launch failed", "bts_project_features");
  return rc;
}

int bts_project_features_tiles(const BtsFieldCfg* cfg, const float* feat_nchw, const float* mlp_params, int32_t N, const uint8_t* tiles, float* proj_nhwc,
                               void* stream) {
  if (!cfg || !feat_nchw || !mlp_params || !proj_nhwc || !tiles || N <= 0 || cfg->H <= 0 || cfg->W <= 0) {
    set_error("%s: NULL pointer or non-positive size", "bts_project_features_tiles");
    return BTS_E_INVALID;
  }
  if (!bts_supported(cfg)) {
    set_error("%s: configuration outside the compiled envelope (C=%ld d_hidden=%ld n_blocks=%ld)", "bts_project_features_tiles", cfg->C, cfg->d_hidden,
              cfg->n_blocks);
    return BTS_E_UNSUPPORTED;
  }
  if (int rc = check_shift(cfg, "bts_project_features_tiles")) return rc;
  int rc = project_features_impl(cfg->C, cfg->d_hidden, feat_nchw, mlp_params, N, (int)feat_texels(cfg), proj_nhwc, tiles, (hipStream_t)stream, false,
                                 cfg->tile_blocks ? cfg->W >> cfg->feat_shift : 0);
  if (rc) set_error("%s: kernel launch failed", "bts_project_features_tiles");
  return rc;
}

int bts_mark_sampled_tiles(const BtsFieldCfg* cfg, const float* K_enc, const float* w2c_enc, const BtsRenderArgs* a, uint8_t* tiles, void* stream) {
  if (!cfg || !K_enc || !w2c_enc || !a || !tiles || !a->rays || (!a->z_samp && !a->jitter) || cfg->n <= 0 || cfg->H <= 0 || cfg->W <= 0 ||
      a->rays_per_sample <= 0 || a->K <= 0) {
    set_error("%s: NULL pointer, non-positive size, or neither z_samp nor jitter", "bts_mark_sampled_tiles");
    return BTS_E_INVALID;
  }
  if (int rc = check_shift(cfg, "bts_mark_sampled_tiles")) return rc;
  int rc = mark_tiles_impl(a->rays, a->z_samp, a->z_samp ? nullptr : a->jitter, w2c_enc, K_enc, (long)cfg->n * a->rays_per_sample, a->rays_per_sample, a->K,
                           a->lindisp, cfg->H, cfg->W, cfg->feat_shift, tiles, (hipStream_t)stream, cfg->tile_blocks);
  if (rc) set_error("%s: kernel launch failed", "bts_mark_sampled_tiles");
  return rc;
}

int bts_project_features_bwd(const BtsFieldCfg* cfg, const float* feat_nchw, const float* d_proj_nhwc, const float* mlp_params,
                             int32_t N, float* d_feat_nchw, float* d_mlp_params, void* stream) {
  if (!cfg || !d_proj_nhwc || !mlp_params || N <= 0 || cfg->H <= 0 || cfg->W <= 0 || (d_mlp_params && !feat_nchw)) {
    set_error("%s: NULL pointer or non-positive size", "bts_project_features_bwd");
    return BTS_E_INVALID;
  }
  if (!bts_supported(cfg)) {
    set_error("%s: configuration outside the compiled envelope (C=%ld d_hidden=%ld n_blocks=%ld)", "bts_project_features_bwd", cfg->C,
              cfg->d_hidden, cfg->n_blocks);
    return BTS_E_UNSUPPORTED;
  }
  if (int rc = check_shift(cfg, "bts_project_features_bwd")) return rc;
  int rc = project_features_bwd_impl(cfg->C, cfg->d_hidden, feat_nchw, d_proj_nhwc, mlp_params, N, (int)feat_texels(cfg), d_feat_nchw,
                                     d_mlp_params, (hipStream_t)stream);
  if (rc) set_error("%s: kernel launch failed", "bts_project_features_bwd");
  return rc;
}

int64_t bts_proj_tile_count(const BtsFieldCfg* cfg) {
  // the same bound every entry point that takes the flags checks (check_shift: 0 .. 6, H and W multiples of 2^feat_shift); an invalid
  // configuration answers -1 with a message, never 0 -- a caller that sized its flag array with 0 would hand the kernels no flags at all
  if (!cfg || cfg->H <= 0 || cfg->W <= 0) {
    set_error("%s: NULL cfg or non-positive size", "bts_proj_tile_count");
    return -1;
  }
  if (check_shift(cfg, "bts_proj_tile_count")) return -1;
  return (feat_texels(cfg) + 63) / 64;
}

int bts_project_features_bwd_tiles(const BtsFieldCfg* cfg, const float* feat_nchw, float* d_proj_nhwc, uint8_t* tiles, const float* mlp_params,
                                   int32_t N, float* d_feat_nchw, float* d_mlp_params, int32_t clear_after, void* stream) {
  if (!cfg || !d_proj_nhwc || !tiles || !mlp_params || N <= 0 || cfg->H <= 0 || cfg->W <= 0 || (d_mlp_params && !feat_nchw)) {
    set_error("%s: NULL pointer or non-positive size", "bts_project_features_bwd_tiles");
    return BTS_E_INVALID;
  }
  if (!bts_supported(cfg)) {
    set_error("%s: configuration outside the compiled envelope (C=%ld d_hidden=%ld n_blocks=%ld)", "bts_project_features_bwd_tiles", cfg->C,
              cfg->d_hidden, cfg->n_blocks);
    return BTS_E_UNSUPPORTED;
  }
  if (int rc = check_shift(cfg, "bts_project_features_bwd_tiles")) return rc;
  int rc = project_features_bwd_tiles_impl(cfg->C, cfg->d_hidden, feat_nchw, d_proj_nhwc, tiles, mlp_params, N, (int)feat_texels(cfg), d_feat_nchw,
                                           d_mlp_params, clear_after, (hipStream_t)stream, false, cfg->tile_blocks ? cfg->W >> cfg->feat_shift : 0);
  if (rc) set_error("%s: kernel launch failed", "bts_project_features_bwd_tiles");
  return rc;
}

int bts_project_features_cl(const BtsFieldCfg* cfg, const float* feat_nhwc, const float* mlp_params, int32_t N, const uint8_t* tiles, float* proj_nhwc,
                            void* stream) {
  if (!cfg || !feat_nhwc || !mlp_params || !proj_nhwc || N <= 0 || cfg->H <= 0 || cfg->W <= 0) {
    set_error("%s: NULL pointer or non-positive size", "bts_project_features_cl");
    return BTS_E_INVALID;
  }
  if (!bts_supported(cfg)) {
    set_error("%s: configuration outside the compiled envelope (C=%ld d_hidden=%ld n_blocks=%ld)", "bts_project_features_cl", cfg->C, cfg->d_hidden,
              cfg->n_blocks);
    return BTS_E_UNSUPPORTED;
  }
  if (int rc = check_shift(cfg, "bts_project_features_cl")) return rc;
  int rc = project_features_impl(cfg->C, cfg->d_hidden, feat_nhwc, mlp_params, N, (int)feat_texels(cfg), proj_nhwc, tiles, (hipStream_t)stream, true,
                                 cfg->tile_blocks ? cfg->W >> cfg->feat_shift : 0);
  if (rc) set_error("%s: kernel launch failed", "bts_project_features_cl");
  return rc;
}

int bts_project_features_bwd_cl(const BtsFieldCfg* cfg, const float* feat_nhwc, float* d_proj_nhwc, uint8_t* tiles, const float* mlp_params, int32_t N,
                                float* d_feat_nhwc, float* d_mlp_params, int32_t clear_after, void* stream) {
  if (!cfg || !d_proj_nhwc || !mlp_params || N <= 0 || cfg->H <= 0 || cfg->W <= 0 || (d_mlp_params && !feat_nhwc)) {
    set_error("%s: NULL pointer or non-positive size", "bts_project_features_bwd_cl");
    return BTS_E_INVALID;
  }
  if (!bts_supported(cfg)) {
    set_error("%s: configuration outside the compiled envelope (C=%ld d_hidden=%ld n_blocks=%ld)", "bts_project_features_bwd_cl", cfg->C, cfg->d_hidden,
              cfg->n_blocks);
    return BTS_E_UNSUPPORTED;
  }
  if (int rc = check_shift(cfg, "bts_project_features_bwd_cl")) return rc;
  int rc = project_features_bwd_tiles_impl(cfg->C, cfg->d_hidden, feat_nhwc, d_proj_nhwc, tiles, mlp_params, N, (int)feat_texels(cfg), d_feat_nhwc,
                                           d_mlp_params, clear_after, (hipStream_t)stream, true, cfg->tile_blocks ? cfg->W >> cfg->feat_shift : 0);
  if (rc) set_error("%s: kernel launch failed", "bts_project_features_bwd_cl");
  return rc;
}

int bts_field_query(const BtsFieldCfg* cfg, const BtsFieldTensors* t, const float* xyz, int32_t P, int32_t only_density,
                    float* rgb, float* invalid, float* sigma, void* stream) {
  int rc = check_cfg(cfg, t, !only_density);
  if (rc) return rc;
  if (!xyz || !sigma || P <= 0 || (!only_density && cfg->nv > 0 && !rgb)) {
    set_error("%s: NULL/empty query argument (P=%ld)", "bts_field_query", P);
    return BTS_E_INVALID;
  }
  return field_query_impl(cfg, t, xyz, P, only_density, rgb, invalid, sigma, (hipStream_t)stream);
}

int bts_occupancy_profile(const BtsFieldCfg* cfg, const BtsFieldTensors* t, const float* xyz, int32_t Y, int32_t columns, float threshold,
                          int32_t only_density, float* profile, float* sigma, void* stream) {
  int rc = check_cfg(cfg, t, !only_density);
  if (rc) return rc;
  if (!xyz || !profile || Y <= 0 || columns <= 0) {
    set_error("%s: NULL/empty argument (Y=%ld, columns=%ld)", "bts_occupancy_profile", Y, columns);
    return BTS_E_INVALID;
  }
  if (Y > 64 || !t->proj_nhwc || (long)Y * columns > 0x7FFFFFFFL) {
    set_error("%s: needs Y <= 64 levels (got %ld), the projected feature map and fewer than 2^31 points", "bts_occupancy_profile", Y);
    return BTS_E_UNSUPPORTED;
  }
  return occupancy_profile_impl(cfg, t, xyz, Y, columns, threshold, only_density, profile, sigma, (hipStream_t)stream);
}

#define BTS_CHECK_LAYOUT(cond, name)                      \
  if (!(cond)) {                                          \
    set_error("%s: NULL pointer or non-positive size", name); \
    return BTS_E_INVALID;                                 \
  }
#define BTS_RET_LAUNCH(expr, name)                              \
  {                                                             \
    int rc_ = (expr);                                           \
    if (rc_) set_error("%s: kernel launch failed", name);       \
    return rc_;                                                 \
  }

int bts_nchw_to_nhwc(const float* src, float* dst, int32_t N, int32_t C, int32_t H, int32_t W, void* stream) {
  BTS_CHECK_LAYOUT(src && dst && N > 0 && C > 0 && H > 0 && W > 0, "bts_nchw_to_nhwc");
  BTS_RET_LAUNCH(transpose_launch(src, dst, N, C, H, W, true, (hipStream_t)stream), "bts_nchw_to_nhwc");
}
int bts_nhwc_to_nchw(const float* src, float* dst, int32_t N, int32_t C, int32_t H, int32_t W, void* stream) {
  BTS_CHECK_LAYOUT(src && dst && N > 0 && C > 0 && H > 0 && W > 0, "bts_nhwc_to_nchw");
  BTS_RET_LAUNCH(transpose_launch(src, dst, N, C, H, W, false, (hipStream_t)stream), "bts_nhwc_to_nchw");
}
int bts_pack_rgb(const float* src, float* dst, int32_t N, int32_t H, int32_t W, float scale, float shift, void* stream) {
  BTS_CHECK_LAYOUT(src && dst && N > 0 && H > 0 && W > 0, "bts_pack_rgb");
  BTS_RET_LAUNCH(pack_rgb_launch(src, dst, N, H, W, scale, shift, (hipStream_t)stream), "bts_pack_rgb");
}
int bts_gen_rays(const float* poses, const float* projs, int32_t V, int32_t H, int32_t W, float z_near, float z_far,
                 int32_t norm_dir, float* rays, void* stream) {
  BTS_CHECK_LAYOUT(poses && projs && rays && V > 0 && H > 0 && W > 0, "bts_gen_rays");
  BTS_RET_LAUNCH(gen_rays_launch(poses, projs, V, H, W, z_near, z_far, norm_dir, rays, (hipStream_t)stream), "bts_gen_rays");
}
int bts_patch_rays(const float* poses, const float* projs, const float* images, const int32_t* patch_v, const int32_t* patch_y,
                   const int32_t* patch_x, int32_t n, int32_t v, int32_t c, int32_t H, int32_t W, int32_t P, int32_t ph, int32_t pw,
                   float z_near, float z_far, int32_t norm_dir, float* rays, float* rgb_gt, void* stream) {
  BTS_CHECK_LAYOUT(poses && projs && patch_v && patch_y && patch_x && rays && n > 0 && v > 0 && H > 0 && W > 0 && P >= 0 && ph > 0 && pw > 0 &&
                       ph <= H && pw <= W && (!images || (rgb_gt && c > 0)),
                   "bts_patch_rays");
  BTS_RET_LAUNCH(patch_rays_launch(poses, projs, images, patch_v, patch_y, patch_x, n, v, c, H, W, P, ph, pw, z_near, z_far, norm_dir, rays,
                                   rgb_gt, (hipStream_t)stream),
                 "bts_patch_rays");
}
int bts_photometric_loss(const BtsLossArgs* a, void* stream) {
  BTS_CHECK_LAYOUT(a && a->rgb && a->rgb_gt && a->parts && a->n_patches >= 0 && a->patch_h > 0 && a->patch_w > 0 &&
                       a->patch_h * a->patch_w <= 64 && a->nv > 0 && a->invalid_policy >= 0 && a->invalid_policy <= 2 &&
                       (a->invalid_policy != 1 || a->invalid_any || (a->invalid && a->K > 0)) &&
                       (a->invalid_policy != 2 || a->invalid_wsum || (a->invalid && a->weights && a->K > 0)) &&
                       (!a->edge_aware_smoothness || a->depth),
                   "bts_photometric_loss");
  BTS_RET_LAUNCH(photometric_loss_impl(a, (hipStream_t)stream), "bts_photometric_loss");
}
int bts_sample_coarse(const float* rays, const float* u, int64_t B, int32_t K, int32_t lindisp, float* z_samp, void* stream) {
  BTS_CHECK_LAYOUT(rays && u && z_samp && B > 0 && K > 0, "bts_sample_coarse");
  BTS_RET_LAUNCH(sample_coarse_launch(rays, u, (long)B, K, lindisp, z_samp, (hipStream_t)stream), "bts_sample_coarse");
}
int bts_distance_to_z(const float* depths, const float* inv_K, int32_t N, int32_t H, int32_t W, float* out, void* stream) {
  BTS_CHECK_LAYOUT(depths && inv_K && out && N > 0 && H > 0 && W > 0, "bts_distance_to_z");
  BTS_RET_LAUNCH(distance_to_z_launch(depths, inv_K, N, H, W, out, (hipStream_t)stream), "bts_distance_to_z");
}

int bts_invert_small(const float* src, float* dst, int32_t N, int32_t dim, void* stream) {
  BTS_CHECK_LAYOUT(src && dst && N > 0 && (dim == 3 || dim == 4), "bts_invert_small");
  BTS_RET_LAUNCH(invert_small_launch(src, dst, N, dim, (hipStream_t)stream), "bts_invert_small");
}

}  // extern "C"
