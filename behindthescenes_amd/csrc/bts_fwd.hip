// Host side of the fused forward + the PROJ = false instantiations (raw channels-last features).
#include "bts_field_kernel.h"

#include <cstdlib>

namespace bts {

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void set_error(const char* fmt, const char* a, long b, long c, long d) {
  snprintf(g_err, sizeof(g_err), fmt, a, b, c, d);
}
const char* last_error() { return g_err; }

bool shape_supported(int C, int HD, int NB) {
  return (C == 64 && HD == 64 && NB == 0) || (C == 32 && HD == 32 && NB == 1) || (C == 32 && HD == 32 && NB == 0);
}

// raw channels-last features (feat_nhwc, for callers that cannot pre-project): the compact lane = sample render kernel and the
// lane = point query kernel; everything else of bts_field_kernel.h is instantiated in the probe build only
template int launch_field<true, false>(const FwdParams&, int, int, int, int, hipStream_t);
template int launch_render<false>(const FwdParams&, int, int, int, int, hipStream_t);
#ifdef BTS_PROBE
template int launch_field<false, false>(const FwdParams&, int, int, int, int, hipStream_t);
extern template int launch_field<false, true>(const FwdParams&, int, int, int, int, hipStream_t);
extern template int launch_render<true>(const FwdParams&, int, int, int, int, hipStream_t);
#endif
int launch_render_pipelined(const FwdParams& p, int C, int HD, int NB, int grid, hipStream_t s);

FwdParams make_params(const BtsFieldCfg* cfg, const BtsFieldTensors* t) {
  FwdParams p;
  memset(&p, 0, sizeof(p));
  p.feat = t->feat_nhwc, p.proj = t->proj_nhwc, p.K_enc = t->K_enc, p.w2c_enc = t->w2c_enc;
  p.imgs = t->imgs_nhwc4, p.K_r = t->K_r, p.w2c_r = t->w2c_r;
  p.empty_feature = t->empty_feature, p.mlp = t->mlp_params;
  p.n = cfg->n, p.H = cfg->H, p.W = cfg->W, p.nv = cfg->nv, p.fs = cfg->feat_shift;
  p.enc_view = cfg->feat_shift ? -1 : cfg->enc_render_view;
  p.code_mode = cfg->code_mode, p.inv_z = cfg->inv_z, p.learn_empty = cfg->learn_empty, p.empty_empty = cfg->empty_empty;
  p.freq_factor = cfg->freq_factor, p.d_min = cfg->d_min, p.d_max = cfg->d_max;
  // python-double constants rounded once to fp32, as `1 / self.d_max` etc. enter the reference's tensor ops (models_bts.py:160-169)
  p.inv_dmax = (float)(1.0 / (double)cfg->d_max);
  p.inv_range = (float)(1.0 / (double)cfg->d_min - 1.0 / (double)cfg->d_max);
  p.range = (float)((double)cfg->d_max - (double)cfg->d_min);
  return p;
}

// lanes per ray / ray groups of the lane = sample kernels: short rays (K <= 32) share a wave iteration when the per-sample ray
// count allows whole groups
void render_geometry(FwdParams& p, int n) {
  int lpr = 64;
  if (p.K <= 32) {
    lpr = p.K <= 8 ? 8 : (p.K <= 16 ? 16 : 32);
    if (p.Bp % (64 / lpr) != 0) lpr = 64;
  } else if (p.K <= 48 && p.Bp % 4 == 0) {
    lpr = 48;   // four rays in three wave iterations (render_kernel_p's 48-lane mode): exp_re10k.yaml's 48 samples per ray
  }
  p.lpr = lpr;
  p.groups = lpr == 48 ? (long)n * p.Bp / 4 : (long)n * p.Bp / (64 / lpr);
}

// persistent grid: 2 work-groups (8 waves) per CU of the current device, multiple of 8 so that every XCD owns an equal contiguous
// share of the rays
int device_cu_count() {
  static thread_local int cus[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
  if (!cus[dev]) {
    hipDeviceProp_t prop;
    cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return cus[dev];
}

int render_grid(const FwdParams& p) {
  const long cap = (long)BTS_FWD_WAVES * device_cu_count();
  const long want = (p.groups + 3) / 4;
  long g = want < cap ? want : cap;
  g = (g + 7) / 8 * 8;
  return (int)g;
}

// chunk of the XCD interleave = twice the waves an XCD runs at once (one round of the resident waves covers half a chunk), at least
// 64 groups
// -- but small enough that every XCD gets at least four chunks: chunk c belongs to XCD c % 8, and with, say, 12 chunks four XCDs would
// run two and four one (the launch then takes two chunks' time for one and a half chunks of work per XCD: the 48-lane mode's 6 144
// groups at the RE10K shape measured 30 % slower than the 24 576 one-ray groups before this)
int render_chunk_log2(int grid, long groups) {
  int waves_per_xcd = grid / 8 * 4, l = 6;
  while ((1 << l) < 2 * waves_per_xcd) ++l;
  while (l > 6 && (groups >> l) < 32) --l;
  return l;
}

int render_fwd_impl(const BtsFieldCfg* cfg, const BtsFieldTensors* t, const BtsRenderArgs* a, hipStream_t s) {
  FwdParams p = make_params(cfg, t);
  p.rays = a->rays, p.z_samp = a->z_samp;
  p.jitter = a->z_samp ? nullptr : a->jitter, p.z_out = a->z_samp ? nullptr : a->z_samp_out, p.lindisp = a->lindisp;
  p.Bp = a->rays_per_sample, p.K = a->K, p.hard_cap = a->hard_alpha_cap, p.white_bkgd = a->white_bkgd;
  p.rgb = a->rgb, p.depth = a->depth, p.weights = a->weights, p.alphas = a->alphas, p.invalid = a->invalid;
  p.rgb_samps = a->rgb_samps, p.sigma_raw = a->sigma_raw, p.trans = a->trans;
  p.invalid_wsum = a->invalid_wsum, p.invalid_any = a->invalid_any;
  p.sigma_noise = a->sigma_noise;
  p.tiles_per_sample = (a->rays_per_sample + 255) / 256;
#ifdef BTS_PROBE   // A/B switches exist only in the probe build (python -m behindthescenes_amd.build --probe); the product has one path
  if (getenv("BTS_LANE_IS_RAY") && !p.fs) {  // round-1a mapping (one lane = one ray); the legacy kernels know full-size maps only
    if (p.proj) return launch_field<false, true>(p, cfg->C, cfg->d_hidden, cfg->n_blocks, p.tiles_per_sample * cfg->n, s);
    return launch_field<false, false>(p, cfg->C, cfg->d_hidden, cfg->n_blocks, p.tiles_per_sample * cfg->n, s);
  }
#endif
#ifdef BTS_PROBE
  if (const char* e = getenv("BTS_ABLATE")) p.ablate |= atoi(e);
  if (const char* e = getenv("BTS_DBG_PTR")) p.dbg = (unsigned long long*)strtoull(e, nullptr, 0);
#endif
  render_geometry(p, cfg->n);
  if (!p.proj && p.lpr == 48) p.lpr = 64, p.groups = (long)cfg->n * p.Bp;   // (the raw-feature route's compact kernel knows power-of-two groups only)
  if (p.groups > 0x7FF00000L) {   // the kernels index ray groups with 32 bits
    set_error("%s: too many rays in one call (%ld groups)", "bts_render_fwd", p.groups);
    return BTS_E_UNSUPPORTED;
  }
  const int grid = render_grid(p);
  p.chunk_log2 = render_chunk_log2(grid, p.groups);
#ifdef BTS_PROBE
  if (p.proj && !p.fs && getenv("BTS_RENDER_V1")) return launch_render<true>(p, cfg->C, cfg->d_hidden, cfg->n_blocks, grid, s);  // compact lane = sample kernel
#endif
  if (p.proj) return launch_render_pipelined(p, cfg->C, cfg->d_hidden, cfg->n_blocks, grid, s);
  if (p.invalid_wsum || p.invalid_any) {
    set_error("%s: invalid_wsum / invalid_any need the projected feature map (proj_nhwc)", "bts_render_fwd");
    return BTS_E_UNSUPPORTED;
  }
  if (!p.z_samp) {
    set_error("%s: sampling inside the kernel (z_samp NULL, jitter given) needs the projected feature map (proj_nhwc)", "bts_render_fwd");
    return BTS_E_UNSUPPORTED;
  }
  return launch_render<false>(p, cfg->C, cfg->d_hidden, cfg->n_blocks, grid, s);
}

int query_impl(const BtsFieldCfg* cfg, const BtsFieldTensors* t, const float* xyz, int P, int only_density, float* rgb, float* invalid,
               float* sigma, int cols, int col_len, float threshold, float* profile, hipStream_t s);   // bts_query.hip

int field_query_impl(const BtsFieldCfg* cfg, const BtsFieldTensors* t, const float* xyz, int P, int only_density, float* rgb,
                     float* invalid, float* sigma, hipStream_t s) {
  // projected feature map (the default hand-over): the pipelined lane = point kernel of bts_query.hip; raw channels-last features
  // (callers that cannot pre-project): the compact kernel below
  if (t->proj_nhwc) return query_impl(cfg, t, xyz, P, only_density, only_density ? nullptr : rgb, invalid, sigma, 0, 0, 0.0f, nullptr, s);
  FwdParams p = make_params(cfg, t);
  p.xyz = xyz, p.Bp = P, p.K = 1, p.only_density = only_density;
  if (only_density) p.nv = 0;
  p.rgb = rgb, p.invalid = invalid, p.q_sigma = sigma;
  p.tiles_per_sample = (P + 255) / 256;
  return launch_field<true, false>(p, cfg->C, cfg->d_hidden, cfg->n_blocks, p.tiles_per_sample * cfg->n, s);
}

int occupancy_profile_impl(const BtsFieldCfg* cfg, const BtsFieldTensors* t, const float* xyz, int Y, int cols, float threshold,
                           int only_density, float* profile, float* sigma, hipStream_t s) {
  return query_impl(cfg, t, xyz, Y * cols, only_density, nullptr, nullptr, sigma, cols, Y, threshold, profile, s);
}

}  // namespace bts
