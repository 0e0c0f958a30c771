/*
 * bts_render.h -- C ABI of the MI355X-native BehindTheScenes density-field renderer (libbts_render.so).
 *
 * The reference (Brummi/BehindTheScenes @ 2024_10_08) is pure Python/PyTorch and has no FFI of its own; its seam for
 * this path is a Python object protocol.  The entry points below are what a binding for that seam
 * needs and nothing more; each one names the reference function(s) it replaces.  Plain pointers and sizes only: no torch
 * types, no allocation inside, no exceptions -- every function returns 0 on success or a negative BTS_E_* code and
 * leaves a message retrievable with bts_last_error().  All pointers are DEVICE pointers to fp32 data unless noted; all
 * work is enqueued on the caller's HIP stream (`stream` is a hipStream_t passed as void*; NULL = default stream) and the
 * functions are re-entrant (autograd may call the backward from another thread).
 *
 * Data layout in HBM:
 *   proj_nhwc   (n, H, W, Hd)       "projected" feature map G = F . w_in[:, :C]^T, channels-last: bilinear interpolation and
 *                                   lin_in are both linear, so lin_in(bilinear(F)) == bilinear(G) + w_in[:, C:] . PE + b_in;
 *                                   bts_project_features builds G from the encoder's NCHW output in one pass (it replaces the
 *                                   NCHW->NHWC hand-over), the render kernels gather G straight into the MFMA accumulators.
 *   feat_nhwc   (n, H, W, C)        alternative: raw channels-last F, lin_in evaluated per point (forward / query only)
 *   imgs_nhwc4  (n, nv, H, W, 4)    rgb0-packed colour frames in [0,1]                 -> one 16-byte load per tap
 *   rays        (n*Bp, 8)           [origin(3), direction(3), near, far]   (reference layout, util.py:270-273)
 *   z_samp      (n*Bp, K)           sample depths per ray                  (reference layout, nerf.py:210-218)
 *   mlp_params  packed fp32: w_in (Hd x d_in, row-major = nn.Linear.weight), b_in (Hd),
 *               n_blocks x [fc_0.weight (Hd x Hd), fc_0.bias (Hd), fc_1.weight (Hd x Hd), fc_1.bias (Hd)],
 *               w_out (Hd), b_out (1);  d_in = C + 3 + 6*num_freqs.
 */
#ifndef BTS_RENDER_H
#define BTS_RENDER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BTS_ABI_VERSION 9

enum {
  BTS_OK = 0,
  BTS_E_INVALID = -1,     /* NULL pointer / non-positive size / inconsistent arguments */
  BTS_E_UNSUPPORTED = -2, /* configuration outside the compiled envelope (see bts_supported) */
  BTS_E_LAUNCH = -3,      /* HIP runtime reported an error at launch */
  BTS_E_WORKSPACE = -4    /* workspace too small (see bts_render_bwd_workspace) */
};

/* Scalar configuration of the field: BTSNet.__init__ (models/bts/model/models_bts.py:18-54),
 * PositionalEncoding (models/common/model/code.py:11-28), ResnetFC shape (models/common/model/resnetfc.py:65-130).
 *
 * Precision of lin_in's inputs (resnetfc.py:147, models_bts.py:150-171).  The 36 trigonometric inputs AND the three raw ones -- the
 * projected x, y and the depth code -- enter the f16 matrix pipe as two round-to-nearest halves each: v = hi + lo + r with
 * |r| <= 2^-22 |v| in the worst case (each half rounds to 11 significand bits; 2^-23 |v| typical), the weights likewise after an exact
 * power-of-two scaling; the products hi.hi + lo.hi + hi.lo are exact in the fp32 accumulator, the dropped lo.lo term is <= 2^-22 |w v|.
 * That is up to four fp32 rounding units (2^-24) per product.  For the sines (|v| <= 1) it vanishes in the accumulation's own rounding.
 * x and y are NOT bounded by 1: a point beside or behind the encoder camera projects to |x|, |y| of hundreds (the perspective divide
 * clamps z at 1e-3), and the split then costs up to 2^-22 |x| ABSOLUTE per product.  It stays invisible next to what the reference's own
 * fp32 arithmetic does there: the twelve encoding inputs sin(f x [+ pi/2]), f = 1.5 .. 48, are evaluated on fp32 arguments whose rounding
 * alone is 2^-24 f |x| -- 12 to 48 times the split's error on the raw row, through weights of the same size.  Beyond |x|, |y| = 2083
 * (encoding arguments past the fast sines' range) a wave takes the exact fp32 routine.  Pinned by
 * tests/test_gpu_parity.py::test_raw_rows_worst_case_vs_fp64: |x|, |y| log-uniform up to 2000, the code at both ends of [-1, 1] and
 * clamped, learn_empty on and off -- never further from an fp64 evaluation than 1.5 x the fp32 reference's own distance. */
typedef struct BtsFieldCfg {
  int32_t n;           /* batch ("super-batch") size */
  int32_t H, W;        /* feature-map and colour-frame size */
  int32_t C;           /* feature channels (encoder.latent_size) */
  int32_t d_hidden;    /* MLP hidden width */
  int32_t n_blocks;    /* number of ResnetBlockFC */
  int32_t nv;          /* number of render (colour) views, 0..BTS_MAX_VIEWS */
  int32_t num_freqs;   /* PE octaves (6 in every shipped config) */
  int32_t code_mode;   /* 0 = "z", 1 = "distance"  (models_bts.py:157-171) */
  int32_t inv_z;       /* 1 = inverse-depth normalisation */
  int32_t learn_empty; /* 1 = replace features of out-of-frustum points by empty_feature (models_bts.py:176-182) */
  int32_t empty_empty; /* 1 = sigma := 0 for out-of-frustum points (models_bts.py:323-324) */
  float freq_factor;   /* PE base frequency (1.5) */
  float d_min, d_max;  /* z_near, z_far of the field */
  /* ABI 4: the feature map of decoder scale s handed over at ITS OWN size.  BTSNet.encode resizes every scale's map to scale 0's
   * size with F.interpolate(mode="nearest") (models_bts.py:115-117) before the renderer samples it; for sizes that differ by 2^s that
   * map is texel (y >> s, x >> s) of the small one, so the kernels index the small map directly: feat / proj / d_proj are
   * (n, H >> feat_shift, W >> feat_shift, .) and every result is bit-identical to the resized map's.  H and W (the colour frames'
   * size, and the size the bilinear taps are computed for) must be multiples of 2^feat_shift.  0 = a full-size map. */
  int32_t feat_shift;
  /* ABI 5: index j of a render view whose camera IS the encoder camera of every batch element -- K_r[:, j] == K_enc and
   * w2c_r[:, j] == w2c_enc bit for bit, i.e. ids_render[j] == ids_encoder[0] in BTSNet.encode (models_bts.py:84-97; the eval_depth
   * configuration renders its colours from the encoder frame) -- or -1.  The forward and query kernels then take that view's
   * projection, frustum flag and bilinear weights from the encoder view's instead of evaluating them a second time: same inputs, same
   * instruction sequence, bit-identical results (tests/test_gpu_abi5.py).  A hint: -1 is always correct. */
  int32_t enc_render_view;
  /* ABI 9: the geometry of this map's tile flags (BtsRenderGrads.d_proj_tiles, every `tiles` argument).  0: tile t = texels 64 t .. 64 t + 63
   * of the row-major map (ABI 6 - 8).  1: tile t = the block of 4 rows x 16 texels at rows 4 (t / (W' / 16)) .., columns 16 (t % (W' / 16)) ..
   * of the (H', W') = (H >> feat_shift, W >> feat_shift) map -- honoured where H' is a multiple of 4 and W' of 16, otherwise the
   * linear form is used as with 0 (bts_proj_tile_count is ceil(H' W' / 64) either way).  A training step's samples lie on streaks along
   * the epipolar lines: they touch 38 % of the 64 x 1 runs and 26 % of the 16 x 4 blocks of exp_kitti_360.yaml's maps.  Blocks pay with a
   * CHANNELS-LAST feature map (a block = four 4 KB pieces of F, dF, G, dG each: train step -3 %); with an NCHW map its rows come
   * apart into 64-byte pieces and the projection backward is 2 % SLOWER than with runs (profiles/r06f) -- so: 1 for channels-last maps, 0
   * for NCHW ones.  Every call that produces or consumes one flag array must see the same value. */
  int32_t tile_blocks;
} BtsFieldCfg;

#define BTS_MAX_VIEWS 8

/* State left behind by BTSNet.encode (models_bts.py:128-136), in the layouts above. */
typedef struct BtsFieldTensors {
  const float* feat_nhwc;     /* (n, H, W, C)   raw features; may be NULL when proj_nhwc is given */
  const float* proj_nhwc;     /* (n, H, W, Hd)  projected features (preferred; required by the backward) or NULL */
  const float* K_enc;         /* (n, 3, 3)   normalised intrinsics of the encoder view */
  const float* w2c_enc;       /* (n, 4, 4)   world -> encoder camera */
  const float* imgs_nhwc4;    /* (n, nv, H, W, 4), may be NULL iff nv == 0 */
  const float* K_r;           /* (n, nv, 3, 3) */
  const float* w2c_r;         /* (n, nv, 4, 4) */
  const float* empty_feature; /* (C) or NULL */
  const float* mlp_params;    /* packed, see above */
} BtsFieldTensors;

/* One call of NeRFRenderer.composite (models/common/render/nerf.py:210-313) on n*Bp rays. */
typedef struct BtsRenderArgs {
  int32_t rays_per_sample; /* Bp */
  int32_t K;               /* samples per ray */
  int32_t hard_alpha_cap;  /* nerf.py:285-286 */
  int32_t white_bkgd;      /* nerf.py:301-304: rgb += 1 - sum(weights); honoured by the forward AND the backward */
  const float* rays;       /* (n*Bp, 8) */
  const float* z_samp;     /* (n*Bp, K), or NULL with `jitter` below */
  /* outputs; the per-sample ones may be NULL when not wanted */
  float* rgb;              /* (n*Bp, nv*3) */
  float* depth;            /* (n*Bp) */
  float* weights;          /* (n*Bp, K)        or NULL */
  float* alphas;           /* (n*Bp, K)        or NULL */
  float* invalid;          /* (n*Bp, K, nv)    or NULL   1.0 = sample outside a frustum */
  float* rgb_samps;        /* (n*Bp, K, nv*3)  or NULL */
  float* sigma_raw;        /* (n*Bp, K)        or NULL   pre-softplus MLP output  } the two per-sample activations the  */
  float* trans;            /* (n*Bp, K)        or NULL   transmittance before sample } backward needs (8 B per sample)     */
  /* ABI 2: what the photometric loss' invalid-ray policies (loss.py:100-118) need of `weights` and `invalid`, reduced over the samples
   * in the kernel's epilogue -- a training step then stores 8 B per ray and view instead of (1 + nv) * 4 * K B per ray */
  float* invalid_wsum;     /* (n*Bp, nv)       or NULL   sum_k weights_k * invalid_k,v   (policy weight_guided: > 0.9) */
  float* invalid_any;      /* (n*Bp, nv)       or NULL   max_k invalid_k,v               (policy strict) */
  /* ABI 3: the density noise of a training step (nerf.py:279-280: sigmas + randn_like(sigmas) * noise_std, drawn by the caller),
   * added to softplus(s) before relu / alpha -- INPUT of the forward and of the backward (which needs the sign of the sum) */
  const float* sigma_noise;/* (n*Bp, K)        or NULL */
  /* ABI 5: NeRFRenderer.sample_coarse (nerf.py:103-123) inside the forward kernel.  With z_samp == NULL the sample depths are computed
   * from every ray's [near, far] and the stratified jitter `jitter` in [0, 1) (the caller's torch.rand_like draw, nerf.py:112) by the
   * routine bts_sample_coarse runs -- bit-identical depths, one launch and 8 bytes per sample of HBM traffic less; `lindisp` selects
   * disparity- (nerf.py:117) or depth-linear (:115) spacing; z_samp_out, when given, receives the depths (a training step hands them
   * to bts_render_bwd as z_samp).  Ignored when z_samp is given.  Needs proj_nhwc. */
  const float* jitter;     /* (n*Bp, K)        or NULL */
  float* z_samp_out;       /* (n*Bp, K)        or NULL */
  int32_t lindisp;
  int32_t reserved_;       /* 0 */
} BtsRenderArgs;

/* Gradients flowing into / out of the renderer (what torch.autograd would compute through nerf.py:283-299,
 * models_bts.py:266-338 and resnetfc.py:132-184).  No gradient is produced for rays, z_samp, poses or colours. */
typedef struct BtsRenderGrads {
  const float* g_rgb;      /* (n*Bp, nv*3) or NULL */
  const float* g_depth;    /* (n*Bp)       or NULL */
  const float* g_weights;  /* (n*Bp, K)    or NULL */
  const float* g_alphas;   /* (n*Bp, K)    or NULL */
  float* d_proj_nhwc;      /* (n, H, W, Hd) gradient w.r.t. proj_nhwc, ACCUMULATED into (caller zero-fills), or NULL to skip */
  float* d_mlp_params;     /* packed like mlp_params, ACCUMULATED into (caller zero-fills), or NULL to skip */
  float* d_empty_proj;     /* (Hd) gradient w.r.t. the PROJECTED empty feature (w_in[:, :C] . empty_feature), accumulated, or NULL */
  /* ABI 6: which parts of d_proj_nhwc received anything.  (n, bts_proj_tile_count(cfg)) bytes, one per tile of 64 texels of an image's
   * map: the backward SETS the byte of every tile it adds into (it never clears one), or NULL.  A training step's rays touch 8-15 % of
   * the texels; bts_project_features_bwd_tiles reads only the flagged tiles.
   * TILE GEOMETRY (ABI 9; the same for every `tiles` argument of this header): BtsFieldCfg.tile_blocks -- runs of 64 consecutive texels
   * (0; ABI 6 - 8: always) or blocks of 4 rows x 16 texels (1).  Flags are produced and consumed by this library (bts_render_bwd,
   * bts_mark_sampled_tiles -> bts_project_features_tiles / _bwd_tiles / _cl); a caller that writes its own needs the mapping
   * (behindthescenes_amd.native.proj_tile_map). */
  uint8_t* d_proj_tiles;
} BtsRenderGrads;

int bts_abi_version(void);
const char* bts_last_error(void);

/* 1 if (C, d_hidden, n_blocks, nv, num_freqs) is inside the compiled envelope, else 0. */
int bts_supported(const BtsFieldCfg* cfg);
/* number of floats in mlp_params for cfg */
int64_t bts_mlp_param_count(const BtsFieldCfg* cfg);

/* Fused sample -> project -> bilinear(F) -> PE -> MLP -> softplus -> colour taps -> alpha-composite.
 * Replaces NeRFRenderer.composite + BTSNet.forward + sample_features + sample_colors + ResnetFC.forward +
 * PositionalEncoding.forward (nerf.py:210-313, models_bts.py:138-338, resnetfc.py:132-184, code.py:30-42). */
int bts_render_fwd(const BtsFieldCfg* cfg, const BtsFieldTensors* t, const BtsRenderArgs* a, void* stream);

/* Backward of bts_render_fwd.  `a` must carry sigma_raw and trans written by the forward (weights/alphas are not needed).  When
 * a->rgb_samps is non-NULL it is READ here: the forward's per-sample colours, which spare the backward one projection + four taps
 * per view and sample (training requests rgb_samps anyway); NULL = recompute them.
 * workspace: bts_render_bwd_workspace(cfg, a) bytes of device scratch -- what the backward's passes hand each other: for the plain
 * MLP (n_blocks = 0) with K <= 64, 20 bytes per sample at d_hidden = 64 (the gradient at the pre-softplus density + the relu gates
 * as bits, per sample and per channel); with ResnetBlockFC layers or K > 64, 4 (d_hidden + 1) bytes per sample (the gradient row at
 * lin_in's output + the gradient at the pre-softplus density).  Contents need no initialisation. */
size_t bts_render_bwd_workspace(const BtsFieldCfg* cfg, const BtsRenderArgs* a);
int bts_render_bwd(const BtsFieldCfg* cfg, const BtsFieldTensors* t, const BtsRenderArgs* a, const BtsRenderGrads* g,
                   void* workspace, size_t workspace_bytes, void* stream);

/* Hand-over from the (PyTorch) encoder, fused with the feature part of lin_in (resnetfc.py:147 restricted to the first C
 * inputs): feat_nchw (N, C, H, W) -> proj_nhwc (N, H, W, Hd) = F . w_in[:, :C]^T.  Uses cfg->C, d_hidden, n_blocks, num_freqs. */
int bts_project_features(const BtsFieldCfg* cfg, const float* feat_nchw, const float* mlp_params, int32_t N, float* proj_nhwc,
                         void* stream);
/* Its backward: d_feat_nchw (N, C, H, W) = d_proj . w_in[:, :C] (written, not accumulated; may be NULL) and
 * d_mlp_params[w_in[:, :C]] += d_proj^T . F (accumulated; may be NULL).  empty_feature handling: d_empty_proj (Hd) is the
 * gradient the render backward accumulated for the projected empty feature; d_empty_feature (C) += w_in[:, :C]^T d_empty_proj. */
int bts_project_features_bwd(const BtsFieldCfg* cfg, const float* feat_nchw, const float* d_proj_nhwc, const float* mlp_params,
                             int32_t N, float* d_feat_nchw, float* d_mlp_params, void* stream);

/* ABI 6: the same backward over a SPARSE d_proj.  `tiles` (N, bts_proj_tile_count(cfg)) is the flag array bts_render_bwd filled
 * (BtsRenderGrads.d_proj_tiles): only flagged tiles of d_proj_nhwc are read, every other texel is taken as zero (it must BE zero if
 * clear_after is set, see below).  d_feat_nchw is written densely (zeros where nothing arrived), d_mlp_params accumulated.  With
 * clear_after != 0 the call also writes zeros over the flagged tiles of d_proj_nhwc and resets their flags: a caller that keeps one
 * (d_proj, tiles) pair per map shape zero-fills it once and never again (the fill of a training step's d_proj is as large as the
 * gradient's whole HBM traffic otherwise: 503 MB at exp_kitti_360.yaml's batch). */
/* tiles per image of cfg's map ((H >> feat_shift) * (W >> feat_shift) texels, 64 per tile, rounded up); -1 for an invalid cfg (feat_shift
 * outside 0 .. 6, H or W not a multiple of 2^feat_shift, non-positive sizes) */
int64_t bts_proj_tile_count(const BtsFieldCfg* cfg);
/* ABI 6: the forward side of the same observation -- a render reads only the tiles its samples' taps land in.
 * bts_mark_sampled_tiles: for every sample of every ray of `a` (rays, rays_per_sample, K, and z_samp or jitter + lindisp exactly as
 * bts_render_fwd takes them; cfg->n, H, W, feat_shift; the encoder cameras) the flags (n, bts_proj_tile_count(cfg)) of the four tap
 * texels' tiles are SET (caller zero-fills) -- computed with the render kernels' own depth / projection / tap routines, i.e. the same
 * texels bit for bit.  bts_project_features_tiles then evaluates the flagged tiles of proj_nhwc only and leaves the rest of the buffer
 * untouched (uninitialised memory stays uninitialised): such a map is good for bts_render_fwd / bts_render_bwd on THAT sample set and
 * for nothing else (no field queries). */
int bts_mark_sampled_tiles(const BtsFieldCfg* cfg, const float* K_enc, const float* w2c_enc, const BtsRenderArgs* a, uint8_t* tiles,
                           void* stream);
int bts_project_features_tiles(const BtsFieldCfg* cfg, const float* feat_nchw, const float* mlp_params, int32_t N, const uint8_t* tiles,
                               float* proj_nhwc, void* stream);
int bts_project_features_bwd_tiles(const BtsFieldCfg* cfg, const float* feat_nchw, float* d_proj_nhwc, uint8_t* tiles,
                                   const float* mlp_params, int32_t N, float* d_feat_nchw, float* d_mlp_params, int32_t clear_after,
                                   void* stream);
/* ABI 8: the encoder's map handed over CHANNELS-LAST -- (N, H >> s, W >> s, C) in memory, i.e. a torch tensor of shape (N, C, h, w) in
 * channels_last format: what MIOpen's NHWC convolutions and bts_conv3x3_fwd (out_nchw = 0) write.  The same products as the NCHW entry
 * points; the forward sums them in another order (G agrees to fp32 rounding), d_feat of a given d_proj is bit-identical.  A 64-texel
 * tile of F / dF is ONE contiguous 64 * C * 4 byte piece instead of C row pieces of 256 bytes, and the sparse backward is traffic-bound
 * (profiles/r05u, r05v).
 * `tiles`: NULL = the whole map (dense forward / backward, nothing cleared); otherwise the flags of bts_mark_sampled_tiles (forward) or of
 * bts_render_bwd's kept pair (backward, with clear_after as in bts_project_features_bwd_tiles). */
int bts_project_features_cl(const BtsFieldCfg* cfg, const float* feat_nhwc, const float* mlp_params, int32_t N, const uint8_t* tiles,
                            float* proj_nhwc, void* stream);
int bts_project_features_bwd_cl(const BtsFieldCfg* cfg, const float* feat_nhwc, float* d_proj_nhwc, uint8_t* tiles, const float* mlp_params,
                                int32_t N, float* d_feat_nhwc, float* d_mlp_params, int32_t clear_after, void* stream);

/* BTSNet.forward on raw points (models_bts.py:266-338): xyz (n, P, 3) -> rgb (n, P, nv*3), invalid (n, P, max(nv,1)),
 * sigma (n, P).  only_density != 0 skips the colour taps: rgb may be NULL and invalid is (n, P, 1). */
int bts_field_query(const BtsFieldCfg* cfg, const BtsFieldTensors* t, const float* xyz, int32_t P, int32_t only_density,
                    float* rgb, float* invalid, float* sigma, void* stream);

/* The occupancy profile of scripts/inference_setup.py:201-229 (render_profile) in one pass: xyz (n, Y * columns, 3) is a dense grid
 * of query points with the vertical level y SLOWEST (get_pts' order, :169-187: point (y, c) at index y * columns + c).  Per column
 * c: sigma := 1 where any view flags the point invalid (:219), running sum over the levels y = 0 .. Y-1 (:224), profile (n, columns)
 * = (number of levels whose running sum is <= threshold) / Y (:225; threshold 8 in the reference).  only_density != 0: only the
 * encoder view's frustum test counts as invalid (the LiDAR / 3D-bbox evaluators' query mode, evaluator_lidar.py:300-308).
 * sigma (n, Y * columns), when given, also receives the raw densities.  Y <= 64 (the reference uses 64); needs proj_nhwc.  ABI 3.
 * Rounding: the running sum is a wave-wide parallel scan, the reference's a sequential fp32 cumsum -- on a column whose running sum
 * passes within ~2e-4 of the threshold the `<=` test can fall the other way and move that column's value by 1 / Y (the parity tests
 * compare the columns that are decided by a wider margin, tests/_cases.py: decided_columns). */
int bts_occupancy_profile(const BtsFieldCfg* cfg, const BtsFieldTensors* t, const float* xyz, int32_t Y, int32_t columns, float threshold,
                          int32_t only_density, float* profile, float* sigma, void* stream);

/* Layout changes at the hand-off from the (PyTorch) encoder: F (N, C, H, W) <-> (N, H, W, C); frames (N, 3, H, W) ->
 * (N, H, W, 4) with `scale`*x + `shift` applied (encode's x*0.5+0.5, models_bts.py:82). */
int bts_nchw_to_nhwc(const float* src, float* dst, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);
int bts_nhwc_to_nchw(const float* src, float* dst, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);
int bts_pack_rgb(const float* src_nchw, float* dst_nhwc4, int32_t N, int32_t H, int32_t W, float scale, float shift,
                 void* stream);

/* gen_rays (models/common/util/util.py:244-273 + unproj_map :113-149) for full images:
 * poses_c2w (V, 4, 4), projs (V, 3, 3) -> rays (V, H, W, 8). */
int bts_gen_rays(const float* poses_c2w, const float* projs, int32_t V, int32_t H, int32_t W, float z_near, float z_far,
                 int32_t norm_dir, float* rays, void* stream);

/* Photometric loss on the renderer's patch outputs, forward AND backward in one pass.  Replaces, for criterion "l1+ssim",
 * ReconstructionLoss.__call__ (models/bts/model/loss.py:83-293): compute_errors_l1ssim (:10-18) with SSIM
 * (models/common/model/layers.py:79-150: 3x3 Gaussian window, zero padding, comp_mode), the minimum over the nv render views,
 * the invalid-ray policy (:100-118) and edge_aware_smoothness (:21-40, masked as in :262-266).
 * Rays are in PatchRaySampler order: B = n_patches * patch_h * patch_w, patch-major, then row, then column; patch_h*patch_w <= 64.
 * parts (n_patches, 4) receives per patch [sum of the rgb term, sum of the smoothness term, number of invalid rays, 0]: the caller
 * forms  loss = scale_rgb * sum(parts[:,0]) + scale_eas * sum(parts[:,1])  (scale_rgb = (lambda_coarse + lambda_fine) / B ...).
 * g_rgb (B, nv, 3) / g_depth (B), when given, receive d loss / d rgb and d loss / d depth for exactly that loss. */
typedef struct {
  const float* rgb;      /* (B, nv, 3) rendered colours per view */
  const float* depth;    /* (B) expected ray termination depth (may be NULL without edge_aware_smoothness) */
  const float* weights;  /* (B, K)      needed by invalid_policy 2   } unless invalid_wsum / invalid_any */
  const float* invalid;  /* (B, K, nv)  needed by invalid_policy 1, 2 } below are given                  */
  const float* rgb_gt;   /* (B, 3) */
  float* parts;          /* (n_patches, 4) */
  float* g_rgb;          /* (B, nv, 3) or NULL */
  float* g_depth;        /* (B) or NULL */
  int32_t n_patches, patch_h, patch_w, nv, K;
  int32_t invalid_policy;          /* 0 none, 1 strict, 2 weight_guided */
  int32_t edge_aware_smoothness;   /* 0 / 1 */
  float scale_rgb, scale_eas;
  /* ABI 2: the renderer's per-ray reductions (BtsRenderArgs.invalid_wsum / invalid_any); when given they replace weights / invalid */
  const float* invalid_wsum;   /* (B, nv) or NULL   policy 2 */
  const float* invalid_any;    /* (B, nv) or NULL   policy 1 */
} BtsLossArgs;
int bts_photometric_loss(const BtsLossArgs* args, void* stream);

/* PatchRaySampler.sample (models/bts/model/ray_sampler.py:125-162) on device: for each of the n samples, P patches given by
 * (view, y0, x0) triples (int32, (n, P) each; the caller draws them -- the reference uses the CPU RNG) of ph x pw pixels.
 * poses_c2w (n, v, 4, 4), projs (n, v, 3, 3), images (n, v, c, H, W) or NULL -> rays (n, P*ph*pw, 8) and, with images,
 * rgb_gt (n, P*ph*pw, c), in the reference's order (patch, row, column).  Patch pixels must lie inside the frame. */
int bts_patch_rays(const float* poses_c2w, const float* projs, const float* images, const int32_t* patch_v, const int32_t* patch_y,
                   const int32_t* patch_x, int32_t n, int32_t v, int32_t c, int32_t H, int32_t W, int32_t P, int32_t ph, int32_t pw,
                   float z_near, float z_far, int32_t norm_dir, float* rays, float* rgb_gt, void* stream);

/* NeRFRenderer.sample_coarse (nerf.py:103-123) with the uniform jitter u (B, K) in [0,1) supplied by the caller. */
int bts_sample_coarse(const float* rays, const float* u, int64_t B, int32_t K, int32_t lindisp, float* z_samp,
                      void* stream);

/* distance_to_z (utils/projection_operations.py:4-16): depths (N, H, W), inv_K (N, 3, 3) = inverse(projs) -> z (N, H, W) */
int bts_distance_to_z(const float* depths, const float* inv_K, int32_t N, int32_t H, int32_t W, float* out, void* stream);

/* Batched inverse of N dim x dim matrices, dim = 3 or 4 (row-major).  Stands in for torch.inverse on the c2w poses in
 * BTSNet.encode (models/bts/model/models_bts.py:71) and on the intrinsics in distance_to_z (utils/projection_operations.py:9):
 * fp64 Gauss-Jordan with partial pivoting rounded to fp32, enqueued on the stream (no solver library, no host sync). */
int bts_invert_small(const float* src, float* dst, int32_t N, int32_t dim, void* stream);


/* ---------------------------------------------------------------------------------------------------------------------------------
 * ABI 7: a training step's share of the renderer as TWO calls.
 *
 * The reference's training forward (BTSWrapper.forward, models/bts/trainer.py:208-259, and the criterion call of
 * utils/base_trainer.py:287-297) is, after the CNN: encode's hand-over (models_bts.py:65-136), PatchRaySampler.sample
 * (ray_sampler.py:125-162), one render per scale (trainer.py:220-259 -> nerf.py:315-375), reconstruct (views only) and
 * ReconstructionLoss.__call__ (loss.py:83-293); autograd then walks the same chain backwards.  Entry by entry that is ~20 calls into
 * this library per step with allocator, autograd and ctypes work around each -- at exp_kitti_raw.yaml's shapes the HOST then takes
 * longer to issue the step (1.1 ms) than the GPU to run it (0.5 ms).  bts_train_step_fwd enqueues the whole forward chain, and
 * bts_train_step_bwd the whole backward chain, from one call each: the same kernels in the same order with the same arguments as the
 * entry-by-entry path (bit-identical forward results; the backward's float atomics make its summation order run-to-run variable
 * either way).  Every buffer is the caller's: nothing is allocated, nothing synchronises, the struct is only read.
 *
 * Frames: images (n, v, 3, H, W) in [-1, 1] as the data loader delivers them; x * img_scale + img_shift (0.5, 0.5: models_bts.py:82,
 * trainer.py's image processor) happens where a frame is read.  id_encoder / ids_render / ids_loss index the v frames of a batch
 * element (trainer.py:118-190 draws them per step); the patches' view index patch_v indexes ids_loss (the reference samples from
 * images_ip[:, ids_loss]).
 * Per scale s (trainer.py:220-242 "multiscale": encoder.scales; otherwise one scale): the decoder's map at ITS size (feat_shift, ABI
 * 4), this render's jitter, and the loss term on this scale's rgb / depth with scale 0's invalid-ray reductions (loss.py:100-118).
 * loss_matrix (9 x 3 n_scales, row-major): the logging dict of loss.py:219-229 and the loss itself are LINEAR in the per-scale sums
 * [rgb term, smoothness term, invalid rays] the loss pass returns; row 8 is the loss (its entries 3 s and 3 s + 1 are also the
 * coefficients d loss / d sums the backward applies), rows 0-7 the dict's other entries, in the order loss_rgb_coarse, loss_rgb_fine,
 * loss_ray_entropy, loss_depth_reg, loss_alpha_reg, loss_eas, loss_depth_smoothness, loss_invalid_ratio.
 * --------------------------------------------------------------------------------------------------------------------------------- */
#define BTS_MAX_SCALES 4
#define BTS_MAX_LOSS_VIEWS 16

typedef struct BtsTrainScale {
  /* inputs */
  const float* feat_nchw;    /* (n, C, H >> feat_shift, W >> feat_shift) the decoder's map of this scale */
  const float* jitter;       /* (n*Bp, K) in [0, 1): this render's stratified jitter (the caller's torch.rand draw, nerf.py:112) */
  /* outputs of the forward */
  float* rgb;                /* (n*Bp, nv*3) */
  float* depth;              /* (n*Bp) */
  float* invalid_wsum;       /* (n*Bp, nv)  } BtsRenderArgs.invalid_wsum / invalid_any (ABI 2) */
  float* invalid_any;        /* (n*Bp, nv)  } */
  /* state the forward leaves for the backward (contents need no initialisation) */
  float* proj_nhwc;          /* (n, H >> s, W >> s, Hd): valid in the tiles flagged in sampled_tiles only (ABI 6) */
  uint8_t* sampled_tiles;    /* (n, tiles of the scale's map), 4-byte aligned */
  float* z_samp;             /* (n*Bp, K) */
  float* sigma_raw;          /* (n*Bp, K) */
  float* trans;              /* (n*Bp, K) */
  float* rgb_samps;          /* (n*Bp, K, nv*3) */
  float* loss_parts;         /* (n*P, 4) per-patch sums of the loss pass */
  float* g_rgb;              /* (n*Bp, nv*3) d (sum of the rgb term) / d rgb            } written by the forward's loss pass */
  float* g_depth;            /* (n*Bp)       d (sum of the smoothness term) / d depth   } */
  float* gs_rgb;             /* the two above times (loss_matrix[8][3 s], [3 s + 1]) * upstream gradient: backward scratch */
  float* gs_depth;
  /* backward */
  float* d_proj_nhwc;        /* (n, H >> s, W >> s, Hd) ALL ZERO on entry, all zero again on return (the kept pair of ABI 6) */
  uint8_t* d_proj_tiles;     /* (n, tiles)              ALL ZERO on entry, all zero again on return */
  float* d_feat_nchw;        /* (n, C, H >> s, W >> s) gradient of feat_nchw, WRITTEN by the backward, or NULL */
  int32_t feat_shift;
  int32_t feat_channels_last; /* ABI 8: 1 = feat_nchw and d_feat_nchw of this scale are channels-last, (n, h, w, C) in memory (see
                               * bts_project_features_cl); 0 = NCHW */
} BtsTrainScale;

typedef struct BtsTrainStep {
  BtsFieldCfg cfg;           /* n, H, W (frame size), C, d_hidden, ..., nv = number of render views; feat_shift is per scale (ignored
                              * here); enc_render_view as in ABI 5 (-1 is always correct) */
  int32_t v;                 /* frames per batch element */
  int32_t id_encoder;        /* ids_encoder[0] */
  int32_t ids_render[BTS_MAX_VIEWS];
  int32_t n_loss;            /* number of loss frames, <= BTS_MAX_LOSS_VIEWS */
  int32_t ids_loss[BTS_MAX_LOSS_VIEWS];
  int32_t P, ph, pw;         /* patches per batch element, patch size: Bp = P * ph * pw rays per batch element, ph * pw <= 64 */
  int32_t K;                 /* samples per ray */
  int32_t lindisp, hard_alpha_cap;          /* nerf.py:117, :285-286 */
  int32_t invalid_policy;    /* 0 none, 1 strict, 2 weight_guided (BtsLossArgs) */
  int32_t edge_aware_smoothness;
  int32_t n_scales;          /* 1 .. BTS_MAX_SCALES */
  /* 1: the scales' chains of kernels run side by side on queues of the library's own, forked from and joined back into the caller's
   * stream inside the call (the scales share nothing but read-only inputs and the atomically accumulated d_mlp_params); the backward then
   * needs bwd_workspace_bytes >= n_scales * (bts_render_bwd_workspace rounded up to 256), else it runs them one after the other.  0:
   * everything on the caller's stream, in order.  Results are the same either way (the backward's float atomics are unordered anyway). */
  int32_t concurrent_scales;
  float z_near, z_far;       /* the ray sampler's (ray_sampler.py:108-123) */
  float img_scale, img_shift;
  float loss_matrix[9 * 3 * BTS_MAX_SCALES];   /* (9, 3 n_scales) row-major, see above */
  /* inputs */
  const float* images;       /* (n, v, 3, H, W) */
  const float* Ks;           /* (n, v, 3, 3) normalised intrinsics */
  const float* poses_c2w;    /* (n, v, 4, 4) */
  const int32_t* patch_v;    /* (n, P) index into ids_loss  } the caller's draws (the reference uses the CPU RNG, */
  const int32_t* patch_y;    /* (n, P) top row              }  ray_sampler.py:141-143)                            */
  const int32_t* patch_x;    /* (n, P) left column          } */
  const float* mlp_params;   /* packed (see the top of this file) */
  const float* empty_feature;/* (C) or NULL (learn_empty) */
  /* outputs of the forward */
  float* rays;               /* (n*Bp, 8) */
  float* rgb_gt;             /* (n*Bp, 3) colours of the patch pixels in [0, 1] */
  float* loss_vals;          /* (9) loss_matrix . sums: [8] = the loss */
  /* scratch (caller-owned, contents need no initialisation) */
  float* cams;               /* n * (9 + 16 + nv * (9 + 16)) floats: K_enc (n, 9), w2c_enc (n, 16), K_r (n, nv, 9), w2c_r (n, nv, 16) */
  float* imgs_nhwc4;         /* (n, nv, H, W, 4) */
  void* bwd_workspace;       /* max over the scales of bts_render_bwd_workspace */
  size_t bwd_workspace_bytes;
  float* d_empty_proj;       /* (Hd) or NULL */
  /* outputs of the backward */
  float* d_mlp_params;       /* packed like mlp_params, WRITTEN (zero-filled inside, then accumulated over the scales), or NULL */
  float* d_empty_feature;    /* (C) WRITTEN, or NULL */
  BtsTrainScale scale[BTS_MAX_SCALES];
} BtsTrainStep;

/* Forward: cameras (bts_invert_small on the encoder / render poses), bts_pack_rgb of the render frames, bts_patch_rays, then per scale
 * bts_mark_sampled_tiles + bts_project_features_tiles + bts_render_fwd (lean outputs: rgb, depth, the invalid-ray reductions and the
 * backward's saved state) + bts_photometric_loss, and one reduction of the per-patch sums into loss_vals. */
int bts_train_step_fwd(const BtsTrainStep* st, void* stream);
/* Backward of the above for the SAME struct contents: g_loss = device pointer to the upstream gradient of loss_vals[8] (a scalar), or
 * NULL for 1.  Per scale bts_render_bwd (adding into the kept (d_proj, tiles) pair) + bts_project_features_bwd_tiles (clear_after);
 * d_mlp_params / d_empty_feature collect every scale's contribution. */
int bts_train_step_bwd(const BtsTrainStep* st, const float* g_loss, void* stream);


/* ---------------------------------------------------------------------------------------------------------------------------------
 * ABI 7: the Monodepth2 decoder's last convolutions, the layers that PRODUCE the renderer's scale-0
 * feature map.  One operator:
 *     y = [ELU]( conv3x3( reflect_pad1( [nearest x2]( x ) ), weight ) + bias )
 * = Conv3x3 / ConvBlock of models/common/model/layers.py:11-40 (ReflectionPad2d(1) + 3 x 3 convolution [+ ELU, alpha 1]) with the
 * decoder's nearest x2 upsampling (monodepth2.py:225) folded into the read of x.  With d_out = 64 the tail is (monodepth2.py:189-239)
 *     upconv(0,0): {elu}            on (N, H/2, W/2, 64)   ->  upconv(0,1): {up2, elu}  ->  dispconv(0): {out_nchw}   at (N, H, W, 64)
 * and its last output, NCHW, is exactly what bts_project_features / bts_train_step_fwd take as feat_nchw.  Tensors are channels-last
 * (N, H, W, C) fp32 -- the memory of a torch tensor in channels_last format -- except an NCHW output when out_nchw is set.  C = 64.
 * fp32 in, fp32 out, fp32 accuracy: every operand enters the bf16 matrix pipe as the exact sum of three bf16 terms, six products per
 * pair reproduce the fp32 product to 2^-24, accumulation in fp32 (the summation order differs from a library convolution's).  A
 * non-finite input gives NaN where a library convolution gives Inf.
 * --------------------------------------------------------------------------------------------------------------------------------- */
typedef struct BtsConv3x3 {
  int32_t N, H, W;       /* OUTPUT size; the input is (N, H, W, C), or (N, H / 2, W / 2, C) with up2 (H, W even) */
  int32_t C;             /* input = output channels: 64 */
  int32_t up2;           /* 1: read x through a nearest x2 upsampling */
  int32_t elu;           /* 1: ELU on the output (ConvBlock) */
  int32_t out_nchw;      /* 1: y is (N, C, H, W); not together with elu */
  int32_t reserved_;
  const float* x;        /* input, channels-last */
  const float* weight;   /* (C, C, 3, 3) = nn.Conv2d.weight */
  const float* bias;     /* (C) or NULL */
  float* y;              /* output (written by bts_conv3x3_fwd; READ by bts_conv3x3_bwd of an ELU layer: elu' comes from the output) */
} BtsConv3x3;

int bts_conv3x3_fwd(const BtsConv3x3* c, void* stream);
/* Backward for the same struct: g_y = gradient of y in y's layout.  d_x (layout of x), d_weight (C, C, 3, 3), d_bias (C) are WRITTEN (not
 * accumulated); any of them may be NULL.  workspace: bts_conv3x3_bwd_workspace(c) bytes (g_y times elu' as a channels-last tensor + the
 * weight gradient's partial sums); contents need no initialisation.  Deterministic: no atomics anywhere. */
size_t bts_conv3x3_bwd_workspace(const BtsConv3x3* c);
int bts_conv3x3_bwd(const BtsConv3x3* c, const float* g_y, void* workspace, size_t workspace_bytes, float* d_x, float* d_weight, float* d_bias,
                    void* stream);


/* ---------------------------------------------------------------------------------------------------------------------------------
 * ABI 7: an evaluation frame in ONE call -- BTSWrapper.forward of models/bts/evaluator.py:60-79 after the CNN: encode's hand-over
 * (models_bts.py:65-136: cameras, rgb0 packing of the render frames, the projection of the feature map), ImageRaySampler.sample
 * (ray_sampler.py:233-260: the rays of every pixel of the `n_ray_views` frames), the render with sample_coarse inside (nerf.py:315-375),
 * distance_to_z (utils/projection_operations.py:4-16).  The same kernels with the same arguments as the entry-by-entry path, enqueued
 * without the host work between them (at 1 ms per frame that is 8 % of the frame).  Every buffer is the caller's.
 * --------------------------------------------------------------------------------------------------------------------------------- */
typedef struct BtsEvalFrame {
  BtsFieldCfg cfg;           /* n, H, W, C, ..., nv = number of render (colour) views; enc_render_view as in ABI 5 */
  int32_t v;                 /* frames per batch element */
  int32_t id_encoder;
  int32_t ids_render[BTS_MAX_VIEWS];
  int32_t K, lindisp, hard_alpha_cap, norm_dir;
  float z_near, z_far;       /* the ray sampler's */
  float img_scale, img_shift;
  /* inputs */
  const float* images;       /* (n, v, 3, H, W) */
  const float* Ks;           /* (n, v, 3, 3) */
  const float* poses_c2w;    /* (n, v, 4, 4) */
  const float* feat_nchw;    /* (n, C, H, W) the encoder's scale-0 map */
  const float* mlp_params;
  const float* empty_feature;/* (C) or NULL */
  const float* jitter;       /* (n * v * H * W, K) in [0, 1) */
  /* scratch */
  float* cams;               /* n * (25 + nv * 25) floats, as in BtsTrainStep */
  float* imgs_nhwc4;         /* (n, nv, H, W, 4) */
  float* proj_nhwc;          /* (n, H, W, Hd) */
  float* inv_K;              /* (n, v, 3, 3) */
  /* outputs: rays of ALL v frames of every batch element, B = n * v * H * W */
  float* rays;               /* (B, 8) */
  float* rgb;                /* (B, nv*3) */
  float* depth;              /* (B) distance along the ray */
  float* depth_z;            /* (B) = (n, v, H, W): distance_to_z of `depth`, or NULL */
  float* weights;            /* (B, K) or NULL */
  float* alphas;             /* (B, K) or NULL */
  float* invalid;            /* (B, K, nv) or NULL */
  int32_t feat_channels_last; /* ABI 9: 1 = feat_nchw is channels-last, (n, H, W, C) in memory -- what the shipped Monodepth2 decoder writes
                              * (as BtsTrainScale.feat_channels_last, ABI 8): read as it is, no layout pass.  Appended: every ABI 8 offset stands */
  int32_t reserved_;
} BtsEvalFrame;
int bts_eval_frame(const BtsEvalFrame* f, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BTS_RENDER_H */
