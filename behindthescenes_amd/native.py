"""Tensor-level access to the HIP renderer: validates torch tensors (device, dtype, contiguity, shape), hands raw
device pointers + the current HIP stream to the C ABI (include/bts_render.h) and wraps the forward/backward pair in a
``torch.autograd.Function``.  PyTorch is plumbing here (memory, streams, autograd bookkeeping); every computation of
the render path happens in libbts_render.so and nothing in this module falls back to torch ops."""
from dataclasses import dataclass
from typing import Optional

import ctypes as C
import weakref

import torch

from . import _lib
from ._lib import BtsFieldCfg, BtsFieldTensors, BtsNativeError, BtsRenderArgs, BtsRenderGrads


@dataclass(frozen=True)
class FieldSpec:
    """Static shape/config of the field (mirrors BtsFieldCfg minus n/H/W/nv, which come from the tensors)."""
    C: int
    d_hidden: int
    n_blocks: int
    num_freqs: int = 6
    freq_factor: float = 1.5
    d_min: float = 3.0
    d_max: float = 80.0
    inv_z: bool = True
    code_mode: str = "z"
    learn_empty: bool = False
    empty_empty: bool = False
    # the geometry of the map's tile flags (BtsFieldCfg.tile_blocks, ABI 9): 16 x 4 blocks (True: faster with a channels-last feature map)
    # or runs of 64 texels (False: faster with an NCHW one).  Every call on one (d_proj, tiles) pair must use the same spec.
    tile_blocks: bool = False

    @property
    def d_in(self):
        return self.C + 3 + 6 * self.num_freqs

    def mlp_param_count(self):
        hd = self.d_hidden
        return hd * self.d_in + hd + self.n_blocks * (2 * hd * hd + 2 * hd) + hd + 1


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(t: torch.Tensor):
    """The current HIP stream of the tensor's device.  The library launches on the CURRENT device, so a tensor that lives elsewhere
    is rejected here instead of enqueuing its pointers on the wrong GPU."""
    if t.device.index is not None and t.device.index != torch.cuda.current_device():
        raise BtsNativeError(f"tensor on {t.device} but the current device is cuda:{torch.cuda.current_device()}: wrap the call in "
                             "torch.cuda.device(...) (one process per GPU never hits this)")
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _req(t: torch.Tensor, name: str, shape=None):
    if not isinstance(t, torch.Tensor):
        raise BtsNativeError(f"{name}: expected a tensor")
    if not t.is_cuda:
        raise BtsNativeError(f"{name}: must live on the GPU (got {t.device}); the HIP renderer has no CPU path")
    if t.dtype != torch.float32:
        raise BtsNativeError(f"{name}: must be float32 (got {t.dtype})")
    if not t.is_contiguous():
        raise BtsNativeError(f"{name}: must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise BtsNativeError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


# --------------------------------------------------------------------------------------------------------------
# layout / edge kernels
# --------------------------------------------------------------------------------------------------------------
def nchw_to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """(N, C, H, W) -> (N, H, W, C) copy (bts_nchw_to_nhwc)."""
    _req(x, "x")
    N, Cc, H, W = x.shape
    out = torch.empty((N, H, W, Cc), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().bts_nchw_to_nhwc(_ptr(x), _ptr(out), N, Cc, H, W, _stream(x)), "bts_nchw_to_nhwc")
    return out


def nhwc_to_nchw(x: torch.Tensor) -> torch.Tensor:
    _req(x, "x")
    N, H, W, Cc = x.shape
    out = torch.empty((N, Cc, H, W), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().bts_nhwc_to_nchw(_ptr(x), _ptr(out), N, Cc, H, W, _stream(x)), "bts_nhwc_to_nchw")
    return out


def pack_rgb(images: torch.Tensor, scale: float = 1.0, shift: float = 0.0) -> torch.Tensor:
    """(..., 3, H, W) -> (..., H, W, 4) rgb0 with x*scale+shift (bts_pack_rgb)."""
    _req(images, "images")
    lead, (c, H, W) = images.shape[:-3], images.shape[-3:]
    if c != 3:
        raise BtsNativeError(f"images: expected 3 channels, got {c}")
    N = 1
    for d in lead:
        N *= d
    out = torch.empty(tuple(lead) + (H, W, 4), device=images.device, dtype=torch.float32)
    _lib.check(_lib.load().bts_pack_rgb(_ptr(images), _ptr(out), N, H, W, scale, shift, _stream(images)), "bts_pack_rgb")
    return out


def gen_rays(poses_c2w: torch.Tensor, projs: torch.Tensor, H: int, W: int, z_near: float, z_far: float,
             norm_dir: bool = True) -> torch.Tensor:
    """poses (V,4,4), projs (V,3,3) -> rays (V,H,W,8) (bts_gen_rays)."""
    V = poses_c2w.shape[0]
    _req(poses_c2w, "poses_c2w", (V, 4, 4)), _req(projs, "projs", (V, 3, 3))
    out = torch.empty((V, H, W, 8), device=poses_c2w.device, dtype=torch.float32)
    _lib.check(_lib.load().bts_gen_rays(_ptr(poses_c2w), _ptr(projs), V, H, W, z_near, z_far, int(norm_dir), _ptr(out),
                                        _stream(out)), "bts_gen_rays")
    return out


def patch_rays(poses_c2w, projs, images, patch_v, patch_y, patch_x, ph: int, pw: int, z_near: float, z_far: float, norm_dir: bool = True):
    """poses (n,v,4,4), projs (n,v,3,3), images (n,v,c,H,W) | None with (H, W) given via ``images`` or a (H, W) tuple; patch_* (n,P)
    int32 -> rays (n, P*ph*pw, 8), rgb_gt (n, P*ph*pw, c) | None  (bts_patch_rays)."""
    n, v = poses_c2w.shape[:2]
    _req(poses_c2w, "poses_c2w", (n, v, 4, 4)), _req(projs, "projs", (n, v, 3, 3))
    if isinstance(images, tuple):
        (H, W), c, images = images, 0, None
    else:
        _req(images, "images")
        c, H, W = images.shape[2:]
    P = patch_v.shape[1]
    if not isinstance(images, tuple) and ((ph > H) or (pw > W)):
        raise BtsNativeError(f"patch {ph}x{pw} does not fit a {H}x{W} frame")
    for t, nme in ((patch_v, "patch_v"), (patch_y, "patch_y"), (patch_x, "patch_x")):
        if t.dtype != torch.int32 or not t.is_cuda or not t.is_contiguous() or tuple(t.shape) != (n, P):
            raise BtsNativeError(f"{nme}: expected a contiguous int32 device tensor of shape {(n, P)}")
    dev = poses_c2w.device
    rays = torch.empty((n, P * ph * pw, 8), device=dev, dtype=torch.float32)
    gt = torch.empty((n, P * ph * pw, c), device=dev, dtype=torch.float32) if images is not None else None
    _lib.check(_lib.load().bts_patch_rays(_ptr(poses_c2w), _ptr(projs), _ptr(images), patch_v.data_ptr(), patch_y.data_ptr(),
                                          patch_x.data_ptr(), n, v, c, H, W, P, ph, pw, z_near, z_far, int(norm_dir), _ptr(rays),
                                          _ptr(gt), _stream(rays)), "bts_patch_rays")
    return rays, gt


INVALID_POLICIES = {None: 0, "none": 0, "strict": 1, "weight_guided": 2}


def photometric_loss(rgb, depth, weights, invalid, rgb_gt, patch_h: int, patch_w: int, invalid_policy, eas: bool,
                     scale_rgb: float, scale_eas: float, need_grad: bool = True, invalid_wsum=None, invalid_any=None):
    """Patch-ordered renderer outputs -> per-patch partial sums and the loss gradients (bts_photometric_loss).
    rgb (B, nv*3), depth (B) | None, weights (B, K) | None, invalid (B, K, nv) | None, rgb_gt (B, 3)
    -> parts (B / (ph*pw), 4), g_rgb (B, nv*3) | None, g_depth (B) | None."""
    B = rgb_gt.shape[0]
    area = patch_h * patch_w
    if B % area:
        raise BtsNativeError(f"{B} rays are not a whole number of {patch_h}x{patch_w} patches")
    nv = rgb.shape[-1] // 3
    policy = INVALID_POLICIES[invalid_policy]
    _req(rgb, "rgb", (B, nv * 3)), _req(rgb_gt, "rgb_gt", (B, 3))
    K = 0
    # the renderer's per-ray reductions (render_fwd(want_invalid_sums=True)) stand in for the per-sample tensors
    sums = (policy == 2 and invalid_wsum is not None) or (policy == 1 and invalid_any is not None)
    if sums:
        _req(invalid_wsum if policy == 2 else invalid_any, "invalid_wsum / invalid_any", (B, nv))
        weights = invalid = None
    elif policy:
        K = invalid.shape[1]
        _req(invalid, "invalid", (B, K, nv))
        if policy == 2:
            _req(weights, "weights", (B, K))
    if eas:
        _req(depth, "depth", (B,))
    dev = rgb.device
    parts = torch.empty((B // area, 4), device=dev, dtype=torch.float32)
    g_rgb = torch.empty_like(rgb) if need_grad else None
    g_depth = torch.empty((B,), device=dev, dtype=torch.float32) if (need_grad and depth is not None) else None
    a = _lib.BtsLossArgs(rgb=rgb.data_ptr(), depth=None if depth is None else depth.data_ptr(),
                         weights=None if (weights is None or policy != 2) else weights.data_ptr(),
                         invalid=None if (invalid is None or not policy) else invalid.data_ptr(), rgb_gt=rgb_gt.data_ptr(),
                         parts=parts.data_ptr(), g_rgb=None if g_rgb is None else g_rgb.data_ptr(),
                         g_depth=None if g_depth is None else g_depth.data_ptr(), n_patches=B // area, patch_h=patch_h, patch_w=patch_w,
                         nv=nv, K=K, invalid_policy=policy, edge_aware_smoothness=int(bool(eas)), scale_rgb=scale_rgb,
                         scale_eas=scale_eas, invalid_wsum=invalid_wsum.data_ptr() if (sums and policy == 2) else None,
                         invalid_any=invalid_any.data_ptr() if (sums and policy == 1) else None)
    _lib.check(_lib.load().bts_photometric_loss(C.byref(a), _stream(rgb)), "bts_photometric_loss")
    return parts, g_rgb, g_depth


def sample_coarse(rays: torch.Tensor, u: torch.Tensor, lindisp: bool) -> torch.Tensor:
    """rays (B,8), u (B,K) uniform jitter -> z_samp (B,K) (bts_sample_coarse)."""
    B, K = u.shape
    _req(rays, "rays", (B, 8)), _req(u, "u")
    out = torch.empty_like(u)
    _lib.check(_lib.load().bts_sample_coarse(_ptr(rays), _ptr(u), B, K, int(lindisp), _ptr(out), _stream(out)),
               "bts_sample_coarse")
    return out


def invert_small(m: torch.Tensor) -> torch.Tensor:
    """(..., d, d) with d in {3, 4} -> inverses (bts_invert_small); the stand-in for torch.inverse on poses / intrinsics."""
    d = m.shape[-1]
    if m.shape[-2] != d or d not in (3, 4):
        raise BtsNativeError(f"invert_small: expected (..., 3, 3) or (..., 4, 4), got {tuple(m.shape)}")
    src = m.float().contiguous()
    _req(src, "matrices")
    out = torch.empty_like(src)
    N = src.numel() // (d * d)
    _lib.check(_lib.load().bts_invert_small(_ptr(src), _ptr(out), N, d, _stream(src)), "bts_invert_small")
    return out


def distance_to_z(depths: torch.Tensor, projs: torch.Tensor) -> torch.Tensor:
    """depths (n,nv,H,W), projs (n,nv,3,3) -> z (n,nv,H,W) (bts_invert_small + bts_distance_to_z)."""
    n, nv, H, W = depths.shape
    _req(depths, "depths")
    inv_K = invert_small(projs).reshape(n * nv, 3, 3)
    _req(inv_K, "inv_K", (n * nv, 3, 3))
    out = torch.empty_like(depths)
    _lib.check(_lib.load().bts_distance_to_z(_ptr(depths), _ptr(inv_K), n * nv, H, W, _ptr(out), _stream(out)),
               "bts_distance_to_z")
    return out


# --------------------------------------------------------------------------------------------------------------
# field state handed to the renderer
# --------------------------------------------------------------------------------------------------------------
def _spec_cfg(spec: FieldSpec, n=1, H=1, W=1, nv=0, feat_shift=0, enc_view=-1) -> BtsFieldCfg:
    return BtsFieldCfg(n=n, H=H, W=W, C=spec.C, d_hidden=spec.d_hidden, n_blocks=spec.n_blocks, nv=nv, num_freqs=spec.num_freqs,
                       code_mode={"z": 0, "distance": 1}[spec.code_mode], inv_z=int(spec.inv_z), learn_empty=int(spec.learn_empty),
                       empty_empty=int(spec.empty_empty), freq_factor=spec.freq_factor, d_min=spec.d_min, d_max=spec.d_max,
                       feat_shift=feat_shift, enc_render_view=enc_view if 0 <= enc_view < nv else -1, tile_blocks=int(spec.tile_blocks))


def proj_storage_order(d_hidden: int) -> torch.Tensor:
    """order[s] = hidden unit stored at channel s of proj_nhwc (bts_common.h: proj_hidden_of_storage): within each group of 32
    hidden units the renderer keeps the 16 accumulator rows of one lane half contiguous."""
    s = torch.arange(d_hidden)
    ht, r = s // 32, s % 32
    h, q, e = r // 16, (r // 4) % 4, r % 4
    return ht * 32 + 8 * q + 4 * h + e


def is_channels_last(t: torch.Tensor) -> bool:
    """A 4-d tensor whose memory is (N, H, W, C) -- torch's channels_last format -- and not also plain contiguous (C = 1 or H = W = 1)."""
    return t.dim() == 4 and not t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last)


def as_feature_map(t: torch.Tensor) -> torch.Tensor:
    """The encoder's map as the library takes it: fp32, dense, in the layout it ALREADY has when that is NCHW or channels-last (ABI 8:
    bts_project_features_cl -- the format MIOpen's NHWC convolutions and bts_conv3x3_fwd write), a contiguous copy otherwise."""
    t = t.float()
    return t if (t.is_contiguous() or is_channels_last(t)) else t.contiguous()


def project_features(spec: FieldSpec, feat_nchw: torch.Tensor, mlp_params: torch.Tensor, tiles: Optional[torch.Tensor] = None) -> torch.Tensor:
    """F (N,C,H,W) -> G (N,H,W,Hd) = F . w_in[:, :C]^T with the channels in storage order (proj_storage_order)
    (bts_project_features).  With ``tiles`` (N, proj_tile_count) uint8 only the flagged 64-texel tiles are evaluated and the rest of G
    is UNINITIALISED (bts_project_features_tiles): a map for the render whose samples ``mark_sampled_tiles`` flagged, nothing else.
    A channels_last F (memory (N,H,W,C)) is read as it is (bts_project_features_cl): the same G to fp32 rounding (another summation order)."""
    N, Cc, H, W = feat_nchw.shape
    cl = is_channels_last(feat_nchw)
    if cl:
        _req(feat_nchw.permute(0, 2, 3, 1), "feat (channels_last)", (N, H, W, spec.C))
    else:
        _req(feat_nchw, "feat_nchw", (N, spec.C, H, W))
    _req(mlp_params, "mlp_params", (spec.mlp_param_count(),))
    out = torch.empty((N, H, W, spec.d_hidden), device=feat_nchw.device, dtype=torch.float32)
    cfg = _spec_cfg(spec, N, H, W)
    if tiles is not None:
        if tiles.dtype != torch.uint8 or not tiles.is_contiguous() or tiles.numel() != N * proj_tile_count(spec, H, W) or tiles.device != out.device:
            raise ValueError("tiles: expected a contiguous uint8 tensor of (N, proj_tile_count) on the map's device")
    if cl:
        _lib.check(_lib.load().bts_project_features_cl(C.byref(cfg), _ptr(feat_nchw), _ptr(mlp_params), N, _ptr(tiles), _ptr(out), _stream(out)),
                   "bts_project_features_cl")
    elif tiles is not None:
        _lib.check(_lib.load().bts_project_features_tiles(C.byref(cfg), _ptr(feat_nchw), _ptr(mlp_params), N, _ptr(tiles), _ptr(out), _stream(out)),
                   "bts_project_features_tiles")
    else:
        _lib.check(_lib.load().bts_project_features(C.byref(cfg), _ptr(feat_nchw), _ptr(mlp_params), N, _ptr(out), _stream(out)),
                   "bts_project_features")
    return out


def mark_sampled_tiles(spec: FieldSpec, n: int, H: int, W: int, feat_shift: int, K_enc, w2c_enc, rays, z_samp=None, jitter=None, lindisp=True):
    """Flags (n, tiles per image of the (H >> s, W >> s) map) uint8 of the 64-texel tiles of the projected map that a render of
    ``rays`` (n*Bp, 8) at the depths ``z_samp`` (n*Bp, K) -- or, as bts_render_fwd forms them, from ``jitter`` -- reads
    (bts_mark_sampled_tiles: the render kernels' own routines, the same texels bit for bit)."""
    src = z_samp if z_samp is not None else jitter
    _req(rays, "rays", (src.shape[0], 8)), _req(src, "z_samp / jitter"), _req(K_enc, "K_enc", (n, 3, 3)), _req(w2c_enc, "w2c_enc", (n, 4, 4))
    cfg = _spec_cfg(spec, n, H, W, feat_shift=feat_shift)
    tiles = torch.zeros((n, proj_tile_count(spec, H >> feat_shift, W >> feat_shift)), device=rays.device, dtype=torch.uint8)
    args = BtsRenderArgs(rays_per_sample=rays.shape[0] // n, K=src.shape[1], rays=rays.data_ptr(), z_samp=None if z_samp is None else z_samp.data_ptr(),
                         jitter=None if z_samp is not None else jitter.data_ptr(), lindisp=int(bool(lindisp)), reserved_=0)
    _lib.check(_lib.load().bts_mark_sampled_tiles(C.byref(cfg), _ptr(K_enc), _ptr(w2c_enc), C.byref(args), _ptr(tiles), _stream(rays)),
               "bts_mark_sampled_tiles")
    return tiles


def project_features_bwd(spec: FieldSpec, feat_nchw, d_proj, mlp_params, need_feat=True, need_mlp=True, tiles=None, clear_after=False):
    """-> (d_feat_nchw | None, d_mlp_params | None) (bts_project_features_bwd).  With ``tiles`` (N, proj_tile_count) uint8 -- the flags
    ``render_bwd`` set next to a d_proj that was all zero before -- only the flagged 64-texel tiles of d_proj are read
    (bts_project_features_bwd_tiles); ``clear_after`` returns the pair to all zero.  For a channels_last F the gradient comes back
    channels_last as well (bts_project_features_bwd_cl): same values, a 64-texel tile of either is one contiguous piece."""
    N, Cc, H, W = feat_nchw.shape
    cl = is_channels_last(feat_nchw)
    if cl:
        _req(feat_nchw.permute(0, 2, 3, 1), "feat (channels_last)")
    else:
        _req(feat_nchw, "feat_nchw")
    _req(d_proj, "d_proj", (N, H, W, spec.d_hidden)), _req(mlp_params, "mlp_params")
    d_feat = torch.empty_like(feat_nchw) if need_feat else None          # (preserve_format: channels_last stays channels_last)
    d_mlp = torch.zeros_like(mlp_params) if need_mlp else None
    cfg = _spec_cfg(spec, N, H, W)
    lib = _lib.load()
    if tiles is not None:
        if tiles.dtype != torch.uint8 or not tiles.is_contiguous() or tiles.numel() != N * proj_tile_count(spec, H, W) or tiles.device != d_proj.device:
            raise ValueError("tiles: expected a contiguous uint8 tensor of (N, proj_tile_count) on d_proj's device")
    if cl:
        _lib.check(lib.bts_project_features_bwd_cl(C.byref(cfg), _ptr(feat_nchw), _ptr(d_proj), _ptr(tiles), _ptr(mlp_params), N, _ptr(d_feat),
                                                   _ptr(d_mlp), 1 if (clear_after and tiles is not None) else 0, _stream(d_proj)),
                   "bts_project_features_bwd_cl")
    elif tiles is not None:
        _lib.check(lib.bts_project_features_bwd_tiles(C.byref(cfg), _ptr(feat_nchw), _ptr(d_proj), _ptr(tiles), _ptr(mlp_params), N, _ptr(d_feat),
                                                      _ptr(d_mlp), 1 if clear_after else 0, _stream(d_proj)), "bts_project_features_bwd_tiles")
    else:
        _lib.check(lib.bts_project_features_bwd(C.byref(cfg), _ptr(feat_nchw), _ptr(d_proj), _ptr(mlp_params), N, _ptr(d_feat),
                                                _ptr(d_mlp), _stream(d_proj)), "bts_project_features_bwd")
    return d_feat, d_mlp


def proj_tile_map(H: int, W: int, blocks: bool = False) -> torch.Tensor:
    """(H, W) int64: the tile (0 .. proj_tile_count - 1) of every texel of an (H, W) projected map -- the geometry of csrc/bts_common.h:
    ``blocks`` (FieldSpec.tile_blocks) and H a multiple of 4, W of 16: blocks of 4 rows x 16 texels, numbered row-major; else 64
    consecutive texels of the row-major map.  What `tiles` flags mean; ``flags[:, proj_tile_map(H, W, blocks)]`` is the per-texel mask."""
    if blocks and H % 4 == 0 and W % 16 == 0:
        y, x = torch.arange(H).view(-1, 1), torch.arange(W).view(1, -1)
        return (y // 4) * (W // 16) + (x // 16)
    return (torch.arange(H * W) // 64).view(H, W)


def proj_tile_count(spec: FieldSpec, H: int, W: int) -> int:
    """Tiles (64 texels each: proj_tile_map) per image of an (H, W) projected map (bts_proj_tile_count)."""
    cfg = _spec_cfg(spec, 1, H, W)
    n = int(_lib.load().bts_proj_tile_count(C.byref(cfg)))
    if n < 0:
        raise BtsNativeError(f"bts_proj_tile_count: {_lib.load().bts_last_error().decode(errors='replace')}")
    return n


class FieldTensors:
    """Device tensors in the layouts of include/bts_render.h.  ``proj_nhwc`` (G) may require grad: it is produced by
    ``ProjectFunction`` from the encoder output and the MLP parameters inside autograd.
    ``feat_shift`` = s: the map is the decoder's scale-s output at ITS size (n, H >> s, W >> s, .) and is read as its nearest-neighbour
    resize to H x W -- what models_bts.py:115-117 materialises -- without the resize (BtsFieldCfg.feat_shift)."""

    def __init__(self, spec: FieldSpec, proj_nhwc, K_enc, w2c_enc, imgs_nhwc4, K_r, w2c_r, empty_feature=None, feat_nhwc=None, feat_shift=0,
                 enc_view=-1):
        """enc_view: index of the render view whose camera is the encoder camera (BtsFieldCfg.enc_render_view), -1 = none / unknown."""
        ref = proj_nhwc if proj_nhwc is not None else feat_nhwc
        n, H, W, ch = ref.shape
        H, W = H << feat_shift, W << feat_shift
        if feat_shift and proj_nhwc is None:
            raise BtsNativeError("feat_shift > 0 needs the projected map")
        if proj_nhwc is not None and ch != spec.d_hidden:
            raise BtsNativeError(f"proj_nhwc has {ch} channels, spec says d_hidden={spec.d_hidden}")
        if proj_nhwc is None and ch != spec.C:
            raise BtsNativeError(f"feat_nhwc has {ch} channels, spec says C={spec.C}")
        nv = 0 if imgs_nhwc4 is None else imgs_nhwc4.shape[1]
        _req(ref.detach(), "proj_nhwc/feat_nhwc"), _req(K_enc, "K_enc", (n, 3, 3)), _req(w2c_enc, "w2c_enc", (n, 4, 4))
        if nv:
            _req(imgs_nhwc4, "imgs_nhwc4", (n, nv, H, W, 4)), _req(K_r, "K_r", (n, nv, 3, 3)), _req(w2c_r, "w2c_r", (n, nv, 4, 4))
        if nv > _lib.BTS_MAX_VIEWS:
            raise BtsNativeError(f"nv={nv} render views exceed BTS_MAX_VIEWS={_lib.BTS_MAX_VIEWS}")
        if spec.learn_empty:
            if empty_feature is None:
                raise BtsNativeError("learn_empty needs empty_feature")
            _req(empty_feature.detach(), "empty_feature", (spec.C,))
        self.spec, self.n, self.H, self.W, self.nv, self.feat_shift = spec, n, H, W, nv, feat_shift
        self.enc_view = enc_view
        self.proj_nhwc, self.feat_nhwc, self.K_enc, self.w2c_enc = proj_nhwc, feat_nhwc, K_enc, w2c_enc
        self.imgs_nhwc4, self.K_r, self.w2c_r = imgs_nhwc4, K_r, w2c_r
        self.empty_feature = empty_feature
        self.proj_link = None   # ProjLink when proj_nhwc came out of ProjectFunction (field.py): see SPARSE_PROJ_GRAD
        # (rays data_ptr, z / jitter data_ptr, rows) of the ONE sample set a sparsely projected map is valid for (BTSNet.native_field(
        # sampled=...): only the tiles those samples read were projected, the rest of proj_nhwc is uninitialised memory), else None.
        # field_query / occupancy_profile reject a partial map, render_fwd / render_bwd check the sample set.
        self.partial = None

    def cfg(self, nv=None) -> BtsFieldCfg:
        return _spec_cfg(self.spec, self.n, self.H, self.W, self.nv if nv is None else nv, self.feat_shift, self.enc_view)

    def tensors(self, mlp_params: torch.Tensor) -> BtsFieldTensors:
        def dp(t):
            return None if t is None else t.data_ptr()
        return BtsFieldTensors(feat_nhwc=dp(self.feat_nhwc), proj_nhwc=dp(self.proj_nhwc), K_enc=dp(self.K_enc), w2c_enc=dp(self.w2c_enc),
                               imgs_nhwc4=dp(self.imgs_nhwc4), K_r=dp(self.K_r), w2c_r=dp(self.w2c_r),
                               empty_feature=dp(self.empty_feature), mlp_params=mlp_params.data_ptr())


def check_supported(spec: FieldSpec, nv: int = 1):
    cfg = _spec_cfg(spec, nv=nv)
    if not _lib.load().bts_supported(C.byref(cfg)):
        raise BtsNativeError(f"field shape outside the compiled envelope: C={spec.C} d_hidden={spec.d_hidden} "
                             f"n_blocks={spec.n_blocks} num_freqs={spec.num_freqs} nv={nv}")


# --------------------------------------------------------------------------------------------------------------
# render forward / backward
# --------------------------------------------------------------------------------------------------------------
_OUT_KEYS = ("rgb", "depth", "weights", "alphas", "invalid", "rgb_samps", "sigma_raw", "trans", "invalid_wsum", "invalid_any")


def _render_args(ft: FieldTensors, rays, z_samp, hard_alpha_cap, white_bkgd, outs, sigma_noise=None, jitter=None, lindisp=True):
    K = (z_samp if z_samp is not None else jitter).shape[1]
    if sigma_noise is not None:
        _req(sigma_noise, "sigma_noise", (rays.shape[0], K))
    return BtsRenderArgs(rays_per_sample=rays.shape[0] // ft.n, K=K, hard_alpha_cap=int(hard_alpha_cap),
                         white_bkgd=int(white_bkgd), rays=rays.data_ptr(), z_samp=None if z_samp is None else z_samp.data_ptr(),
                         sigma_noise=None if sigma_noise is None else sigma_noise.data_ptr(),
                         jitter=None if jitter is None else jitter.data_ptr(), lindisp=int(bool(lindisp)), reserved_=0,
                         z_samp_out=None if outs.get("z_samp") is None else outs["z_samp"].data_ptr(),
                         **{k: (None if outs.get(k) is None else outs[k].data_ptr()) for k in _OUT_KEYS})


def render_fwd(ft: FieldTensors, mlp_params: torch.Tensor, rays: torch.Tensor, z_samp: torch.Tensor, *, hard_alpha_cap: bool,
               white_bkgd: bool = False, want_weights=False, want_alphas=False, want_invalid=True, want_rgb_samps=False,
               want_saved=False, want_invalid_sums=False, sigma_noise=None, jitter=None, lindisp=True, want_z=False):
    """rays (n*Bp, 8), z_samp (n*Bp, K) -> dict of fresh tensors (bts_render_fwd).  want_saved adds the two per-sample
    activations the backward needs (sigma_raw, trans); want_invalid_sums the per-ray reductions the loss' invalid-ray policies need
    (invalid_wsum = sum_k weights * invalid, invalid_any = max_k invalid, (n*Bp, nv) each) -- with them a training step can leave
    weights / invalid / rgb_samps unrequested.
    z_samp=None with ``jitter`` (n*Bp, K) in [0, 1): NeRFRenderer.sample_coarse runs inside the kernel (BtsRenderArgs.jitter, ``lindisp``
    as the renderer's flag); the depths come back as out["z_samp"] when ``want_z`` (bit-identical to ``sample_coarse(rays, jitter)``)."""
    if z_samp is None:
        if jitter is None:
            raise BtsNativeError("render_fwd needs z_samp or jitter")
        if ft.proj_nhwc is None:
            raise BtsNativeError("in-kernel sampling (jitter) needs the projected feature map")
        _req(jitter, "jitter")
    else:
        _req(z_samp, "z_samp")
        jitter = None
    B, K = (z_samp if z_samp is not None else jitter).shape
    _req(rays, "rays", (B, 8)), _req(mlp_params.detach(), "mlp_params", (ft.spec.mlp_param_count(),))
    if B % ft.n != 0:
        raise BtsNativeError(f"{B} rays do not split evenly over n={ft.n} samples")
    _check_partial(ft, rays, z_samp if z_samp is not None else jitter)
    dev, nv = rays.device, ft.nv

    def new(*shape):
        return torch.empty(shape, device=dev, dtype=torch.float32)

    outs = dict(rgb=new(B, nv * 3), depth=new(B), weights=new(B, K) if want_weights else None,
                alphas=new(B, K) if want_alphas else None, invalid=new(B, K, nv) if want_invalid else None,
                rgb_samps=new(B, K, nv * 3) if want_rgb_samps else None, sigma_raw=new(B, K) if want_saved else None,
                trans=new(B, K) if want_saved else None, invalid_wsum=new(B, nv) if want_invalid_sums else None,
                invalid_any=new(B, nv) if want_invalid_sums else None,
                z_samp=new(B, K) if (z_samp is None and want_z) else None)
    cfg, tens = ft.cfg(), ft.tensors(mlp_params)
    args = _render_args(ft, rays, z_samp, hard_alpha_cap, white_bkgd, outs, sigma_noise, jitter, lindisp)
    _lib.check(_lib.load().bts_render_fwd(C.byref(cfg), C.byref(tens), C.byref(args), _stream(rays)), "bts_render_fwd")
    if z_samp is not None:
        outs["z_samp"] = z_samp
    return outs


def _check_partial(ft: "FieldTensors", rays, samples, what="render"):
    """A map projected for one sample set (FieldTensors.partial) serves that set and nothing else."""
    if ft.partial is None:
        return
    if samples is None or rays is None:
        raise BtsNativeError(f"{what}: this field's projected map holds only the tiles ONE render's samples read (native_field(sampled=...)); "
                             "field queries need the full map: call net.native_field() without `sampled`")
    if (rays.data_ptr(), rays.shape[0]) != ft.partial[:2]:
        raise BtsNativeError(f"{what}: the projected map of this field was built for another ray set (sparse projection, native_field(sampled=...))")


def render_bwd(ft: FieldTensors, mlp_params, rays, z_samp, sigma_raw, trans, *, hard_alpha_cap, g_rgb=None, g_depth=None,
               g_weights=None, g_alphas=None, need_proj=True, need_mlp=True, need_empty=False, white_bkgd=False, rgb_samps=None,
               sigma_noise=None, proj_grad=None):
    """Returns (d_proj_nhwc | None, d_mlp_params | None, d_empty_proj | None) (bts_render_bwd).  ``proj_grad`` = (d_proj, tiles): add
    into THIS d_proj (the caller vouches that it is zero wherever ``tiles`` is) and flag the 64-texel tiles that received something
    (BtsRenderGrads.d_proj_tiles) -- no fill of a map-sized tensor."""
    B, K = z_samp.shape
    for name, g in (("g_rgb", g_rgb), ("g_depth", g_depth), ("g_weights", g_weights), ("g_alphas", g_alphas)):
        if g is not None:
            _req(g, name)
    _req(sigma_raw, "sigma_raw", (B, K)), _req(trans, "trans", (B, K))
    dev = rays.device
    tiles = None
    if need_proj and proj_grad is not None:
        d_proj, tiles = proj_grad
        _req(d_proj, "proj_grad[0]", tuple(ft.proj_nhwc.shape))
    else:
        d_proj = torch.zeros(ft.proj_nhwc.shape, device=dev, dtype=torch.float32) if need_proj else None
    d_mlp = torch.zeros(ft.spec.mlp_param_count(), device=dev, dtype=torch.float32) if need_mlp else None
    d_empty = torch.zeros(ft.spec.d_hidden, device=dev, dtype=torch.float32) if need_empty else None
    cfg, tens = ft.cfg(), ft.tensors(mlp_params)
    if rgb_samps is not None:
        _req(rgb_samps, "rgb_samps", (B, K, ft.nv * 3))
    args = _render_args(ft, rays, z_samp, hard_alpha_cap, white_bkgd, dict(sigma_raw=sigma_raw, trans=trans, rgb_samps=rgb_samps), sigma_noise)

    def dp(t):
        return None if t is None else t.data_ptr()
    grads = BtsRenderGrads(g_rgb=dp(g_rgb), g_depth=dp(g_depth), g_weights=dp(g_weights), g_alphas=dp(g_alphas),
                           d_proj_nhwc=dp(d_proj), d_mlp_params=dp(d_mlp), d_empty_proj=dp(d_empty), d_proj_tiles=dp(tiles))
    lib = _lib.load()
    ws_bytes = lib.bts_render_bwd_workspace(C.byref(cfg), C.byref(args))
    ws = _workspace(dev, int(ws_bytes))
    _lib.check(lib.bts_render_bwd(C.byref(cfg), C.byref(tens), C.byref(args), C.byref(grads), _ptr(ws), ws_bytes, _stream(rays)),
               "bts_render_bwd")
    return d_proj, d_mlp, d_empty


_WS = {}
WORKSPACE_CACHE_LIMIT = 256 << 20   # bytes: larger scratch is allocated per call and goes back to torch's caching allocator


def _workspace(dev, n_bytes):
    """Scratch of the backward (what its passes hand each other: 20 B per sample on the bit path, 4 + 4 d_hidden B per sample on the row
    path).  Buffers up to WORKSPACE_CACHE_LIMIT are kept per device and stream, grown on demand and reused -- the contents need no
    initialisation and every use is ordered on the stream; anything larger is a plain torch allocation of this call, so that an
    occasional huge backward does not pin its scratch for the life of the process (release_workspace() drops the kept ones too)."""
    if n_bytes > WORKSPACE_CACHE_LIMIT:
        return torch.empty(n_bytes // 4 + 1, device=dev, dtype=torch.float32)
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() * 4 < n_bytes + 4:
        buf = torch.empty(n_bytes // 4 + 1, device=dev, dtype=torch.float32)
        _WS[key] = buf
    return buf


def release_workspace():
    """Drops the cached backward scratch (e.g. when a process switches from training to evaluation)."""
    _WS.clear()


def field_query(ft: FieldTensors, mlp_params: torch.Tensor, xyz: torch.Tensor, only_density: bool = False):
    """xyz (n, P, 3) -> rgb (n,P,nv*3) | None, invalid (n,P,nv or 1), sigma (n,P,1)  (bts_field_query)."""
    n, P, _ = xyz.shape
    _req(xyz, "xyz", (ft.n, P, 3))
    _check_partial(ft, None, None, "field_query")
    dev = xyz.device
    nv = 0 if only_density else ft.nv
    rgb = torch.empty((n, P, nv * 3), device=dev, dtype=torch.float32) if nv else None
    invalid = torch.empty((n, P, max(nv, 1)), device=dev, dtype=torch.float32)
    sigma = torch.empty((n, P, 1), device=dev, dtype=torch.float32)
    if nv == 0 and not only_density:  # no colour views: invalid is the feature-frustum test only
        only_density = True
    cfg, tens = ft.cfg(), ft.tensors(mlp_params)
    _lib.check(_lib.load().bts_field_query(C.byref(cfg), C.byref(tens), _ptr(xyz), P, int(only_density), _ptr(rgb), _ptr(invalid),
                                           _ptr(sigma), _stream(xyz)), "bts_field_query")
    return rgb, invalid, sigma


def occupancy_profile(ft: FieldTensors, mlp_params: torch.Tensor, xyz: torch.Tensor, levels: int, threshold: float = 8.0,
                      only_density: bool = False, want_sigma: bool = False):
    """xyz (n, levels * columns, 3): a dense grid of query points, the vertical level slowest -> profile (n, columns)
    [, sigma (n, levels * columns)]  (bts_occupancy_profile: render_profile of scripts/inference_setup.py in one pass)."""
    n, P, _ = xyz.shape
    _req(xyz, "xyz", (ft.n, P, 3))
    if levels <= 0 or P % levels:
        raise BtsNativeError(f"{P} points are not {levels} whole levels")
    _check_partial(ft, None, None, "occupancy_profile")
    cols = P // levels
    profile = torch.empty((n, cols), device=xyz.device, dtype=torch.float32)
    sigma = torch.empty((n, P), device=xyz.device, dtype=torch.float32) if want_sigma else None
    cfg, tens = ft.cfg(nv=0 if only_density else None), ft.tensors(mlp_params)
    _lib.check(_lib.load().bts_occupancy_profile(C.byref(cfg), C.byref(tens), _ptr(xyz), levels, cols, float(threshold), int(only_density),
                                                 _ptr(profile), _ptr(sigma), _stream(xyz)), "bts_occupancy_profile")
    return (profile, sigma) if want_sigma else profile


# --------------------------------------------------------------------------------------------------------------
# the Monodepth2 decoder's tail: reflect-pad 3x3 convolution [+ nearest x2 in front] [+ ELU], channels-last
# --------------------------------------------------------------------------------------------------------------
def _conv_struct(x, weight, bias, y, N, H, W, up2, elu, out_nchw):
    return _lib.BtsConv3x3(N=N, H=H, W=W, C=x.shape[-1], up2=int(up2), elu=int(elu), out_nchw=int(out_nchw), reserved_=0, x=x.data_ptr(),
                           weight=weight.data_ptr(), bias=None if bias is None else bias.data_ptr(), y=None if y is None else y.data_ptr())


def conv3x3_fwd(x, weight, bias, up2=False, elu=False, out_nchw=False):
    """x (N, Hs, Ws, C) channels-last, weight (C, C, 3, 3), bias (C) | None -> y (N, H, W, C), or (N, C, H, W) with ``out_nchw``; H, W = 2 Hs,
    2 Ws with ``up2`` (bts_conv3x3_fwd: reflection pad 1 + 3 x 3 convolution, the x2 nearest upsampling in front and the ELU behind fused)."""
    _req(x, "x"), _req(weight, "weight", (x.shape[-1], x.shape[-1], 3, 3))
    if bias is not None:
        _req(bias, "bias", (x.shape[-1],))
    N, Hs, Ws, Cc = x.shape
    H, W = (2 * Hs, 2 * Ws) if up2 else (Hs, Ws)
    y = torch.empty((N, Cc, H, W) if out_nchw else (N, H, W, Cc), device=x.device, dtype=torch.float32)
    c = _conv_struct(x, weight, bias, y, N, H, W, up2, elu, out_nchw)
    _lib.check(_lib.load().bts_conv3x3_fwd(C.byref(c), _stream(x)), "bts_conv3x3_fwd")
    return y


def conv3x3_bwd(x, weight, y, g_y, up2=False, elu=False, out_nchw=False, need=(True, True, True)):
    """-> (d_x | None, d_weight | None, d_bias | None) of conv3x3_fwd (bts_conv3x3_bwd); ``y`` = the forward's output (read for elu')."""
    _req(x, "x"), _req(weight, "weight"), _req(g_y, "g_y", tuple(y.shape))
    N, Hs, Ws, Cc = x.shape
    H, W = (2 * Hs, 2 * Ws) if up2 else (Hs, Ws)
    d_x = torch.empty_like(x) if need[0] else None
    d_w = torch.empty_like(weight) if need[1] else None
    d_b = torch.empty(Cc, device=x.device, dtype=torch.float32) if need[2] else None
    c = _conv_struct(x, weight, None, y, N, H, W, up2, elu, out_nchw)
    lib = _lib.load()
    ws_bytes = int(lib.bts_conv3x3_bwd_workspace(C.byref(c)))
    ws = torch.empty(ws_bytes // 4 + 4, device=x.device, dtype=torch.float32)     # (map-sized: goes back to the caching allocator)
    _lib.check(lib.bts_conv3x3_bwd(C.byref(c), _ptr(g_y), _ptr(ws), ws_bytes, _ptr(d_x), _ptr(d_w), _ptr(d_b), _stream(x)), "bts_conv3x3_bwd")
    return d_x, d_w, d_b


class Conv3x3Function(torch.autograd.Function):
    """layers.py:11-40 (Conv3x3 / ConvBlock) [+ the decoder's nearest x2, monodepth2.py:225] as one differentiable op on channels-last
    tensors: (x (N, Hs, Ws, C), weight, bias | None, up2, elu, out_nchw) -> y."""

    @staticmethod
    def forward(ctx, x, weight, bias, up2, elu, out_nchw):
        x, weight = x.contiguous(), weight.contiguous()
        y = conv3x3_fwd(x, weight, None if bias is None else bias.contiguous(), up2, elu, out_nchw)
        ctx.flags = (bool(up2), bool(elu), bool(out_nchw), bias is not None)
        ctx.save_for_backward(x, weight, y if elu else x.new_empty(0))
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, y = ctx.saved_tensors
        up2, elu, out_nchw, has_bias = ctx.flags
        if not elu:      # (a plain convolution's backward does not read its output: only the shape matters)
            N, Hs, Ws, Cc = x.shape
            H, W = (2 * Hs, 2 * Ws) if up2 else (Hs, Ws)
            y = g
        need = (ctx.needs_input_grad[0], ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[2])
        d_x, d_w, d_b = conv3x3_bwd(x, weight, y, g.contiguous(), up2, elu, out_nchw, need)
        return d_x, d_w, d_b, None, None, None


def eval_frame(fr, stream):
    """bts_eval_frame on a filled ``_lib.BtsEvalFrame`` (behindthescenes_amd.train_step.FusedEvalFrame builds it)."""
    _lib.check(_lib.load().bts_eval_frame(C.byref(fr), stream), "bts_eval_frame")


def train_step_fwd(st, stream):
    """bts_train_step_fwd on a filled ``_lib.BtsTrainStep`` (behindthescenes_amd.train_step builds it)."""
    _lib.check(_lib.load().bts_train_step_fwd(C.byref(st), stream), "bts_train_step_fwd")


def train_step_bwd(st, g_loss, stream):
    """bts_train_step_bwd: ``g_loss`` a 0-dim float32 device tensor (the upstream gradient of the loss) or None (= 1)."""
    _lib.check(_lib.load().bts_train_step_bwd(C.byref(st), None if g_loss is None else C.c_void_p(g_loss.data_ptr()), stream), "bts_train_step_bwd")


# --------------------------------------------------------------------------------------------------------------
# autograd glue
# --------------------------------------------------------------------------------------------------------------
# The gradient of a projected map G is SPARSE in a training step: the rays of a few thousand 8 x 8 patches reach 8-15 % of its texels,
# yet as a dense autograd tensor it costs a map-sized zero fill, and a full read in the projection's backward (at exp_kitti_360.yaml's
# batch 503 MB each -- as much as everything else the backward moves).  When exactly ONE render reads a map that ProjectFunction made,
# RenderFunction.backward therefore adds into a kept (d_proj, tile flags) pair that is all zero between steps, and
# ProjectFunction.backward reads the flagged tiles only and returns the pair to zero (bts_project_features_bwd_tiles).  Every other
# constellation -- several renders of one map, a hook or retain_grad on G, somebody touching the gradient on its way -- takes the dense
# path; the results are the same either way (tests/test_gpu_sparse_grad.py).
SPARSE_PROJ_GRAD = True


class ProjLink:
    """Ties a map made by ProjectFunction to the RenderFunction calls that read it (FieldTensors.proj_link)."""
    __slots__ = ("renders", "entry", "__weakref__")

    def __init__(self):
        self.renders, self.entry = 0, None


class _SparseGrad:
    __slots__ = ("buf", "tiles", "busy", "version", "owner")


def _reclaim_stale(entry):
    """A busy pair whose map is gone -- the projection's backward never ran for it (torch.autograd.grad(loss, [G]), `inputs=` without the
    encoder, an interrupted backward): the ProjLink it was lent to died with its graph.  The pair is zeroed and free again; without this
    every later step would silently take the dense path and the map-sized buffer would stay pinned half written."""
    if entry.busy and (entry.owner is None or entry.owner() is None or entry.owner().entry is not entry):
        entry.buf.zero_(), entry.tiles.zero_()
        entry.busy = False


_SPARSE = {}


def _sparse_grad(ft: "FieldTensors"):
    g = ft.proj_nhwc
    key = (g.device, tuple(g.shape), torch.cuda.current_stream(g.device).cuda_stream)
    e = _SPARSE.get(key)
    if e is None:
        e = _SparseGrad()
        e.buf = torch.zeros(g.shape, device=g.device, dtype=torch.float32)
        e.tiles = torch.zeros((g.shape[0], proj_tile_count(ft.spec, g.shape[1], g.shape[2])), device=g.device, dtype=torch.uint8)
        e.busy, e.owner = False, None
        _SPARSE[key] = e
    return e


def release_sparse_grads():
    """Drops the kept (d_proj, tile flags) pairs (one map-sized buffer per map shape, device and stream)."""
    _SPARSE.clear()


class ProjectFunction(torch.autograd.Function):
    """(F nchw, packed mlp params) -> G nhwc.  Backward: per-pixel GEMMs in bts_project_features_bwd (_tiles)."""

    @staticmethod
    def forward(ctx, feat_nchw, mlp_params, spec, link=None, tiles=None):
        feat_nchw = as_feature_map(feat_nchw)
        ctx.set_materialize_grads(False)
        ctx.spec, ctx.link = spec, link
        ctx.save_for_backward(feat_nchw, mlp_params)
        return project_features(spec, feat_nchw, mlp_params.contiguous(), tiles)

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None, None, None
        feat_nchw, mlp_params = ctx.saved_tensors
        need = ctx.needs_input_grad[:2]
        entry = ctx.link.entry if ctx.link is not None else None
        if entry is not None:
            ctx.link.entry = None
            if g.data_ptr() == entry.buf.data_ptr() and entry.buf._version == entry.version and g.shape == entry.buf.shape:
                d_feat, d_mlp = project_features_bwd(ctx.spec, feat_nchw, entry.buf, mlp_params, *need, tiles=entry.tiles, clear_after=True)
                entry.busy = False
                return d_feat, d_mlp, None, None, None
        d_feat, d_mlp = project_features_bwd(ctx.spec, feat_nchw, g.contiguous(), mlp_params, *need)
        if entry is not None:   # the kept pair was replaced or modified on its way here: dense on what arrived, and the pair starts over
            entry.buf.zero_(), entry.tiles.zero_()
            entry.busy = False
        return d_feat, d_mlp, None, None, None


class RenderFunction(torch.autograd.Function):
    """composite() as one differentiable op.  Differentiable inputs: proj_nhwc (G), mlp_params, empty_feature.
    Differentiable outputs: rgb, depth, weights, alphas (invalid / rgb_samps carry no gradient, as in the reference where
    they only depend on poses and colours)."""

    @staticmethod
    def forward(ctx, proj_nhwc, mlp_params, empty_feature, ft: FieldTensors, rays, z_samp, hard_alpha_cap, white_bkgd,
                want_weights, want_alphas, want_rgb_samps, grad_mode=True, want_invalid=True, want_invalid_sums=False, sigma_noise=None,
                jitter=None, lindisp=True, want_z=False):
        # needs_input_grad reflects requires_grad even under torch.no_grad(), and inside forward() grad mode is always off: the
        # caller passes the mode it was invoked in, so that evaluation does not allocate / write the 8 B per sample of saved state
        # (outputs nobody differentiates through arrive as None in backward, not as zero-filled tensors of their size: in a lean training
        # step that was four fills per render -- rgb_samps, z_samp, the two per-ray reductions)
        ctx.set_materialize_grads(False)
        needs_grad = any(ctx.needs_input_grad[:3]) and grad_mode
        # z_samp None: sample_coarse inside the kernel from `jitter`; the depths are materialised only for the backward / on request
        out = render_fwd(ft, mlp_params, rays, z_samp, hard_alpha_cap=hard_alpha_cap, white_bkgd=white_bkgd,
                         want_weights=want_weights, want_alphas=want_alphas, want_invalid=want_invalid, want_rgb_samps=want_rgb_samps,
                         want_saved=needs_grad, want_invalid_sums=want_invalid_sums, sigma_noise=sigma_noise, jitter=jitter, lindisp=lindisp,
                         want_z=want_z or needs_grad)
        ctx.ft, ctx.hard_alpha_cap, ctx.white_bkgd = ft, hard_alpha_cap, white_bkgd
        ctx.link = getattr(ft, "proj_link", None) if (needs_grad and ctx.needs_input_grad[0]) else None
        if ctx.link is not None:
            ctx.link.renders += 1
        ctx.sigma_noise = sigma_noise if needs_grad else None      # (a plain tensor without graph: kept on the context)
        if needs_grad:
            # rgb_samps is non-differentiable output the caller asked for: kept for the backward too (it then skips the colour taps)
            ctx.save_for_backward(mlp_params, rays, out["z_samp"], out["sigma_raw"], out["trans"], *([out["rgb_samps"]] if want_rgb_samps else []))
        empty = rays.new_empty(0)
        z_out = out["z_samp"] if (z_samp is None and out["z_samp"] is not None) else empty
        res = (out["rgb"], out["depth"], out["weights"] if want_weights else empty, out["alphas"] if want_alphas else empty,
               out["invalid"] if want_invalid else empty, out["rgb_samps"] if want_rgb_samps else empty,
               out["invalid_wsum"] if want_invalid_sums else empty, out["invalid_any"] if want_invalid_sums else empty, z_out)
        ctx.mark_non_differentiable(*res[4:])
        ctx.has = (want_weights, want_alphas)
        return res

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_weights, g_alphas, _g_inv, _g_rs, _g_iw, _g_ia, _g_z):
        mlp_params, rays, z_samp, sigma_raw, trans, *rest = ctx.saved_tensors
        rgb_samps = rest[0] if rest else None

        def prep(g, present=True):
            return g.contiguous() if (g is not None and present and g.numel() > 0) else None

        ft = ctx.ft
        need_proj, need_mlp, need_empty = ctx.needs_input_grad[:3]
        pg, link, G, entry = None, ctx.link, ft.proj_nhwc, None
        if (need_proj and SPARSE_PROJ_GRAD and link is not None and link.renders == 1 and link.entry is None and not G.retains_grad
                and not G._backward_hooks):
            entry = _sparse_grad(ft)
            _reclaim_stale(entry)
            if not entry.busy:
                entry.busy, entry.version, link.entry = True, entry.buf._version, entry
                entry.owner = weakref.ref(link)
                pg = (entry.buf, entry.tiles)
            else:
                entry = None
        try:
            d_proj, d_mlp, d_eproj = render_bwd(ft, mlp_params, rays, z_samp, sigma_raw, trans, hard_alpha_cap=ctx.hard_alpha_cap,
                                                g_rgb=prep(g_rgb), g_depth=prep(g_depth), g_weights=prep(g_weights, ctx.has[0]), white_bkgd=ctx.white_bkgd,
                                                rgb_samps=rgb_samps, g_alphas=prep(g_alphas, ctx.has[1]), need_proj=need_proj, need_mlp=need_mlp,
                                                need_empty=need_empty or (need_mlp and ft.spec.learn_empty), sigma_noise=ctx.sigma_noise, proj_grad=pg)
        except Exception:
            if pg is not None:      # a failed launch may have left the kept pair half written: it starts over, and the link forgets it
                entry.buf.zero_(), entry.tiles.zero_()
                entry.busy, link.entry = False, None
            raise
        d_empty = None
        if d_eproj is not None:
            # the projected empty feature is w_in[:, :C] @ empty_feature (a 64x64 GEMV): chain rule on parameter-sized tensors
            spec = ft.spec
            w_f = mlp_params[:spec.d_hidden * spec.d_in].view(spec.d_hidden, spec.d_in)[:, :spec.C]
            if need_empty:
                d_empty = w_f.t() @ d_eproj
            if need_mlp:
                d_mlp[:spec.d_hidden * spec.d_in].view(spec.d_hidden, spec.d_in)[:, :spec.C] += torch.outer(d_eproj, ft.empty_feature.detach())
        return (d_proj, d_mlp, d_empty) + (None,) * 15
