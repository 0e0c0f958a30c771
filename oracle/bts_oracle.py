"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the BehindTheScenes density-field
rendering hot path.  Not shipped, never imported by ``behindthescenes_amd``; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may use it (as the checker /
the timed CPU port, never as the product).

It is a from-scratch functional restatement in fp32 torch-CPU ops of what the reference computes
(citations are ``file:line`` into the upstream tree at tag 2024_10_08).  The floating-point primitives
(`grid_sample`, `addmm`, `softplus`, `cumprod`, `sin`) are the very torch ops the reference calls, so the
oracle tracks the reference's rounding behaviour as closely as a CPU restatement can.

Pinning: the reference ships NO tests / golden vectors for this path, so the
oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF: ``tests/golden/gen_golden.py`` imports the real
reference through ``oracle/ref_shim.py`` in the build container and commits input/output fixtures under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this file against those fixtures (and, when the
reference tree is present, against the live reference).

Conventions (all tensors fp32):
  rays      (B, 8)   [origin(3), direction(3), near, far]          nerf.py:106, util.py:270-273
  z_samp    (B, K)   sample depths along each ray                   nerf.py:210-218
  feat      (n, C, H, W)   pixel-aligned feature map of the ONE encoder view (ids_encoder=[0])
  K_enc     (n, 3, 3), w2c_enc (n, 4, 4)                             models_bts.py:128-131
  imgs      (n, nv, 3, H, W) colours in [0,1]; K_r (n,nv,3,3); w2c_r (n,nv,4,4)   models_bts.py:133-136
"""
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import math
import torch
import torch.nn.functional as F

EPS = 1e-3  # models_bts.py:14


# ----------------------------------------------------------------------------------------------
# configuration / parameter containers
# ----------------------------------------------------------------------------------------------
@dataclass
class FieldConfig:
    """Scalar configuration of the field (models_bts.py:18-54) and the PE (code.py:11-28)."""
    d_min: float = 3.0            # z_near
    d_max: float = 80.0           # z_far
    inv_z: bool = True
    code_mode: str = "z"          # "z" | "distance"
    num_freqs: int = 6
    freq_factor: float = 1.5
    include_input: bool = True
    learn_empty: bool = False
    empty_empty: bool = False


@dataclass
class MlpParams:
    """ResnetFC parameters (resnetfc.py:65-130): lin_in, n_blocks x (fc_0, fc_1), lin_out."""
    w_in: torch.Tensor            # (Hd, d_in)
    b_in: torch.Tensor            # (Hd,)
    blocks: List[Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]] = field(default_factory=list)
    w_out: torch.Tensor = None    # (d_out, Hd)
    b_out: torch.Tensor = None    # (d_out,)

    def tensors(self):
        out = [self.w_in, self.b_in]
        for blk in self.blocks:
            out.extend(blk)
        out.extend([self.w_out, self.b_out])
        return out


@dataclass
class FieldState:
    """What ``BTSNet.encode`` leaves behind for the renderer (models_bts.py:128-136)."""
    feat: torch.Tensor            # (n, C, H, W)
    K_enc: torch.Tensor           # (n, 3, 3)
    w2c_enc: torch.Tensor         # (n, 4, 4)
    imgs: torch.Tensor            # (n, nv, 3, H, W)
    K_r: torch.Tensor             # (n, nv, 3, 3)
    w2c_r: torch.Tensor           # (n, nv, 4, 4)
    empty_feature: Optional[torch.Tensor] = None   # (C,), only with learn_empty


def init_mlp(d_in: int, d_hidden: int, n_blocks: int, d_out: int = 1, gen: Optional[torch.Generator] = None,
             out_std: Optional[float] = 0.3) -> MlpParams:
    """Reference initialisation (kaiming-normal fan_in, zero bias: resnetfc.py:36-39, 88-94) and then, as
    SURVEY.md section 8d prescribes for synthetic parity inputs, lin_out.weight ~ N(0, out_std) and fc_1 non-zero
    so that sigma is not constant."""
    def kaiming(o, i):
        return torch.randn(o, i, generator=gen) * math.sqrt(2.0 / i)

    p = MlpParams(w_in=kaiming(d_hidden, d_in), b_in=torch.zeros(d_hidden))
    for _ in range(n_blocks):
        w0 = kaiming(d_hidden, d_hidden)
        w1 = torch.zeros(d_hidden, d_hidden) if out_std is None else kaiming(d_hidden, d_hidden) * 0.5
        p.blocks.append((w0, torch.zeros(d_hidden), w1, torch.zeros(d_hidden)))
    p.w_out = kaiming(d_out, d_hidden) if out_std is None else torch.randn(d_out, d_hidden, generator=gen) * out_std
    p.b_out = torch.zeros(d_out)
    return p


# ----------------------------------------------------------------------------------------------
# a1: ray generation  (util.py:113-149 unproj_map, util.py:244-273 gen_rays)
# ----------------------------------------------------------------------------------------------
def pixel_directions(width: int, height: int, focal: torch.Tensor, center: torch.Tensor, norm_dir: bool = True):
    """(v, H, W, 3) camera-frame directions.  Pixel grid is linspace(-1, 1, W) x linspace(-1, 1, H)
    (pixel *corners* convention, util.py:140-141); d = ((x - cx)/fx, (y - cy)/fy, 1), optionally normalised."""
    v = focal.shape[0]
    gx = torch.linspace(-1, 1, width, dtype=torch.float32).view(1, 1, width).expand(v, height, width)
    gy = torch.linspace(-1, 1, height, dtype=torch.float32).view(1, height, 1).expand(v, height, width)
    g = torch.stack((gx, gy), dim=-1)
    g = (g - center.view(v, 1, 1, 2)) / focal.view(v, 1, 1, 2)
    d = torch.cat((g, torch.ones_like(gx).unsqueeze(-1)), dim=-1)
    if norm_dir:
        d = d / torch.norm(d, dim=-1).unsqueeze(-1)
    return d


def gen_rays(poses_c2w: torch.Tensor, width: int, height: int, z_near: float, z_far: float,
             focal: torch.Tensor, center: torch.Tensor, norm_dir: bool = True) -> torch.Tensor:
    """(v, H, W, 8) rays: origin = c2w translation, direction = R @ d_cam, near, far (util.py:244-273)."""
    v = poses_c2w.shape[0]
    d_cam = pixel_directions(width, height, focal, center, norm_dir)
    origins = poses_c2w[:, None, None, :3, 3].expand(-1, height, width, -1)
    d_world = torch.matmul(poses_c2w[:, None, None, :3, :3], d_cam.unsqueeze(-1))[..., 0]
    near = torch.full((v, height, width, 1), float(z_near), dtype=torch.float32)
    far = torch.full((v, height, width, 1), float(z_far), dtype=torch.float32)
    return torch.cat((origins, d_world, near, far), dim=-1)


def image_rays(poses_c2w: torch.Tensor, projs: torch.Tensor, height: int, width: int, z_near: float, z_far: float,
               norm_dir: bool = True) -> torch.Tensor:
    """ImageRaySampler.sample for the rays only (ray_sampler.py:233-260): (n, v*H*W, 8)."""
    n = poses_c2w.shape[0]
    out = []
    for i in range(n):
        focal = projs[i, :, [0, 1], [0, 1]]
        center = projs[i, :, [0, 1], [2, 2]]
        out.append(gen_rays(poses_c2w[i], width, height, z_near, z_far, focal, center, norm_dir).reshape(-1, 8))
    return torch.stack(out)


# ----------------------------------------------------------------------------------------------
# a4: stratified depth samples  (nerf.py:103-123)
# ----------------------------------------------------------------------------------------------
def sample_coarse(rays: torch.Tensor, n_coarse: int, lindisp: bool, u: torch.Tensor) -> torch.Tensor:
    """z = lerp in depth or in disparity of s_k = k/K + u_k/K.  ``u`` (B, K) in [0,1) replaces the
    reference's in-place ``rand_like`` (nerf.py:116) so that the jitter can be injected."""
    near, far = rays[:, 6:7], rays[:, 7:8]
    step = 1.0 / n_coarse
    s = torch.linspace(0, 1 - step, n_coarse, dtype=torch.float32).to(rays.device).unsqueeze(0).repeat(rays.shape[0], 1)
    s = s + u * step
    if not lindisp:
        return near * (1 - s) + far * s
    return 1 / (1 / near * (1 - s) + 1 / far * s)


# ----------------------------------------------------------------------------------------------
# a9: positional encoding (code.py:11-42)
# ----------------------------------------------------------------------------------------------
def positional_encoding(x: torch.Tensor, num_freqs: int = 6, freq_factor: float = 1.5, include_input: bool = True):
    """x (N, 3) -> (N, 3 + 6*num_freqs).  Output order: [x, then per octave k: sin(f_k x)(3), sin(f_k x + pi/2)(3)],
    where the "cos" is evaluated as sin(phase + x*f) with phase = fl32(pi/2) through addcmul (code.py:25-28, 38)."""
    freqs = (freq_factor * 2.0 ** torch.arange(0, num_freqs)).to(x.device)
    f = torch.repeat_interleave(freqs, 2).view(1, -1, 1).to(torch.float32)
    ph = torch.zeros(2 * num_freqs, device=x.device)
    ph[1::2] = math.pi * 0.5
    ph = ph.view(1, -1, 1)
    e = x.unsqueeze(1).repeat(1, 2 * num_freqs, 1)
    e = torch.sin(torch.addcmul(ph, e, f)).reshape(x.shape[0], -1)
    return torch.cat((x, e), dim=-1) if include_input else e


# ----------------------------------------------------------------------------------------------
# projection into a view (models_bts.py:144-155 and 220-231)
# ----------------------------------------------------------------------------------------------
def project(xyz: torch.Tensor, w2c: torch.Tensor, Ks: torch.Tensor):
    """xyz (n, P, 3); w2c (n, nv, 4, 4); Ks (n, nv, 3, 3) ->
    xy (n, nv, P, 2) in [-1,1] image coords, z (n, nv, P, 1), distance (n, nv, P, 1), invalid (n, nv, P, 1) bool."""
    n, P, _ = xyz.shape
    hom = torch.cat((xyz, torch.ones_like(xyz[..., :1])), dim=-1).unsqueeze(1)          # (n,1,P,4)
    cam = w2c[:, :, :3, :] @ hom.permute(0, 1, 3, 2)                                    # (n,nv,3,P)
    dist = torch.norm(cam, dim=-2).unsqueeze(-1)                                        # (n,nv,P,1)
    q = (Ks @ cam).permute(0, 1, 3, 2)                                                  # (n,nv,P,3)
    z = q[..., 2:3]
    xy = q[..., 0:2] / z.clamp_min(EPS)
    invalid = (z <= EPS) | (xy[..., :1] < -1) | (xy[..., :1] > 1) | (xy[..., 1:2] < -1) | (xy[..., 1:2] > 1)
    return xy, z, dist, invalid


def _bilinear_border(img: torch.Tensor, xy: torch.Tensor) -> torch.Tensor:
    """img (m, c, H, W), xy (m, P, 2) -> (m, P, c): F.grid_sample bilinear / border / align_corners=False
    (models_bts.py:179, 234)."""
    m, c = img.shape[:2]
    out = F.grid_sample(img, xy.view(m, 1, -1, 2), mode="bilinear", padding_mode="border", align_corners=False)
    return out.view(m, c, -1).permute(0, 2, 1)


# ----------------------------------------------------------------------------------------------
# a8: MLP input = [bilinear(F), PE(x, y, depth-code)]  (models_bts.py:138-216, nv_enc = 1)
# ----------------------------------------------------------------------------------------------
def sample_features(xyz: torch.Tensor, st: FieldState, cfg: FieldConfig):
    n, P, _ = xyz.shape
    xy, z, dist, invalid = project(xyz, st.w2c_enc.unsqueeze(1), st.K_enc.unsqueeze(1))
    if cfg.code_mode == "z":
        code = z
    elif cfg.code_mode == "distance":
        code = dist
    else:
        raise NotImplementedError(cfg.code_mode)
    if cfg.inv_z:
        code = (1 / code.clamp_min(EPS) - 1 / cfg.d_max) / (1 / cfg.d_min - 1 / cfg.d_max)
    else:
        code = (code - cfg.d_min) / (cfg.d_max - cfg.d_min)
    code = 2 * code - 1
    pe_in = torch.cat((xy, code), dim=-1).view(n * P, 3)
    pe = positional_encoding(pe_in, cfg.num_freqs, cfg.freq_factor, cfg.include_input).view(n, P, -1)
    f = _bilinear_border(st.feat, xy[:, 0])                                             # (n,P,C)
    inv = invalid[:, 0]                                                                 # (n,P,1)
    if cfg.learn_empty:
        f = torch.where(inv, st.empty_feature.view(1, 1, -1).expand_as(f), f)
    return torch.cat((f, pe), dim=-1), inv


# ----------------------------------------------------------------------------------------------
# a11: the MLP (resnetfc.py:132-184, 53-62)
# ----------------------------------------------------------------------------------------------
def mlp_forward(p: MlpParams, x: torch.Tensor) -> torch.Tensor:
    h = F.linear(x, p.w_in, p.b_in)
    for (w0, b0, w1, b1) in p.blocks:
        h = h + F.linear(torch.relu(F.linear(torch.relu(h), w0, b0)), w1, b1)
    return F.linear(torch.relu(h), p.w_out, p.b_out)


# ----------------------------------------------------------------------------------------------
# a12: colour taps from the nv render views (models_bts.py:218-264)
# ----------------------------------------------------------------------------------------------
def sample_colors(xyz: torch.Tensor, st: FieldState):
    n, P, _ = xyz.shape
    nv = st.imgs.shape[1]
    xy, _, _, invalid = project(xyz, st.w2c_r, st.K_r)
    c = _bilinear_border(st.imgs.reshape(n * nv, *st.imgs.shape[2:]), xy.reshape(n * nv, P, 2)).view(n, nv, P, 3)
    return c, invalid


# ----------------------------------------------------------------------------------------------
# a13: the field (models_bts.py:266-338)
# ----------------------------------------------------------------------------------------------
def field_forward(xyz: torch.Tensor, st: FieldState, mlp: MlpParams, cfg: FieldConfig, only_density: bool = False):
    """xyz (n, P, 3) -> rgb (n, P, nv*3), invalid (n, P, nv) float, sigma (n, P, 1)."""
    n, P, _ = xyz.shape
    x, inv_f = sample_features(xyz, st, cfg)
    sigma = F.softplus(mlp_forward(mlp, x)[..., :1])
    if cfg.empty_empty:
        sigma = torch.where(inv_f, torch.zeros_like(sigma), sigma)
    nv = st.imgs.shape[1]
    if only_density:
        return torch.zeros(n, P, nv * 3, device=sigma.device), inv_f.to(sigma.dtype), sigma
    c, inv_c = sample_colors(xyz, st)
    rgb = c.permute(0, 2, 1, 3).reshape(n, P, nv * 3)
    invalid = (inv_c.permute(0, 2, 1, 3).reshape(n, P, nv) | inv_f).to(rgb.dtype)
    return rgb, invalid, sigma


# ----------------------------------------------------------------------------------------------
# a5: alpha compositing (nerf.py:210-313)
# ----------------------------------------------------------------------------------------------
def composite(rays: torch.Tensor, z_samp: torch.Tensor, sb: int, st: FieldState, mlp: MlpParams, cfg: FieldConfig,
              hard_alpha_cap: bool = True, white_bkgd: bool = False, chunk: int = 100000, sigma_noise: Optional[torch.Tensor] = None):
    """rays (sb*B', 8), z_samp (sb*B', K) -> (weights (B,K), rgb (B,nv*3), depth (B), alphas (B,K),
    invalid (B,K,nv), z_samp (B,K), rgb_samps (B,K,nv*3)).  Point queries are chunked like nerf.py:238-268 (chunking does
    not change any value, only peak memory)."""
    B, K = z_samp.shape
    deltas = torch.cat((z_samp[:, 1:] - z_samp[:, :-1], torch.full((B, 1), 1e10, device=z_samp.device, dtype=z_samp.dtype)), dim=-1)
    pts = (rays[:, None, :3] + z_samp.unsqueeze(2) * rays[:, None, 3:6]).reshape(sb, -1, 3)
    per = (chunk - 1) // sb + 1
    rgbs, invs, sigs = [], [], []
    for part in torch.split(pts, per, dim=1):
        r, i, s = field_forward(part, st, mlp, cfg)
        rgbs.append(r), invs.append(i), sigs.append(s)
    rgbs = torch.cat(rgbs, dim=1).reshape(B, K, -1)
    invalid = torch.cat(invs, dim=1).reshape(B, K, -1)
    sigmas = torch.cat(sigs, dim=1).reshape(B, K)

    if sigma_noise is not None:   # nerf.py:279-280: training-mode density noise (the draw randn_like(sigmas) * noise_std is injected)
        sigmas = sigmas + sigma_noise
    alphas = 1 - torch.exp(-deltas.abs() * torch.relu(sigmas))
    if hard_alpha_cap:
        alphas = torch.cat((alphas[:, :-1], torch.ones_like(alphas[:, -1:])), dim=-1)
    trans = torch.cumprod(torch.cat((torch.ones_like(alphas[:, :1]), 1 - alphas + 1e-10), dim=-1), dim=-1)
    weights = alphas * trans[:, :-1]
    rgb = torch.sum(weights.unsqueeze(-1) * rgbs, dim=-2)
    depth = torch.sum(weights * z_samp, dim=-1)
    if white_bkgd:
        rgb = rgb + 1 - weights.sum(dim=1).unsqueeze(-1)
    return weights, rgb, depth, alphas, invalid, z_samp, rgbs


def render(rays: torch.Tensor, z_samp: torch.Tensor, st: FieldState, mlp: MlpParams, cfg: FieldConfig,
           hard_alpha_cap: bool = True, white_bkgd: bool = False):
    """rays (SB, B', 8), z_samp (SB*B', K) -> dict shaped like the "coarse" entry of nerf.py:377-401."""
    sb = rays.shape[0]
    w, rgb, depth, a, inv, z, rs = composite(rays.reshape(-1, 8), z_samp, sb, st, mlp, cfg, hard_alpha_cap, white_bkgd)
    K = z.shape[-1]
    return dict(rgb=rgb.reshape(sb, -1, rgb.shape[-1]), depth=depth.reshape(sb, -1),
                invalid=inv.reshape(sb, -1, K, inv.shape[-1]), weights=w.reshape(sb, -1, K),
                alphas=a.reshape(sb, -1, K), z_samps=z.reshape(sb, -1, K),
                rgb_samps=rs.reshape(sb, -1, K, rs.shape[-1]))


# ----------------------------------------------------------------------------------------------
# f3: occupancy profile (scripts/inference_setup.py:84-97 get_pts, :201-229 render_profile)
# ----------------------------------------------------------------------------------------------
def profile_points(x_range=(-9, 9), y_range=(.0, .75), z_range=(21, 3), x_res=256, y_res=64, z_res=256):
    """get_pts (inference_setup.py:84-97) with the OUT_RES defaults (:46-52), no camera-inclination adjustment: (y_res, z_res, x_res, 3),
    the vertical level y slowest."""
    x = torch.linspace(x_range[0], x_range[1], x_res).view(1, 1, x_res).expand(y_res, z_res, -1)
    z = torch.linspace(z_range[0], z_range[1], z_res).view(1, z_res, 1).expand(y_res, -1, x_res)
    y = torch.linspace(y_range[0], y_range[1], y_res).view(y_res, 1, 1).expand(-1, z_res, x_res)
    return torch.stack((x, y, z), dim=-1)


def occupancy_profile(q_pts: torch.Tensor, st: FieldState, mlp: MlpParams, cfg: FieldConfig, threshold: float = 8.0, batch_size: int = 50000):
    """render_profile (inference_setup.py:201-229): q_pts (Y, Z, X, 3) -> profile (Z, X), sigma (Y*Z*X,), invalid (Y*Z*X, nv).
    Field queries in chunks of 50 000 points (:205-217), sigma := 1 where ANY view flags the point (:219), running sum along y
    (:224), fraction of levels whose running sum is <= 8 (:225)."""
    Y, Z, X, _ = q_pts.shape
    pts = q_pts.reshape(1, -1, 3)
    sig, inv = [], []
    for f in range(0, pts.shape[1], batch_size):
        _, i_, s_ = field_forward(pts[:, f:f + batch_size], st, mlp, cfg)
        sig.append(s_), inv.append(i_)
    sigmas, invalid = torch.cat(sig, dim=1), torch.cat(inv, dim=1)
    raw = sigmas.reshape(-1).clone()
    sigmas[torch.any(invalid > 0, dim=-1)] = 1
    alphas = sigmas.reshape(Y, Z, X)
    profile = (torch.cumsum(alphas, dim=0) <= threshold).float().sum(dim=0) / Y
    return profile, raw, invalid[0]


# ----------------------------------------------------------------------------------------------
# a14: ray distance -> z depth (projection_operations.py:4-16)
# ----------------------------------------------------------------------------------------------
def distance_to_z(depths: torch.Tensor, projs: torch.Tensor) -> torch.Tensor:
    n, nv, h, w = depths.shape
    inv_K = torch.inverse(projs)
    gx = torch.linspace(-1, 1, w).view(1, 1, 1, -1).expand(-1, -1, h, -1)
    gy = torch.linspace(-1, 1, h).view(1, 1, -1, 1).expand(-1, -1, -1, w)
    pix = torch.stack((gx, gy, torch.ones_like(gx)), dim=2).expand(n, nv, -1, -1, -1)
    cam = (inv_K @ pix.reshape(n, nv, 3, -1)).view(n, nv, 3, h, w)
    return depths * (cam[:, :, 2] / torch.norm(cam, dim=2))


# ----------------------------------------------------------------------------------------------
# a15 (consumer, parity only): depth metrics formula (evaluator.py:96-151, abs_rel part)
# ----------------------------------------------------------------------------------------------
def abs_rel(depth_pred: torch.Tensor, depth_gt: torch.Tensor) -> float:
    """depth_pred (1,1,h,w) z-depth, depth_gt (1,1,H,W) sparse (0 = no measurement)."""
    pred = F.interpolate(depth_pred, tuple(depth_gt.shape[-2:]))
    pred = torch.clamp(pred, 1e-3, 80)
    mask = depth_gt != 0
    return torch.mean(torch.abs(depth_gt[mask] - pred[mask]) / depth_gt[mask]).item()


# ----------------------------------------------------------------------------------------------
# synthetic, seeded inputs shared by tests, smoke and bench
# ----------------------------------------------------------------------------------------------
K_KITTI360 = [[0.7849, 0.0, -0.0312], [0.0, 2.9391, 0.2701], [0.0, 0.0, 1.0]]     # gen_img_custom.py:54-59
K_KITTIRAW = [[1.1619, 0.0, -0.0184], [0.0, 3.8482, -0.0781], [0.0, 0.0, 1.0]]    # gen_img_custom.py:72-77
K_RE10K = [[1.0056, 0.0, 0.0], [0.0, 1.7877, 0.0], [0.0, 0.0, 1.0]]               # gen_img_custom.py:90-95


def _pose(tx=0.0, ty=0.0, tz=0.0, yaw_deg=0.0):
    a = math.radians(yaw_deg)
    m = torch.eye(4)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = math.cos(a), math.sin(a), -math.sin(a), math.cos(a)
    m[0, 3], m[1, 3], m[2, 3] = tx, ty, tz
    return m


def synthetic_scene(n: int, v: int, H: int, W: int, C: int, seed: int = 0, intrinsics=None, baseline: float = 0.54,
                    yaw_deg: float = 0.0, smooth: bool = False):
    """Seeded frames / feature map / poses.  View 0 is the keyframe (identity pose); odd views are the stereo partner
    (x += baseline), views >= 2 additionally move forward (z += 1 per temporal step) and may be yawed.
    Returns dict(images (n,v,3,H,W) in [-1,1], feat (n,C,H,W), projs (n,v,3,3), poses (n,v,4,4) c2w)."""
    g = torch.Generator().manual_seed(seed)
    Kmat = torch.tensor(K_KITTI360 if intrinsics is None else intrinsics, dtype=torch.float32)
    images = torch.rand(n, v, 3, H, W, generator=g) * 2 - 1
    feat = torch.randn(n, C, H, W, generator=g)
    if smooth:  # low-pass so that bilinear taps are well conditioned (used by gradient checks)
        feat = F.avg_pool2d(feat, 5, 1, 2) * 3
        images = F.avg_pool2d(images.view(n * v, 3, H, W), 5, 1, 2).view(n, v, 3, H, W) * 3
        images = images.clamp(-1, 1)
    poses = torch.stack([torch.stack([_pose(tx=baseline * (j % 2), tz=float(j // 2), yaw_deg=yaw_deg * (j // 2))
                                      for j in range(v)]) for _ in range(n)])
    # small per-sample perturbation so that batch elements differ
    poses[:, :, :3, 3] += 0.05 * torch.randn(n, v, 3, generator=g) * (torch.arange(v).view(1, v, 1) > 0)
    projs = Kmat.view(1, 1, 3, 3).expand(n, v, 3, 3).contiguous()
    return dict(images=images, feat=feat, projs=projs, poses=poses)


def make_state(scene, ids_render, cfg: FieldConfig, empty_feature=None) -> FieldState:
    """The part of ``BTSNet.encode`` that is not the CNN (models_bts.py:65-136) with ids_encoder=[0]."""
    poses_w2c = torch.inverse(scene["poses"])
    imgs01 = scene["images"] * 0.5 + 0.5
    return FieldState(feat=scene["feat"], K_enc=scene["projs"][:, 0].contiguous(), w2c_enc=poses_w2c[:, 0].contiguous(),
                      imgs=imgs01[:, ids_render].contiguous(), K_r=scene["projs"][:, ids_render].contiguous(),
                      w2c_r=poses_w2c[:, ids_render].contiguous(), empty_feature=empty_feature)
