"""The two field modes no shipped config uses, as PyTorch compositions ("provide via torch fallback, not
HIP"): several ENCODER views whose samples are merged per point (``combine_ids``, the waymo mode: models_bts.py:93-136, 190-210,
237-258) and colours predicted by the MLP instead of sampled from frames (``sample_color=False``, models_bts.py:315-321).

These run as ordinary PyTorch-ROCm ops on whatever device the tensors live on -- they are NOT the fused HIP path and make no claim
on its speed; ``BTSNet`` switches to them only for those two modes (and says so once), every shipped configuration stays on the
kernels of libbts_render.so.  Pinned to the real reference by tests/golden/modes.npz (tests/golden/gen_golden_modes.py)."""
import warnings

import torch
import torch.nn.functional as F

EPS = 1e-3
_warned = set()


def warn_once(mode):
    if mode not in _warned:
        _warned.add(mode)
        warnings.warn(f"behindthescenes_amd: `{mode}` is served by a PyTorch composition, not by the fused HIP kernels (no shipped "
                      "configuration uses it; SURVEY.md section 8 row a16)", stacklevel=3)


def _to_views(xyz, w2c, Ks):
    """World points (n, P, 3) into every view (n, V, ...): normalised image coordinates, camera depth, distance, frustum flag."""
    hom = torch.cat((xyz, torch.ones_like(xyz[..., :1])), dim=-1).unsqueeze(1)             # (n, 1, P, 4)
    cam = w2c[:, :, :3, :] @ hom.transpose(-1, -2)                                          # (n, V, 3, P)
    dist = torch.norm(cam, dim=-2).unsqueeze(-1)                                            # (n, V, P, 1)
    pix = (Ks @ cam).transpose(-1, -2)                                                      # (n, V, P, 3)
    depth = pix[..., 2:3]
    uv = pix[..., :2] / depth.clamp_min(EPS)
    outside = (depth <= EPS) | (uv[..., :1] < -1) | (uv[..., :1] > 1) | (uv[..., 1:2] < -1) | (uv[..., 1:2] > 1)
    return uv, depth, dist, outside


def _tap(maps, uv):
    """maps (n, V, c, h, w), uv (n, V, P, 2) -> (n, V, P, c): border-clamped bilinear taps (grid_sample, align_corners=False)."""
    n, V, c, h, w = maps.shape
    got = F.grid_sample(maps.reshape(n * V, c, h, w), uv.reshape(n * V, 1, -1, 2), mode="bilinear", padding_mode="border", align_corners=False)
    return got.view(n, V, c, -1).transpose(-1, -2)


def _merge_groups(values, outside, groups, singles_twice):
    """Per group of views keep, point by point, the member whose frustum flag is smallest (first one on ties): the reference's
    ``torch.min(invalid, dim=1)`` + gather.  ``singles_twice``: the feature merge of the reference appends a one-member group and then
    ALSO falls through to the general branch (models_bts.py:196-209 lacks the ``continue`` its colour twin has at :243), so such a view
    enters the later mean twice -- reproduced, since the reference's numbers are the contract."""
    vals, flags = [], []
    for g in groups:
        if len(g) == 1:
            vals.append(values[:, g]), flags.append(outside[:, g])
            if not singles_twice:
                continue
        o, v = outside[:, g], values[:, g]
        # the reference's `torch.min(invalid, dim=1)[1]`: on ties the CPU kernel returns the FIRST member (what the fixtures pin), the GPU
        # kernel whichever its reduction tree meets -- made explicit here so that every device gives the reference's CPU answer
        order = torch.arange(len(g), device=o.device, dtype=torch.int32).view(1, -1, 1, 1)
        pick = (o.to(torch.int32) * len(g) + order).argmin(dim=1, keepdim=True)
        flags.append(torch.gather(o, 1, pick))
        vals.append(torch.gather(v, 1, pick.expand(-1, -1, -1, v.shape[-1])))
    return torch.cat(vals, dim=1), torch.cat(flags, dim=1)


def sample_features(net, xyz, single=True):
    """models_bts.py:138-216 over ALL encoder views: (n, P, C + 39) features [+ per-view axis when not ``single``], frustum flag."""
    maps = net.grid_f_features[net._scale]
    n, V = maps.shape[:2]
    uv, depth, dist, outside = _to_views(xyz, net.grid_f_poses_w2c[:, :V], net.grid_f_Ks[:, :V])
    code = depth if net.code_mode == "z" else dist
    if net.inv_z:
        code = (1 / code.clamp_min(EPS) - 1 / net.d_max) / (1 / net.d_min - 1 / net.d_max)
    else:
        code = (code - net.d_min) / (net.d_max - net.d_min)
    enc_in = torch.cat((uv, 2 * code - 1), dim=-1)
    pe = net.code_xyz(enc_in.reshape(-1, 3)).view(n, V, xyz.shape[1], -1)
    feats = _tap(maps, uv)
    if net.learn_empty:
        feats = torch.where(outside, net.empty_feature.view(1, 1, 1, -1), feats)
    feats = torch.cat((feats, pe), dim=-1)
    if net.grid_f_combine is not None:
        feats, outside = _merge_groups(feats, outside, net.grid_f_combine, singles_twice=True)
    if single:
        feats, outside = feats.mean(dim=1), torch.any(outside, dim=1)
    return feats, outside


def sample_colors(net, xyz):
    """models_bts.py:218-264: (n, nv', P, 3) colours and flags, nv' = merged render-view groups when ``combine_ids`` was given."""
    imgs = net.grid_c_imgs
    uv, _, _, outside = _to_views(xyz, net.grid_c_poses_w2c, net.grid_c_Ks)
    cols = _tap(imgs, uv)
    if net.grid_c_combine is not None:
        cols, outside = _merge_groups(cols, outside, net.grid_c_combine, singles_twice=False)
    return cols, outside


def field_forward(net, xyz, coarse=True, only_density=False):
    """BTSNet.forward (models_bts.py:266-338) for the two modes: -> rgb (n, P, nv * 3), invalid (n, P, nv) float, sigma (n, P, 1)."""
    n, P, _ = xyz.shape
    nv = len(net.grid_c_combine) if net.grid_c_combine is not None else net.grid_c_imgs.shape[1]
    if only_density and net.grid_f_features[net._scale].shape[1] > 1:
        raise NotImplementedError("only_density with several encoder views: the reference itself fails here (models_bts.py:281-284 flattens "
                                  "the view axis into the feature axis)")
    x, inv_f = sample_features(net, xyz, single=not only_density)
    out = net.mlp(coarse)(x.reshape(n, P, -1)).reshape(n, P, net._d_out)
    if net.sample_color:
        sigma = F.softplus(out[..., :1])
        cols, inv_c = sample_colors(net, xyz)
    else:
        sigma = torch.relu(out[..., :1])
        cols, inv_c, nv = torch.sigmoid(out[..., 1:4]).reshape(n, 1, P, 3), inv_f.unsqueeze(-2), 1
    if net.empty_empty:
        sigma = torch.where(inv_f[..., :1].reshape(n, P, 1), torch.zeros_like(sigma), sigma)
    if only_density:
        return torch.zeros((n, P, nv * 3), device=sigma.device), inv_f.to(sigma.dtype), sigma
    rgb = cols.permute(0, 2, 1, 3).reshape(n, P, nv * 3)
    invalid = (inv_c.permute(0, 2, 1, 3).reshape(n, P, nv) | inv_f.reshape(n, P, 1)).to(rgb.dtype)
    return rgb, invalid, sigma


def sample_coarse(rays, u, lindisp):
    """nerf.py:103-123 from the caller's jitter ``u`` (B, K) in [0, 1): stratified depths, linear in disparity or in depth (only where the
    HIP routine cannot run: a CPU tensor in one of the PyTorch-composed modes)."""
    K = u.shape[1]
    step = 1.0 / K
    s = torch.linspace(0, 1 - step, K, device=rays.device, dtype=rays.dtype).unsqueeze(0) + u * step
    near, far = rays[:, -2:-1], rays[:, -1:]
    return 1 / (1 / near * (1 - s) + 1 / far * s) if lindisp else near * (1 - s) + far * s


def composite(renderer, net, rays, z_samp, coarse=True, sb=0):
    """NeRFRenderer.composite (nerf.py:210-313) on top of ``field_forward`` -- one field query for all points (no chunking), then the
    alpha compositing.  -> (weights, rgb, depth, alphas, invalid, z_samp, rgb_samps) like the reference."""
    B, K = z_samp.shape
    delta = torch.cat((z_samp[:, 1:] - z_samp[:, :-1], torch.full_like(z_samp[:, :1], 1e10)), dim=-1)
    pts = rays[:, None, :3] + z_samp.unsqueeze(2) * rays[:, None, 3:6]
    if sb > 0:
        pts = pts.reshape(sb, -1, 3)
    else:
        pts = pts.reshape(1, -1, 3)
    rgbs, invalid, sigmas = field_forward(net, pts, coarse=coarse)
    rgbs, invalid, sigmas = rgbs.reshape(B, K, -1), invalid.reshape(B, K, -1), sigmas.reshape(B, K)
    if renderer.training and renderer.noise_std > 0.0:
        sigmas = sigmas + torch.randn_like(sigmas) * renderer.noise_std
    alphas = 1 - torch.exp(-delta.abs() * torch.relu(sigmas))
    if renderer.hard_alpha_cap:
        alphas = torch.cat((alphas[:, :-1], torch.ones_like(alphas[:, -1:])), dim=-1)
    trans = torch.cumprod(torch.cat((torch.ones_like(alphas[:, :1]), 1 - alphas + 1e-10), dim=-1), dim=-1)
    weights = alphas * trans[:, :-1]
    rgb = torch.sum(weights.unsqueeze(-1) * rgbs, dim=-2)
    depth = torch.sum(weights * z_samp, dim=-1)
    if renderer.white_bkgd:
        rgb = rgb + 1 - weights.sum(dim=1).unsqueeze(-1)
    return weights, rgb, depth, alphas, invalid, z_samp, rgbs
