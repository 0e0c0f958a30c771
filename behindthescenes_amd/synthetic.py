"""Seeded synthetic inputs for benchmarks and probes: frames, a stand-in feature map, normalised
intrinsics, poses and reference-style MLP initialisation.  Input generation only -- no part of the render path lives here, and
nothing here touches ``oracle/`` (the tests check that these generators and the oracle's produce identical tensors)."""
import math

import torch
import torch.nn.functional as F

# default normalised intrinsics of the reference's demo scripts (scripts/images/gen_img_custom.py:54-59, 72-77, 90-95)
K_KITTI360 = [[0.7849, 0.0, -0.0312], [0.0, 2.9391, 0.2701], [0.0, 0.0, 1.0]]
K_KITTIRAW = [[1.1619, 0.0, -0.0184], [0.0, 3.8482, -0.0781], [0.0, 0.0, 1.0]]
K_RE10K = [[1.0056, 0.0, 0.0], [0.0, 1.7877, 0.0], [0.0, 0.0, 1.0]]


def field_conf(C, d_hidden, n_blocks, H, W, z_near=3.0, z_far=80.0, inv_z=True, code_mode="z", learn_empty=False,
               empty_empty=False, num_freqs=6, freq_factor=1.5):
    """A ``BTSNet`` config dict with the reference's keys (configs/*.yaml ``model_conf``), the CNN replaced by a feature-map
    stand-in of the same output shape."""
    return dict(z_near=z_near, z_far=z_far, inv_z=inv_z, learn_empty=learn_empty, empty_empty=empty_empty, code_mode=code_mode,
                code=dict(num_freqs=num_freqs, freq_factor=freq_factor, include_input=True),
                encoder=dict(type="feature_map", size=(H, W), d_out=C),
                mlp_coarse=dict(type="resnet", n_blocks=n_blocks, d_hidden=d_hidden), mlp_fine=dict(type="empty"))


def _pose(tx=0.0, ty=0.0, tz=0.0, yaw_deg=0.0):
    a = math.radians(yaw_deg)
    m = torch.eye(4)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = math.cos(a), math.sin(a), -math.sin(a), math.cos(a)
    m[0, 3], m[1, 3], m[2, 3] = tx, ty, tz
    return m


def synthetic_scene(n, v, H, W, C, seed=0, intrinsics=None, baseline=0.54, yaw_deg=0.0, smooth=False):
    """View 0 is the keyframe (identity pose); odd views are the stereo partner (x += baseline); views >= 2 also move forward
    1 m per temporal step.  -> dict(images (n,v,3,H,W) in [-1,1], feat (n,C,H,W), projs (n,v,3,3), poses (n,v,4,4) c2w)."""
    g = torch.Generator().manual_seed(seed)
    Kmat = torch.tensor(K_KITTI360 if intrinsics is None else intrinsics, dtype=torch.float32)
    images = torch.rand(n, v, 3, H, W, generator=g) * 2 - 1
    feat = torch.randn(n, C, H, W, generator=g)
    if smooth:
        feat = F.avg_pool2d(feat, 5, 1, 2) * 3
        images = (F.avg_pool2d(images.view(n * v, 3, H, W), 5, 1, 2).view(n, v, 3, H, W) * 3).clamp(-1, 1)
    poses = torch.stack([torch.stack([_pose(tx=baseline * (j % 2), tz=float(j // 2), yaw_deg=yaw_deg * (j // 2))
                                      for j in range(v)]) for _ in range(n)])
    poses[:, :, :3, 3] += 0.05 * torch.randn(n, v, 3, generator=g) * (torch.arange(v).view(1, v, 1) > 0)
    projs = Kmat.view(1, 1, 3, 3).expand(n, v, 3, 3).contiguous()
    return dict(images=images, feat=feat, projs=projs, poses=poses)


def init_mlp_(mlp, seed=7, out_std=0.3):
    """In-place reference initialisation of a ``ResnetFC`` (kaiming-normal fan_in, zero bias: resnetfc.py:36-39, 88-94), then
    lin_out.weight ~ N(0, out_std) and fc_1 non-zero so that the density is not constant."""
    g = torch.Generator().manual_seed(seed)

    def kaiming(o, i):
        return torch.randn(o, i, generator=g) * math.sqrt(2.0 / i)

    with torch.no_grad():
        hd, d_in = mlp.lin_in.weight.shape
        mlp.lin_in.weight.copy_(kaiming(hd, d_in)), mlp.lin_in.bias.zero_()
        for blk in mlp.blocks:
            blk.fc_0.weight.copy_(kaiming(hd, hd)), blk.fc_0.bias.zero_()
            blk.fc_1.weight.copy_(kaiming(hd, hd) * 0.5), blk.fc_1.bias.zero_()
        mlp.lin_out.weight.copy_(torch.randn(mlp.lin_out.weight.shape, generator=g) * out_std), mlp.lin_out.bias.zero_()
    return mlp


def set_feature_map(net, feat):
    """Loads ``feat`` (n, C, H, W) into the feature-map stand-in encoder of ``net``."""
    with torch.no_grad():
        p = net.encoder.feats[0]
        from .native import is_channels_last
        fmt = torch.channels_last if is_channels_last(p) else torch.contiguous_format      # (the parameter keeps its memory format)
        p.data = feat.to(p.device).clone().contiguous(memory_format=fmt)
    return net


def build_net(scene, d_hidden=64, n_blocks=0, ids_render=(0,), device="cuda", train=False, mlp_seed=7, **conf):
    """``BTSNet`` over the scene's stand-in feature map with seeded MLP weights, encoded and ready to render."""
    from .field import BTSNet
    n, C, H, W = scene["feat"].shape
    net = BTSNet(field_conf(C, d_hidden, n_blocks, H, W, **conf))
    init_mlp_(net.mlp_coarse, seed=mlp_seed)
    set_feature_map(net, scene["feat"])
    net = net.to(device)
    net.train(train)
    net.encode(scene["images"].to(device), scene["projs"].to(device), scene["poses"].to(device), ids_encoder=[0],
               ids_render=list(ids_render))
    return net


def profile_points(x_range=(-9, 9), y_range=(.0, .75), z_range=(21, 3), x_res=256, y_res=64, z_res=256):
    """The query grid of the reference's occupancy profile (scripts/inference_setup.py:84-97 with the defaults of :46-52):
    (y_res, z_res, x_res, 3), the vertical level slowest -- 4.19 M points."""
    x = torch.linspace(x_range[0], x_range[1], x_res).view(1, 1, x_res).expand(y_res, z_res, -1)
    z = torch.linspace(z_range[0], z_range[1], z_res).view(1, z_res, 1).expand(y_res, -1, x_res)
    y = torch.linspace(y_range[0], y_range[1], y_res).view(y_res, 1, 1).expand(-1, z_res, x_res)
    return torch.stack((x, y, z), dim=-1)
