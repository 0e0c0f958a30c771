"""GPU: sampler front ends and the white-background branch against fixtures of the REAL reference (tests/golden/protocol.npz,
written by tests/golden/gen_golden_protocol.py): ImageRaySampler.sample (rays of every pixel of every frame + ground-truth colours),
RandomRaySampler.sample with the reference's seeded CPU draws, composite(white_bkgd=True) outputs and gradients."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = {k: torch.from_numpy(v) if v.dtype.kind == "f" else v
     for k, v in np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "protocol.npz")).items()}


@pytest.fixture(scope="module")
def hip():
    import behindthescenes_amd as bts
    from behindthescenes_amd import _lib
    assert torch.cuda.is_available()
    _lib.load()
    return bts


def test_image_ray_sampler_vs_reference(hip):
    images, poses, projs = G["img_images"].cuda(), G["img_poses"].cuda(), G["img_projs"].cuda()
    s = hip.ImageRaySampler(3.0, 80.0)                       # height / width from the images at first use, like the reference
    rays, gt = s.sample(images, poses, projs)
    assert (s.height, s.width) == (6, 10)
    torch.testing.assert_close(rays.cpu(), G["img_rays"], rtol=0, atol=2e-6)
    assert torch.equal(gt.cpu(), G["img_gt"])
    rays_u, gt_u = hip.ImageRaySampler(3.0, 80.0, 6, 10, norm_dir=False).sample(None, poses, projs)
    assert gt_u is None
    torch.testing.assert_close(rays_u.cpu(), G["img_rays_unnorm"], rtol=0, atol=2e-6)
    # reconstruct on device tensors: views only, values identical to the reference's
    part = {k[7:]: G[k].cuda() for k in G if k.startswith("img_in_")}
    rd = s.reconstruct(dict(coarse=dict(part), fine=dict(part), rgb_gt=gt))
    for k, t in rd["coarse"].items():
        assert torch.equal(t.cpu(), G[f"img_out_{k}"]), k
    assert torch.equal(rd["rgb_gt"].cpu(), G["img_out_rgb_gt"])


def test_random_ray_sampler_vs_reference(hip):
    images, poses, projs = G["img_images"].cuda(), G["img_poses"].cuda(), G["img_projs"].cuda()
    s = hip.RandomRaySampler(ray_batch_size=37, z_near=3.0, z_far=80.0)
    torch.manual_seed(123)                                   # the reference draws its pixel indices from the CPU generator
    rays, gt = s.sample(images, poses, projs)
    torch.testing.assert_close(rays.cpu(), G["rnd_rays"], rtol=0, atol=2e-6)
    assert torch.equal(gt.cpu(), G["rnd_gt"])


def test_white_background_forward_and_backward_vs_reference(hip):
    """nerf.py:301-304: rgb + 1 - sum(weights).  Outputs within the forward tolerances, every gradient within 1e-4 of its largest
    entry (the -sum_channels(g_rgb) term reaches the weights inside bts_render_bwd)."""
    from oracle import bts_oracle as O
    from tests._hip_helpers import build_net
    from tests.test_gpu_grad import _hip_grads, _rel_to_max
    cfg = O.FieldConfig()
    mlp = O.MlpParams(G["wb_w_in"], G["wb_b_in"], [], G["wb_w_out"], G["wb_b_out"])
    scene = dict(images=G["wb_images"], feat=G["wb_feat"], projs=G["wb_projs"], poses=G["wb_poses"])
    net = build_net(cfg, mlp, scene, [1, 2], train=True)
    renderer = hip.NeRFRenderer(n_coarse=G["wb_z"].shape[1], lindisp=True, hard_alpha_cap=False, white_bkgd=True).cuda()
    rays, z = G["wb_rays"].reshape(-1, 8).cuda(), G["wb_z"].cuda()
    with torch.no_grad():
        w, rgb, depth, *_ = renderer.composite(net, rays, z, sb=2)
    torch.testing.assert_close(rgb.cpu(), G["wb_rgb"], rtol=0, atol=1e-5)
    torch.testing.assert_close(w.cpu(), G["wb_weights"], rtol=0, atol=1e-5)
    torch.testing.assert_close(depth.cpu(), G["wb_depth"], rtol=1e-4, atol=1e-5)
    g_rgb = G["wb_gin_rgb"].cuda()
    grads = _hip_grads(hip, net, renderer, rays, z, 2, lambda w, rgb, depth, a: (rgb * g_rgb).sum())
    for g, name in zip(grads, ["g_w_in", "g_b_in", "g_w_out", "g_b_out", "g_feat"]):
        err = _rel_to_max(g, G[f"wb_{name}"].view_as(g.cpu()))
        assert err <= 1e-4, (name, err)
