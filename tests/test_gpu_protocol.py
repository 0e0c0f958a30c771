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


def test_two_pass_forward_with_the_shared_mlp_end_to_end(hip):
    """`using_fine` (n_fine > 0) with mlp_fine: empty -- the fine pass queries the COARSE MLP (models_bts.py:300-304) on
    the sorted union of the coarse samples, n_fine - n_fine_depth importance samples and n_fine_depth samples around the coarse depth
    (nerf.py:352-373).  No shipped config turns it on; the drop-in runs it end to end on the fused kernel: both passes against the
    oracle on the very samples the renderer drew (want_z_samps), the fine samples inside [near, far], sorted, the coarse ones among
    them, and gradients reaching the MLP from both passes."""
    from oracle import bts_oracle as O
    from tests._cases import robust_ray_mask
    from tests._hip_helpers import build_net
    cfg = O.FieldConfig()
    g = torch.Generator().manual_seed(12)
    n, v, H, W, Kc, Kf, Kfd = 2, 3, 48, 160, 16, 12, 4
    scene = O.synthetic_scene(n, v, H, W, 64, seed=12, intrinsics=O.K_KITTI360, smooth=True)
    mlp = O.init_mlp(103, 64, 0, gen=g)
    rays = O.image_rays(scene["poses"], scene["projs"], H, W, 3.0, 80.0)
    rays = rays[:, torch.randperm(rays.shape[1], generator=g)[:640].sort().values].contiguous()
    net = build_net(cfg, mlp, scene, [1, 2], train=True)
    renderer = hip.NeRFRenderer.from_conf(dict(n_coarse=Kc, n_fine=Kf, n_fine_depth=Kfd, depth_std=1.0, lindisp=True, hard_alpha_cap=True)).cuda().train()
    assert renderer.using_fine
    torch.manual_seed(3)
    out = renderer.bind_parallel(net)(rays.cuda(), want_weights=True, want_alphas=True, want_z_samps=True)
    assert set(out) == {"coarse", "fine"}
    zc, zf = out["coarse"]["z_samps"], out["fine"]["z_samps"]
    assert zc.shape == (n, 640, Kc) and zf.shape == (n, 640, Kc + Kf)
    assert (zf[..., 1:] >= zf[..., :-1]).all() and zf.min() >= 3.0 - 1e-4 and zf.max() <= 80.0 + 1e-3
    # every coarse sample is among the fine pass' samples
    assert (torch.isclose(zc.unsqueeze(-1), zf.unsqueeze(-2), rtol=0, atol=0).any(-1)).all()
    st = O.make_state(scene, [1, 2], cfg)
    for part, z in (("coarse", zc), ("fine", zf)):
        zz = z.reshape(-1, z.shape[-1]).detach().cpu()
        with torch.no_grad():
            ow, orgb, odepth, oa, oinv, _, _ = O.composite(rays.reshape(-1, 8), zz, n, st, mlp, cfg, hard_alpha_cap=True)
        ok = robust_ray_mask(st, rays, zz)
        d = out[part]
        torch.testing.assert_close(d["depth"].detach().cpu().reshape(-1)[ok], odepth[ok], rtol=1e-4, atol=0)
        torch.testing.assert_close(d["rgb"].detach().cpu().reshape(-1, 6)[ok], orgb[ok], rtol=0, atol=1e-5)
        torch.testing.assert_close(d["weights"].detach().cpu().reshape(-1, z.shape[-1])[ok], ow[ok], rtol=0, atol=1e-5)
    # both passes are differentiable through the one MLP
    net.zero_grad(set_to_none=True)
    (out["coarse"]["rgb"].square().mean() + out["fine"]["rgb"].square().mean() + 0.01 * out["fine"]["depth"].mean()).backward()
    gw = net.mlp_coarse.lin_in.weight.grad
    assert gw is not None and torch.isfinite(gw).all() and float(gw.abs().sum()) > 0


@pytest.mark.parametrize("model", ["kitti", "re10k"])
def test_density_noise_forward_and_backward_vs_oracle(hip, model):
    """SURVEY 8 row a16, nerf.py:279-280: in training mode the reference adds randn_like(sigmas) * noise_std to the densities before
    relu / alpha (no shipped config turns it on).  The drop-in draws the noise and hands it to the kernels (BtsRenderArgs.sigma_noise):
    with the draw injected, outputs and gradients against the oracle -- the relu cuts the gradient where sigma + noise <= 0, through
    both backward paths (gate bits: KITTI MLP; rows: RE10K MLP)."""
    from oracle import bts_oracle as O
    from tests._cases import robust_ray_mask
    from tests._hip_helpers import build_net
    from tests.test_gpu_grad import _gate_safe_rays, _rel_to_max
    g = torch.Generator().manual_seed(21)
    kitti = model == "kitti"
    cfg = O.FieldConfig() if kitti else O.FieldConfig(d_min=1.0, d_max=100.0, code_mode="distance")
    C, nb, K, n, H, W = (64, 0, 32, 2, 48, 160) if kitti else (32, 1, 24, 2, 64, 96)
    scene = O.synthetic_scene(n, 3, H, W, C, seed=21, intrinsics=O.K_KITTI360 if kitti else O.K_RE10K, baseline=0.5, smooth=True)
    mlp = O.init_mlp(C + 39, C, nb, gen=g)
    mlp.b_out = torch.tensor([-1.0])            # densities of ~0.3: noise of std 0.5 pushes a good part of them below zero
    rays = O.image_rays(scene["poses"], scene["projs"], H, W, cfg.d_min, cfg.d_max)
    rays = rays[:, torch.randperm(rays.shape[1], generator=g)[:1200].sort().values].contiguous()
    z = O.sample_coarse(rays.reshape(-1, 8), K, True, torch.rand(n * 1200, K, generator=g))
    keep = robust_ray_mask(O.make_state(scene, [1, 2], cfg), rays, z).view(n, -1)
    safe, _ = _gate_safe_rays(scene, mlp, cfg, [1, 2], rays, z, None, margin=2e-5)
    keep = keep & safe
    NR = min(512, int(keep.sum(1).min()))
    assert NR >= 256
    idx = torch.stack([torch.nonzero(keep[i])[:NR, 0] for i in range(n)])
    rays = torch.gather(rays, 1, idx.unsqueeze(-1).expand(-1, -1, 8)).contiguous()
    z = torch.gather(z.view(n, -1, K), 1, idx.unsqueeze(-1).expand(-1, -1, K)).reshape(-1, K).contiguous()
    noise = torch.randn(n * NR, K, generator=g) * 0.5
    c_rgb = torch.randn(n * NR, 6, generator=g)
    # oracle
    params = [t.clone().requires_grad_(True) for t in mlp.tensors()]
    feat = scene["feat"].clone().requires_grad_(True)
    m = O.MlpParams(params[0], params[1], [tuple(params[2 + 4 * i: 6 + 4 * i]) for i in range(nb)], params[-2], params[-1])
    st = O.make_state(dict(scene, feat=feat), [1, 2], cfg)
    ow, orgb, odepth, oa, *_ = O.composite(rays.reshape(-1, 8), z, n, st, m, cfg, hard_alpha_cap=kitti, sigma_noise=noise)
    cut = float((oa[:, :-1] == 0).float().mean())
    assert 0.05 < cut < 0.95, cut               # the noise really switches a share of the samples off
    ref = torch.autograd.grad((orgb * c_rgb).sum() + 0.05 * odepth.sum(), params + [feat])
    # HIP: the renderer in training mode draws its own noise unless one is injected
    net = build_net(cfg, mlp, scene, [1, 2], train=True)
    renderer = hip.NeRFRenderer.from_conf(dict(n_coarse=K, lindisp=True, hard_alpha_cap=kitti, noise_std=0.5)).cuda().train()
    w, rgb, depth, a, *_ = renderer.composite(net, rays.reshape(-1, 8).cuda(), z.cuda(), sb=n, sigma_noise=noise.cuda())
    torch.testing.assert_close(depth.detach().cpu(), odepth.detach(), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(rgb.detach().cpu(), orgb.detach(), rtol=0, atol=1e-5)
    # (alphas amplify the last bits of sigma by delta exp(-delta sigma), delta up to 20 m here: the 3 x noise-floor bound of test_gpu_parity._check)
    torch.testing.assert_close(a.detach().cpu(), oa.detach(), rtol=0, atol=1.5e-4)
    assert ((a.detach().cpu() - oa.detach()).abs() > 1e-5).float().mean().item() <= 2e-4
    ((rgb * c_rgb.cuda()).sum() + 0.05 * depth.sum()).backward()
    mc = net.mlp_coarse
    ours = [mc.lin_in.weight.grad, mc.lin_in.bias.grad] + sum([[b.fc_0.weight.grad, b.fc_0.bias.grad, b.fc_1.weight.grad, b.fc_1.bias.grad] for b in mc.blocks], []) \
        + [mc.lin_out.weight.grad, mc.lin_out.bias.grad, net.encoder.feats[0].grad]
    for i, (x, y) in enumerate(zip(ours, ref)):
        err = _rel_to_max(x, y.view_as(x.cpu()))
        assert err <= 1e-4, (i, err)
    # without an injected draw two training-mode renders differ (fresh noise), two eval-mode renders do not (no noise)
    with torch.no_grad():
        r1 = renderer.composite(net, rays.reshape(-1, 8).cuda(), z.cuda(), sb=n)[2]
        r2 = renderer.composite(net, rays.reshape(-1, 8).cuda(), z.cuda(), sb=n)[2]
        assert not torch.equal(r1, r2)
        renderer.eval()
        e1 = renderer.composite(net, rays.reshape(-1, 8).cuda(), z.cuda(), sb=n)[2]
        e2 = renderer.composite(net, rays.reshape(-1, 8).cuda(), z.cuda(), sb=n)[2]
        assert torch.equal(e1, e2)


def test_density_noise_vs_reference_golden(hip):
    """nerf.py:279-280 against the REAL reference: tests/golden/noise.npz (gen_golden_noise.py) holds the outputs and autograd gradients
    of the reference's renderer in train() mode with noise_std = 0.7 together with the noise tensor its seeded draw produced.  The HIP
    forward and backward with that tensor in BtsRenderArgs.sigma_noise (17 % of the samples end up on relu's zero side)."""
    from tests._cases import Case
    from tests._hip_helpers import net_from_case
    from tests.test_gpu_grad import _rel_to_max, GRAD_RTOL
    c = Case("noise")
    t = c.t
    net = net_from_case(c, train=True)
    renderer = hip.NeRFRenderer.from_conf(dict(n_coarse=c.meta["K"], lindisp=True, hard_alpha_cap=True, noise_std=c.meta["noise_std"])).cuda().train()
    net.zero_grad(set_to_none=True)
    w, rgb, depth, a, inv, *_ = renderer.composite(net, c.rays.reshape(-1, 8).cuda(), c.z_samp.cuda(), sb=c.rays.shape[0],
                                                   sigma_noise=t["sigma_noise"].cuda())
    assert torch.equal(inv.cpu(), t["out_invalid"])
    torch.testing.assert_close(depth.detach().cpu(), t["out_depth"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(rgb.detach().cpu(), t["out_rgb"], rtol=0, atol=1e-5)
    torch.testing.assert_close(w.detach().cpu(), t["out_weights"], rtol=0, atol=1e-5)
    torch.testing.assert_close(a.detach().cpu(), t["out_alphas"], rtol=0, atol=2e-5)
    ((rgb * t["gin_rgb"].cuda()).sum() + (depth * t["gin_depth"].cuda()).sum()).backward()
    mc = net.mlp_coarse
    ours = [mc.lin_in.weight.grad, mc.lin_in.bias.grad, mc.lin_out.weight.grad, mc.lin_out.bias.grad, net.encoder.feats[0].grad]
    for g_, nme in zip(ours, ["g_w_in", "g_b_in", "g_w_out", "g_b_out", "g_feat"]):
        assert g_ is not None, nme
        err = _rel_to_max(g_, t[nme].view_as(g_.cpu()))
        assert err <= GRAD_RTOL, (nme, err)
