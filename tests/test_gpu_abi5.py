"""GPU: what ABI 5 added to the render call, and the separate fine MLP.

* BtsFieldCfg.enc_render_view -- a render view that IS the encoder frame (eval_depth: ids_render = [0]) takes its projection and bilinear
  weights from the encoder view's: results must be bit-identical to evaluating them a second time (hint off).
* BtsRenderArgs.jitter / z_samp_out / lindisp -- NeRFRenderer.sample_coarse (nerf.py:103-123) inside the render kernel: bit-identical
  depths AND outputs against bts_sample_coarse + the injected-z_samp call, every lanes-per-ray mode, K > 64, both spacings; the training
  path hands the kernel's depths to the backward.
* mlp_fine (models_bts.py:45, 293-307): `coarse=False` selects a second packed parameter vector through the same kernels."""
import dataclasses

import pytest
import torch

from oracle import bts_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import behindthescenes_amd as bts
    from behindthescenes_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    _lib.load()
    return bts


def _scene_net(n, v, H, W, C, Hd, nb, ids_render, cfg, seed, intr=O.K_KITTIRAW, train=False):
    from tests._hip_helpers import build_net
    g = torch.Generator().manual_seed(seed)
    scene = O.synthetic_scene(n, v, H, W, C, seed=seed, intrinsics=intr, smooth=True)
    mlp = O.init_mlp(C + 39, Hd, nb, gen=g)
    empty = torch.randn(C, generator=g) if cfg.learn_empty else None
    net = build_net(cfg, mlp, scene, ids_render, empty_feature=empty, train=train)
    return scene, mlp, empty, net, g


@pytest.mark.parametrize("ids_render,K", [([0], 64), ([1, 0], 64), ([0, 1, 2], 32), ([0], 128)])
def test_encoder_view_hint_is_bit_identical(hip, ids_render, K):
    from behindthescenes_amd import native
    cfg = O.FieldConfig(learn_empty=True)
    scene, mlp, empty, net, g = _scene_net(1, 3, 96, 320, 64, 64, 0, ids_render, cfg, seed=41)
    ft = net.native_field()
    assert ft.enc_view == ids_render.index(0) and ft.cfg().enc_render_view == ids_render.index(0)
    rays = O.image_rays(scene["poses"], scene["projs"], 96, 320, cfg.d_min, cfg.d_max).reshape(-1, 8).cuda()
    z = native.sample_coarse(rays, torch.rand(rays.shape[0], K, generator=g).cuda(), True)
    params = net.mlp_coarse.packed().detach()
    kw = dict(hard_alpha_cap=True, want_weights=True, want_alphas=True, want_rgb_samps=True)
    a = native.render_fwd(ft, params, rays, z, **kw)
    off = native.FieldTensors(ft.spec, ft.proj_nhwc, ft.K_enc, ft.w2c_enc, ft.imgs_nhwc4, ft.K_r, ft.w2c_r, ft.empty_feature, enc_view=-1)
    assert off.cfg().enc_render_view == -1
    b = native.render_fwd(off, params, rays, z, **kw)
    for k in ("rgb", "depth", "weights", "alphas", "invalid", "rgb_samps"):
        assert torch.equal(a[k], b[k]), k
    # the field query and the occupancy profile read the same hint
    pts = (rays[:4096, :3] + z[:4096, 7:8] * rays[:4096, 3:6]).reshape(1, -1, 3).contiguous()
    qa, qb = native.field_query(ft, params, pts), native.field_query(off, params, pts)
    for x, y in zip(qa, qb):
        assert torch.equal(x, y)


@pytest.mark.parametrize("K,n_rays,lindisp,model", [(64, 5000, True, "kitti"), (48, 3001, True, "re10k"), (32, 4096, True, "kitti"), (32, 4097, False, "kitti"),
                                                   (16, 2048, True, "kitti"), (8, 1000, True, "kitti"), (128, 1500, True, "re10k"), (130, 700, False, "kitti")])
def test_sample_coarse_inside_the_kernel_is_bit_identical(hip, K, n_rays, lindisp, model):
    from behindthescenes_amd import native
    re = model == "re10k"
    cfg = O.FieldConfig(d_min=1.0, d_max=100.0, code_mode="distance") if re else O.FieldConfig(learn_empty=True)
    C, Hd, nb = (32, 32, 1) if re else (64, 64, 0)
    scene, mlp, empty, net, g = _scene_net(2, 3, 64, 96, C, Hd, nb, [1, 2], cfg, seed=50 + K, intr=O.K_RE10K if re else O.K_KITTI360)
    rays = O.image_rays(scene["poses"], scene["projs"], 64, 96, cfg.d_min, cfg.d_max)
    idx = torch.randperm(rays.shape[1], generator=g)[:n_rays].sort().values
    rays = rays[:, idx].reshape(-1, 8).contiguous().cuda()
    u = torch.rand(rays.shape[0], K, generator=g).cuda()
    ft, params = net.native_field(), net.mlp_coarse.packed().detach()
    z = native.sample_coarse(rays, u, lindisp)
    kw = dict(hard_alpha_cap=not re, want_weights=True, want_alphas=True, want_rgb_samps=True)
    a = native.render_fwd(ft, params, rays, z, **kw)
    b = native.render_fwd(ft, params, rays, None, jitter=u, lindisp=lindisp, want_z=True, **kw)
    assert torch.equal(b["z_samp"], z), "in-kernel depths differ from bts_sample_coarse"
    for k in ("rgb", "depth", "weights", "alphas", "invalid", "rgb_samps"):
        assert torch.equal(a[k], b[k]), k
    c = native.render_fwd(ft, params, rays, None, jitter=u, lindisp=lindisp, **kw)    # depths not materialised
    assert c["z_samp"] is None and torch.equal(c["depth"], a["depth"]) and torch.equal(c["weights"], a["weights"])
    # oracle anchor for the depths themselves (nerf.py:103-123)
    torch.testing.assert_close(z.cpu(), O.sample_coarse(rays.cpu(), K, lindisp, u.cpu()), rtol=3e-6, atol=0)


def test_training_step_through_the_in_kernel_sampling(hip):
    """renderer(...) in training mode: the jitter is drawn by torch, the depths come out of the forward kernel (z_samp_out) and reach
    bts_render_bwd; gradients equal those of the injected-z_samp route on the same depths."""
    cfg = O.FieldConfig()
    scene, mlp, empty, net, g = _scene_net(2, 3, 48, 160, 64, 64, 0, [1, 2], cfg, seed=61, intr=O.K_KITTI360, train=True)
    renderer = hip.NeRFRenderer.from_conf(dict(n_coarse=64, lindisp=True, hard_alpha_cap=True)).cuda().train()
    rays = O.image_rays(scene["poses"], scene["projs"], 48, 160, 3.0, 80.0)[:, :1536].contiguous().cuda()
    torch.manual_seed(9)
    out = renderer.bind_parallel(net)(rays, want_weights=True, want_z_samps=True)["coarse"]
    z = out["z_samps"]
    assert z.shape == (2, 1536, 64) and (z[..., 1:] >= z[..., :-1]).all() and z.min() >= 3.0 - 1e-4 and z.max() <= 80.0 + 1e-3
    c_rgb = torch.randn(2, 1536, 6, generator=g).cuda()
    net.zero_grad(set_to_none=True)
    ((out["rgb"] * c_rgb).sum() + 0.05 * out["depth"].sum()).backward()
    g1 = [p.grad.clone() for p in (net.mlp_coarse.lin_in.weight, net.mlp_coarse.lin_out.weight, net.encoder.feats[0])]
    net.zero_grad(set_to_none=True)
    net.encode(scene["images"].cuda(), scene["projs"].cuda(), scene["poses"].cuda(), ids_encoder=[0], ids_render=[1, 2])
    w, rgb, depth, *_ = renderer.composite(net, rays.reshape(-1, 8), z.reshape(-1, 64).detach(), sb=2)
    assert torch.equal(rgb.reshape(2, 1536, 6), out["rgb"]) and torch.equal(depth.reshape(2, 1536), out["depth"])
    ((rgb.reshape(2, 1536, 6) * c_rgb).sum() + 0.05 * depth.sum()).backward()
    g2 = [p.grad for p in (net.mlp_coarse.lin_in.weight, net.mlp_coarse.lin_out.weight, net.encoder.feats[0])]
    for a, b in zip(g1, g2):
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()      # float atomics: summation order only


def test_separate_fine_mlp_through_the_same_kernels(hip):
    """models_bts.py:293-307: with `mlp_fine` configured, `coarse=False` queries IT.  The two-pass renderer (n_fine > 0) then runs the
    coarse pass on mlp_coarse and the fine pass on mlp_fine -- each against the oracle with that MLP, on the samples the renderer
    drew; gradients reach both MLPs; net(xyz, coarse=False) queries the fine MLP; state-dict keys follow the reference's."""
    from tests._cases import robust_ray_mask
    from tests._hip_helpers import make_conf, load_mlp
    import behindthescenes_amd as bts
    cfg = O.FieldConfig()
    g = torch.Generator().manual_seed(77)
    n, v, H, W, Kc, Kf = 2, 3, 48, 160, 16, 16
    scene = O.synthetic_scene(n, v, H, W, 64, seed=77, intrinsics=O.K_KITTI360, smooth=True)
    mlp_c, mlp_f = O.init_mlp(103, 64, 0, gen=g), O.init_mlp(103, 64, 0, gen=g)
    conf = make_conf(cfg, 64, 64, 0, H, W)
    conf["mlp_fine"] = dict(type="resnet", n_blocks=0, d_hidden=64)
    net = bts.BTSNet(conf)
    assert any(k.startswith("mlp_fine.lin_in") for k in net.state_dict())
    load_mlp(net, mlp_c)
    with torch.no_grad():
        net.mlp_fine.lin_in.weight.copy_(mlp_f.w_in), net.mlp_fine.lin_in.bias.copy_(mlp_f.b_in)
        net.mlp_fine.lin_out.weight.copy_(mlp_f.w_out), net.mlp_fine.lin_out.bias.copy_(mlp_f.b_out)
        net.encoder.feats[0].data = scene["feat"].clone()
    net = net.cuda().train()
    net.encode(scene["images"].cuda(), scene["projs"].cuda(), scene["poses"].cuda(), ids_encoder=[0], ids_render=[1, 2])
    renderer = hip.NeRFRenderer.from_conf(dict(n_coarse=Kc, n_fine=Kf, n_fine_depth=4, depth_std=1.0, lindisp=True, hard_alpha_cap=True)).cuda().train()
    rays = O.image_rays(scene["poses"], scene["projs"], H, W, 3.0, 80.0)
    rays = rays[:, torch.randperm(rays.shape[1], generator=g)[:512].sort().values].contiguous()
    torch.manual_seed(4)
    out = renderer.bind_parallel(net)(rays.cuda(), want_weights=True, want_z_samps=True)
    st = O.make_state(scene, [1, 2], cfg)
    for part, mlp in (("coarse", mlp_c), ("fine", mlp_f)):
        z = out[part]["z_samps"]
        zz = z.reshape(-1, z.shape[-1]).detach().cpu()
        with torch.no_grad():
            ow, orgb, odepth, *_ = O.composite(rays.reshape(-1, 8), zz, n, st, mlp, cfg, hard_alpha_cap=True)
        ok = robust_ray_mask(st, rays, zz)
        d = out[part]
        torch.testing.assert_close(d["depth"].detach().cpu().reshape(-1)[ok], odepth[ok], rtol=1e-4, atol=0)
        torch.testing.assert_close(d["rgb"].detach().cpu().reshape(-1, 6)[ok], orgb[ok], rtol=0, atol=1e-5)
        torch.testing.assert_close(d["weights"].detach().cpu().reshape(-1, z.shape[-1])[ok], ow[ok], rtol=0, atol=1e-5)
    # the two MLPs differ, so the passes must: the fine pass through the coarse MLP would fail the bound above
    net.zero_grad(set_to_none=True)
    (out["coarse"]["rgb"].square().mean() + out["fine"]["rgb"].square().mean()).backward()
    for m in (net.mlp_coarse, net.mlp_fine):
        gw = m.lin_in.weight.grad
        assert gw is not None and torch.isfinite(gw).all() and float(gw.abs().sum()) > 0
    # point queries: coarse=False -> mlp_fine (models_bts.py:300-307)
    pts = (rays[:, :64, :3] + 9.0 * rays[:, :64, 3:6]).cuda().contiguous()
    with torch.no_grad():
        _, _, s_c = net(pts, coarse=True)
        _, _, s_f = net(pts, coarse=False)
        _, _, o_f = O.field_forward(pts.cpu(), st, mlp_f, cfg)
    assert not torch.allclose(s_c, s_f)
    torch.testing.assert_close(s_f.cpu(), o_f, rtol=1e-4, atol=1e-6)
