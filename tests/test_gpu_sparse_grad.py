"""GPU: the sparse gradient of the projected map (ABI 6).

A training step's rays reach 8-15 % of the texels of G, so bts_render_bwd can flag the 64-texel tiles it adds into
(BtsRenderGrads.d_proj_tiles) and bts_project_features_bwd_tiles reads those only, writes d_feat densely and -- with clear_after --
returns the (d_proj, tiles) pair to all zero, which spares the step a map-sized fill and a map-sized read.  Checked here:
* the tile kernel against the dense kernel (d_feat bit for bit: same contraction order per pixel; d_w to summation order) on whole
  and ragged maps, every combination of wanted gradients, with and without clear_after;
* bts_render_bwd's flags: every texel it wrote lies in a flagged tile, on the gate-bit and on the row backward;
* the autograd route (ProjectFunction <- RenderFunction through the kept pair) against the dense route: same gradients, the pair is all
  zero after the step, and the constellations that must fall back (two renders of one map, retain_grad on G) do.
The oracle-anchored gradient tests (test_gpu_grad.py, test_gpu_train_step.py, test_gpu_scales.py) run through the sparse route too."""
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


def _spec(C, Hd, nb, blocks=False):
    from behindthescenes_amd.native import FieldSpec
    return FieldSpec(C=C, d_hidden=Hd, n_blocks=nb, tile_blocks=blocks)


@pytest.mark.parametrize("blocks", [False, True], ids=["runs64", "blocks16x4"])
@pytest.mark.parametrize("C,Hd,N,H,W", [(64, 64, 3, 24, 40), (64, 64, 2, 7, 13), (32, 32, 2, 16, 48), (32, 32, 1, 5, 31), (64, 64, 1, 192, 640)])
def test_tile_kernel_matches_the_dense_kernel(hip, C, Hd, N, H, W, blocks):
    """Both tile geometries (BtsFieldCfg.tile_blocks, ABI 9): runs of 64 consecutive texels and blocks of 4 rows x 16 texels -- the latter only
    where H is a multiple of 4 and W of 16 ((16, 48), (192, 640)); the other sizes fall back to runs, which proj_tile_map mirrors."""
    from behindthescenes_amd import native
    g = torch.Generator().manual_seed(C + H)
    spec = _spec(C, Hd, 0, blocks)
    feat = torch.randn(N, C, H, W, generator=g).cuda()
    mlp = torch.randn(spec.mlp_param_count(), generator=g).cuda()
    nt = native.proj_tile_count(spec, H, W)
    assert nt == (H * W + 63) // 64
    flags = (torch.rand(N, nt, generator=g) < 0.25)
    flags[0, -1] = True                                        # the (possibly ragged) last tile of an image
    texel_on = flags[:, native.proj_tile_map(H, W, blocks)].unsqueeze(-1)      # (N, H, W, 1): 16 x 4 blocks or 64 consecutive texels
    dG = (torch.randn(N, H, W, Hd, generator=g) * texel_on).cuda()
    tiles = flags.to(torch.uint8).cuda()
    ref_f, ref_w = native.project_features_bwd(spec, feat, dG, mlp)
    for need_feat, need_mlp in ((True, True), (True, False), (False, True)):
        buf, tl = dG.clone(), tiles.clone()
        d_f, d_w = native.project_features_bwd(spec, feat, buf, mlp, need_feat, need_mlp, tiles=tl, clear_after=False)
        assert torch.equal(buf, dG) and torch.equal(tl, tiles)
        if need_feat:
            assert torch.equal(d_f, ref_f)
        if need_mlp:
            assert (d_w - ref_w).abs().max().item() <= 1e-5 * ref_w.abs().max().item()
        d_f2, d_w2 = native.project_features_bwd(spec, feat, buf, mlp, need_feat, need_mlp, tiles=tl, clear_after=True)
        assert buf.abs().max().item() == 0.0 and tl.max().item() == 0
        if need_feat:
            assert torch.equal(d_f2, ref_f)
        if need_mlp:
            assert (d_w2 - ref_w).abs().max().item() <= 1e-5 * ref_w.abs().max().item()
    # nothing flagged: zeros out, nothing read (the buffer holds NaN where it must not be looked at)
    buf, tl = torch.full_like(dG, float("nan")), torch.zeros_like(tiles)
    d_f, d_w = native.project_features_bwd(spec, feat, buf, mlp, tiles=tl, clear_after=True)
    assert d_f.abs().max().item() == 0.0 and d_w.abs().max().item() == 0.0


def _scene_net(model, seed, train=True):
    from tests._hip_helpers import build_net
    re = model == "re10k"
    cfg = O.FieldConfig(d_min=1.0, d_max=100.0, code_mode="distance") if re else O.FieldConfig(learn_empty=True)
    C, Hd, nb = (32, 32, 1) if re else (64, 64, 0)
    n, v, H, W = 2, 3, 64, 160
    g = torch.Generator().manual_seed(seed)
    scene = O.synthetic_scene(n, v, H, W, C, seed=seed, intrinsics=O.K_RE10K if re else O.K_KITTI360, smooth=True)
    mlp = O.init_mlp(C + 39, Hd, nb, gen=g)
    empty = torch.randn(C, generator=g) if cfg.learn_empty else None
    net = build_net(cfg, mlp, scene, [1, 2], empty_feature=empty, train=train)
    return scene, net, g, cfg, (48 if re else 64)


def _patch_rays(scene, cfg, H, W, n_patches, g):
    """8 x 8 patches like PatchRaySampler's: (n, n_patches * 64, 8)"""
    n = scene["poses"].shape[0]
    rays = O.image_rays(scene["poses"], scene["projs"], H, W, cfg.d_min, cfg.d_max).reshape(n, -1, H, W, 8)[:, 1]   # from view 1
    out = []
    for b in range(n):
        ys = torch.randint(0, H - 8, (n_patches,), generator=g)
        xs = torch.randint(0, W - 8, (n_patches,), generator=g)
        out.append(torch.stack([rays[b, y:y + 8, x:x + 8].reshape(64, 8) for y, x in zip(ys.tolist(), xs.tolist())]).reshape(-1, 8))
    return torch.stack(out).contiguous()


@pytest.mark.parametrize("model", ["kitti", "re10k"])
def test_render_bwd_flags_every_tile_it_writes(hip, model):
    from behindthescenes_amd import native
    scene, net, g, cfg, K = _scene_net(model, seed=7)
    ft, params = net.native_field(), net.mlp_coarse.packed().detach()
    rays = _patch_rays(scene, cfg, 64, 160, 6, g).reshape(-1, 8).cuda()
    z = native.sample_coarse(rays, torch.rand(rays.shape[0], K, generator=g).cuda(), True)
    out = native.render_fwd(ft, params, rays, z, hard_alpha_cap=model == "kitti", want_saved=True, want_rgb_samps=True)
    g_rgb, g_depth = torch.randn(out["rgb"].shape, generator=g).cuda(), torch.randn(out["depth"].shape, generator=g).cuda()
    kw = dict(hard_alpha_cap=model == "kitti", g_rgb=g_rgb, g_depth=g_depth, rgb_samps=out["rgb_samps"], need_empty=cfg.learn_empty)
    dense, dm0, de0 = native.render_bwd(ft, params, rays, z, out["sigma_raw"], out["trans"], **kw)
    buf = torch.zeros_like(dense)
    tiles = torch.zeros((2, native.proj_tile_count(ft.spec, 64, 160)), dtype=torch.uint8, device="cuda")
    got, dm1, de1 = native.render_bwd(ft, params, rays, z, out["sigma_raw"], out["trans"], proj_grad=(buf, tiles), **kw)
    assert got.data_ptr() == buf.data_ptr()
    scale = dense.abs().max().item()
    assert (got - dense).abs().max().item() <= 2e-5 * scale                  # float atomics: summation order only
    assert (dm1 - dm0).abs().max().item() <= 2e-5 * dm0.abs().max().item()
    texel_written = (got != 0).any(dim=-1).reshape(2, -1)
    flagged = tiles.bool()[:, native.proj_tile_map(64, 160, ft.spec.tile_blocks).cuda()].reshape(2, -1)
    assert not (texel_written & ~flagged).any(), "a texel outside the flagged tiles received a contribution"
    frac = tiles.float().mean().item()
    assert 0.0 < frac < 0.9, frac
    # and the projection's backward over exactly these flags equals the dense one on the dense gradient
    feat = net.encoder.feats[0].detach()
    feat = feat.reshape(feat.shape[0], *feat.shape[-3:]).contiguous()
    ref_f, ref_w = native.project_features_bwd(ft.spec, feat, dense, params)
    d_f, d_w = native.project_features_bwd(ft.spec, feat, buf, params, tiles=tiles, clear_after=True)
    assert (d_f - ref_f).abs().max().item() <= 2e-5 * ref_f.abs().max().item()
    assert (d_w - ref_w).abs().max().item() <= 2e-5 * ref_w.abs().max().item()
    assert buf.abs().max().item() == 0.0 and tiles.max().item() == 0


def _step(hip, net, scene, rays, coef, K, renders=1, retain=False):
    """encode -> `renders` composites of the same map -> scalar -> backward; returns the gradients."""
    from behindthescenes_amd import native
    net.zero_grad(set_to_none=True)
    net.encode(scene["images"].cuda(), scene["projs"].cuda(), scene["poses"].cuda(), ids_encoder=[0], ids_render=[1, 2])
    renderer = hip.NeRFRenderer.from_conf(dict(n_coarse=K, lindisp=True, hard_alpha_cap=K == 64)).cuda().train()
    n = rays.shape[0]
    gz = torch.Generator().manual_seed(3)
    total = 0.0
    for r in range(renders):
        z = native.sample_coarse(rays.reshape(-1, 8), torch.rand(n * rays.shape[1], K, generator=gz).cuda(), True)
        w, rgb, depth, *_ = renderer.composite(net, rays.reshape(-1, 8), z, sb=n)
        total = total + (rgb.reshape(coef.shape) * coef).sum() * (r + 1) + 0.05 * depth.sum()
    if retain:
        net.native_field().proj_nhwc.retain_grad()
    total.backward()
    ps = [net.mlp_coarse.lin_in.weight, net.mlp_coarse.lin_in.bias, net.mlp_coarse.lin_out.weight, net.encoder.feats[0]]
    if net.learn_empty:
        ps.append(net.empty_feature)
    return [p.grad.clone() for p in ps], (net.native_field().proj_nhwc.grad if retain else None)


@pytest.mark.parametrize("model", ["kitti", "re10k"])
def test_autograd_route_through_the_kept_pair(hip, model):
    from behindthescenes_amd import native
    scene, net, g, cfg, K = _scene_net(model, seed=11)
    rays = _patch_rays(scene, cfg, 64, 160, 5, g).cuda()
    coef = torch.randn(2, rays.shape[1], 6, generator=g).cuda()
    native.release_sparse_grads()

    def close(a, b):
        for x, y in zip(a, b):
            assert (x - y).abs().max().item() <= 3e-5 * y.abs().max().item()

    native.SPARSE_PROJ_GRAD = False
    try:
        dense, _ = _step(hip, net, scene, rays, coef, K)
        assert not native._SPARSE
        dense2, _ = _step(hip, net, scene, rays, coef, K, renders=2)
    finally:
        native.SPARSE_PROJ_GRAD = True
    for rep in range(3):                                                   # the pair is reused step after step
        sparse, _ = _step(hip, net, scene, rays, coef, K)
        close(sparse, dense)
        assert len(native._SPARSE) == 1
        (e,) = native._SPARSE.values()
        assert not e.busy and e.buf.abs().max().item() == 0.0 and e.tiles.max().item() == 0
    # two renders of one map: autograd sums their gradients -- the dense route, the pair untouched
    two, _ = _step(hip, net, scene, rays, coef, K, renders=2)
    close(two, dense2)
    assert not e.busy and e.buf.abs().max().item() == 0.0 and e.tiles.max().item() == 0
    # retain_grad on G: the caller keeps the gradient -- it must not be the pair's buffer
    kept, gG = _step(hip, net, scene, rays, coef, K, retain=True)
    close(kept, dense)
    assert gG is not None and gG.data_ptr() != e.buf.data_ptr() and gG.abs().max().item() > 0
    native.release_sparse_grads()


@pytest.mark.parametrize("model", ["kitti", "re10k"])
def test_sparse_forward_projection_covers_every_tap(hip, model):
    """bts_mark_sampled_tiles + bts_project_features_tiles: the map built for one render's samples.  Every texel the render reads must
    be computed -- shown the hard way: the unflagged part of the map is NaN, and the render (forward, from the jitter and from injected
    depths, and backward) must equal the dense map's bit for bit."""
    from behindthescenes_amd import native
    scene, net, g, cfg, K = _scene_net(model, seed=23)
    ft, params = net.native_field(), net.mlp_coarse.packed().detach()
    rays = _patch_rays(scene, cfg, 64, 160, 6, g).reshape(-1, 8).cuda()
    u = torch.rand(rays.shape[0], K, generator=g).cuda()
    z = native.sample_coarse(rays, u, True)
    feat = net.encoder.feats[0].detach()
    feat = feat.reshape(feat.shape[0], *feat.shape[-3:]).contiguous()
    kw = dict(hard_alpha_cap=model == "kitti", want_saved=True, want_rgb_samps=True, want_weights=True)
    for mode in ("jitter", "z"):
        tiles = native.mark_sampled_tiles(ft.spec, 2, 64, 160, 0, ft.K_enc, ft.w2c_enc, rays, z if mode == "z" else None, u if mode == "jitter" else None, True)
        frac = tiles.float().mean().item()
        assert 0.0 < frac < 0.9, frac
        G = native.project_features(ft.spec, feat, params, tiles=tiles)
        flagged = tiles.bool()[:, native.proj_tile_map(64, 160, ft.spec.tile_blocks).cuda()]
        assert torch.equal(G[flagged], ft.proj_nhwc.detach()[flagged])                 # the flagged tiles are the dense map's
        G = torch.where(flagged.unsqueeze(-1), G, torch.full_like(G, float("nan")))      # everything else must never be read
        sp = native.FieldTensors(ft.spec, G, ft.K_enc, ft.w2c_enc, ft.imgs_nhwc4, ft.K_r, ft.w2c_r, ft.empty_feature, enc_view=ft.enc_view)
        a = native.render_fwd(ft, params, rays, z if mode == "z" else None, jitter=u if mode == "jitter" else None, **kw)
        b = native.render_fwd(sp, params, rays, z if mode == "z" else None, jitter=u if mode == "jitter" else None, **kw)
        for k_ in ("rgb", "depth", "weights", "sigma_raw", "trans", "rgb_samps"):
            assert torch.equal(a[k_], b[k_]), (mode, k_)
    g_rgb, g_depth = torch.randn(a["rgb"].shape, generator=g).cuda(), torch.randn(a["depth"].shape, generator=g).cuda()
    bk = dict(hard_alpha_cap=model == "kitti", g_rgb=g_rgb, g_depth=g_depth, rgb_samps=a["rgb_samps"], need_empty=cfg.learn_empty)
    ga = native.render_bwd(ft, params, rays, z, a["sigma_raw"], a["trans"], **bk)
    gb = native.render_bwd(sp, params, rays, z, a["sigma_raw"], a["trans"], **bk)
    for x, y in zip(ga, gb):
        if x is not None:
            assert torch.isfinite(y).all() and (x - y).abs().max().item() <= 2e-5 * x.abs().max().item()


def test_lean_training_step_with_and_without_the_sparse_projection(hip):
    """renderer(...) in training mode with lean outputs: the sparse projection (default) against the dense one -- same outputs bit for
    bit, same gradients to summation order; multi-scale maps (feat_shift) included."""
    import torch.nn.functional as F
    from tests._hip_helpers import make_conf, load_mlp
    n, v, H, W, C = 2, 3, 64, 160, 64
    cfg = O.FieldConfig(learn_empty=True)
    g = torch.Generator().manual_seed(31)
    scene = O.synthetic_scene(n, v, H, W, C, seed=31, intrinsics=O.K_KITTI360, smooth=True)
    conf = make_conf(cfg, C, 64, 0, H, W)
    conf["encoder"].update(n_scales=3, pyramid=True, num_views=n)
    net = hip.BTSNet(conf)
    load_mlp(net, O.init_mlp(C + 39, 64, 0, gen=g))
    with torch.no_grad():
        for s, dst in enumerate(net.encoder.feats):
            dst.copy_(F.avg_pool2d(torch.randn(n, C, H >> s, W >> s, generator=g), 3, 1, 1) * 2)
        net.empty_feature.copy_(torch.randn(C, generator=g))
    net = net.cuda().train()
    renderer = hip.NeRFRenderer.from_conf(dict(n_coarse=64, lindisp=True, hard_alpha_cap=True, lean_training_outputs=True)).cuda().train()
    rays = _patch_rays(scene, cfg, H, W, 5, g).cuda()
    coef = torch.randn(n, rays.shape[1], 6, generator=g).cuda()
    res = {}
    for sparse in (True, False):
        renderer.sparse_projection = sparse
        net.zero_grad(set_to_none=True)
        net.encode(scene["images"].cuda(), scene["projs"].cuda(), scene["poses"].cuda(), ids_encoder=[0], ids_render=[1, 2])
        outs, total = [], 0.0
        for scale in (0, 1, 2):
            net.set_scale(scale)
            torch.manual_seed(100 + scale)                                   # the same jitter in both runs
            o = renderer.bind_parallel(net)(rays)["coarse"]
            outs.append(o)
            total = total + (o["rgb"] * coef).sum() + 0.05 * o["depth"].sum()
        net.set_scale(0)
        total.backward()
        grads = [p.grad.clone() for p in (net.mlp_coarse.lin_in.weight, net.mlp_coarse.lin_out.weight, net.empty_feature, *net.encoder.feats)]
        res[sparse] = (outs, grads)
    for a, b in zip(res[True][0], res[False][0]):
        for k_ in ("rgb", "depth", "invalid_wsum", "invalid_any"):
            assert torch.equal(a[k_], b[k_]), k_
    for a, b in zip(res[True][1], res[False][1]):
        assert torch.isfinite(a).all() and (a - b).abs().max().item() <= 3e-5 * b.abs().max().item()
