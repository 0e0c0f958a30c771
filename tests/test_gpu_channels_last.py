"""GPU: the encoder's map handed over channels-last (ABI 8: bts_project_features_cl / _bwd_cl, BtsTrainScale.feat_channels_last).

A (N, C, H, W) tensor in torch's channels_last format is (N, H, W, C) in memory -- what MIOpen's NHWC convolutions and bts_conv3x3_fwd
write.  The entry points read / write that memory directly.  The products are the NCHW kernels'; what differs is the ORDER of the fp32
sums of the forward (a lane's float4 pairs channel 8 q + e with 8 q + 4 + e in a k-step, the NCHW rows pair 2 s with 2 s + 1): G agrees to
fp32 rounding, the feature gradient of a given dG bit for bit (same pairs, same order), the weight gradient to the order of its atomics."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _spec(C, Hd, blocks=False):
    from behindthescenes_amd import native
    return native.FieldSpec(C=C, d_hidden=Hd, n_blocks=0 if C == 64 else 1, tile_blocks=blocks)


@pytest.mark.parametrize("blocks", [False, True], ids=["runs64", "blocks16x4"])
@pytest.mark.parametrize("C,Hd,N,H,W", [(64, 64, 2, 24, 40), (64, 64, 1, 7, 13), (32, 32, 2, 16, 48), (32, 32, 1, 5, 31), (64, 64, 1, 192, 640)])
def test_projection_of_a_channels_last_map_equals_the_nchw_one(C, Hd, N, H, W, blocks):
    from behindthescenes_amd import native
    g = torch.Generator().manual_seed(C + H)
    spec = _spec(C, Hd, blocks)
    feat = torch.randn(N, C, H, W, generator=g).cuda()
    feat_cl = feat.contiguous(memory_format=torch.channels_last)
    assert native.is_channels_last(feat_cl) and not native.is_channels_last(feat)
    mlp = torch.randn(spec.mlp_param_count(), generator=g).cuda()
    nt = native.proj_tile_count(spec, H, W)
    flags = torch.rand(N, nt, generator=g) < 0.3
    flags[0, -1] = True
    tiles = flags.to(torch.uint8).cuda()
    # forward: dense and flagged tiles
    G = native.project_features(spec, feat, mlp)
    G_cl = native.project_features(spec, feat_cl, mlp)
    tol = 2e-6 * G.abs().max().item()                  # (64 products per output, summed in a different order)
    assert (G_cl - G).abs().max().item() <= tol
    Gt, Gt_cl = native.project_features(spec, feat, mlp, tiles), native.project_features(spec, feat_cl, mlp, tiles)
    on = flags[:, native.proj_tile_map(H, W, blocks)].cuda()
    assert torch.equal(Gt_cl[on], G_cl[on]) and torch.equal(Gt[on], G[on])
    # backward: dense, then the tile form with and without clearing
    texel_on = on.unsqueeze(-1)
    dG = torch.randn(N, H, W, Hd, generator=g).cuda()
    ref_f, ref_w = native.project_features_bwd(spec, feat, dG, mlp)
    d_f, d_w = native.project_features_bwd(spec, feat_cl, dG, mlp)
    assert native.is_channels_last(d_f) and torch.equal(d_f, ref_f)
    assert (d_w - ref_w).abs().max().item() <= 1e-5 * ref_w.abs().max().item()
    dGs = dG * texel_on
    ref_f, ref_w = native.project_features_bwd(spec, feat, dGs, mlp)
    for need_feat, need_mlp in ((True, True), (True, False), (False, True)):
        buf, tl = dGs.clone(), tiles.clone()
        d_f, d_w = native.project_features_bwd(spec, feat_cl, buf, mlp, need_feat, need_mlp, tiles=tl, clear_after=False)
        assert torch.equal(buf, dGs) and torch.equal(tl, tiles)
        if need_feat:
            assert native.is_channels_last(d_f) and torch.equal(d_f, ref_f)
        if need_mlp:
            assert (d_w - ref_w).abs().max().item() <= 1e-5 * ref_w.abs().max().item()
        d_f2, d_w2 = native.project_features_bwd(spec, feat_cl, buf, mlp, need_feat, need_mlp, tiles=tl, clear_after=True)
        assert buf.abs().max().item() == 0.0 and tl.max().item() == 0
        if need_feat:
            assert torch.equal(d_f2, ref_f)
    # nothing flagged: zeros out, nothing read
    buf, tl = torch.full_like(dG, float("nan")), torch.zeros_like(tiles)
    d_f, d_w = native.project_features_bwd(spec, feat_cl, buf, mlp, tiles=tl, clear_after=True)
    assert d_f.abs().max().item() == 0.0 and d_w.abs().max().item() == 0.0


@pytest.mark.parametrize("multiscale", [False, True])
def test_fused_train_step_on_channels_last_maps_equals_the_nchw_step(multiscale):
    """FusedTrainStep with the stand-in encoder's maps in channels_last format against the NCHW run (same draws: the generators are
    re-seeded): G differs by fp32 rounding (see above), so the loss, the outputs and the gradients agree to 1e-4 of their largest entry."""
    import behindthescenes_amd as bts
    from behindthescenes_amd import synthetic as S
    n, V, H, W, C = 2, 3, 64, 96, 64
    scene = S.synthetic_scene(n, V, H, W, C, seed=4, intrinsics=S.K_KITTIRAW, smooth=True)
    res = []
    for cl in (False, True):
        torch.manual_seed(11)
        n_scales = 3 if multiscale else 1
        net = bts.BTSNet(S.field_conf(C, 64, 0, H, W))
        net.encoder = bts.FeatureMapEncoder((H, W), C, num_views=n, n_scales=n_scales, pyramid=multiscale, channels_last=cl)
        with torch.no_grad():
            for s, p in enumerate(net.encoder.feats):
                p.copy_(torch.nn.functional.avg_pool2d(scene["feat"], 2 ** s) if s else scene["feat"])
        S.init_mlp_(net.mlp_coarse, seed=7)
        net = net.cuda().train()
        assert all(bts.native.is_channels_last(p) == cl for p in net.encoder.feats)
        renderer = bts.NeRFRenderer.from_conf(dict(n_coarse=32, lindisp=True, hard_alpha_cap=True, lean_training_outputs=True)).cuda().train()
        sampler = bts.PatchRaySampler(ray_batch_size=512, z_near=3.0, z_far=80.0, patch_size=8)
        crit = bts.ReconstructionLoss({"criterion": "l1+ssim", "invalid_policy": "weight_guided", "lambda_edge_aware_smoothness": 0.001})
        step = bts.FusedTrainStep(renderer.bind_parallel(net).train(), sampler, crit, multiscale=multiscale)
        images, projs, poses = scene["images"].cuda(), scene["projs"].cuda(), scene["poses"].cuda()
        torch.manual_seed(5), torch.cuda.manual_seed(5)
        loss, loss_dict, data = step(images, projs, poses, ids_encoder=[0], ids_render=[1, 2], ids_loss=[0, 1])
        assert step.last_path == "fused"
        loss.backward()
        grads = [p.grad for p in net.encoder.feats]
        assert all(bts.native.is_channels_last(g_) == cl for g_ in grads)
        res.append((loss.detach().clone(), [lv["rgb"].clone() for lv in data["coarse"]], [g_.contiguous() for g_ in grads],
                    torch.cat([p.grad.reshape(-1) for p in net.mlp_coarse.parameters()])))
    (l0, o0, g0, m0), (l1, o1, g1, m1) = res

    def close(a, b, rel):
        return (a - b).abs().max().item() <= rel * max(a.abs().max().item(), 1e-30)
    assert close(l0, l1, 1e-5) and all(close(a, b, 1e-4) for a, b in zip(o0, o1))
    assert all(close(a, b, 1e-4) for a, b in zip(g0, g1)) and close(m0, m1, 1e-4)
