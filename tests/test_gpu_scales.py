"""GPU: the decoder's coarser scales handed over at THEIR OWN size (BtsFieldCfg.feat_shift, ABI 4).

BTSNet.encode resizes every scale's feature map to scale 0's size with F.interpolate(mode="nearest") (models_bts.py:115-117) and
the renderer samples the resized map.  A nearest resize by 2^s repeats texels, so the kernels index the small map at
(y >> s, x >> s): same taps, same weights, same blend order -- the forward must be BIT-identical to the resized map's, the gradient
w.r.t. the small map the resize's backward (the sum over each texel's 4^s copies) up to summation order.  The oracle-anchored check of
the same path is tests/test_gpu_grad.py::test_multiscale_render_and_backward_vs_oracle (oracle on the resized map)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import bts_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import behindthescenes_amd as bts
    from behindthescenes_amd import _lib
    assert torch.cuda.is_available()
    _lib.load()
    return bts


SHAPES = {  # name: (n, v, H, W, C, Hd, nb, K, ids_render, hard cap, field config)
    "kitti360": (2, 5, 64, 160, 64, 64, 0, 64, [1, 2, 3, 4], True, O.FieldConfig(learn_empty=True)),                # gate-bit backward
    "re10k_k48": (2, 3, 64, 96, 32, 32, 1, 48, [1, 2], False, O.FieldConfig(d_min=1.0, d_max=100.0, code_mode="distance")),   # row backward
    "re10k_k128": (1, 3, 64, 96, 32, 32, 1, 128, [1, 2], False, O.FieldConfig(d_min=1.0, d_max=100.0, code_mode="distance")),  # K > 64
}


def _pair(hip, name, seed):
    """Two nets with the same weights and the same four-scale pyramid: one hands the scales over at their own size, the other
    materialises the resized maps like the reference."""
    from tests._hip_helpers import make_conf, load_mlp
    n, v, H, W, C, Hd, nb, K, ids, cap, cfg = SHAPES[name]
    g = torch.Generator().manual_seed(seed)
    scene = O.synthetic_scene(n, v, H, W, C, seed=seed, intrinsics=O.K_RE10K if "re10k" in name else O.K_KITTI360,
                              baseline=0.2 if "re10k" in name else 0.6, smooth=True)
    mlp = O.init_mlp(C + 39, Hd, nb, gen=g)
    feats = [F.avg_pool2d(torch.randn(n, C, H >> s, W >> s, generator=g), 3, 1, 1) * 2 for s in range(4)]
    nets = []
    for native_maps in (True, False):
        conf = make_conf(cfg, C, Hd, nb, H, W)
        conf["encoder"].update(n_scales=4, pyramid=True, num_views=n)
        conf["native_scale_maps"] = native_maps
        net = hip.BTSNet(conf)
        load_mlp(net, mlp)
        with torch.no_grad():
            for dst, src in zip(net.encoder.feats, feats):
                dst.copy_(src)
            if cfg.learn_empty:
                net.empty_feature.copy_(torch.randn(C, generator=torch.Generator().manual_seed(seed + 1)))
        net = net.cuda().train()
        net.encode(scene["images"].cuda(), scene["projs"].cuda(), scene["poses"].cuda(), ids_encoder=[0], ids_render=ids)
        nets.append(net)
    assert nets[0]._shift_ms == [0, 1, 2, 3] and nets[1]._shift_ms == [0, 0, 0, 0]
    return nets, scene, g


@pytest.mark.parametrize("name", list(SHAPES))
def test_scale_maps_at_their_own_size_render_the_resized_maps_bits(hip, name):
    n, v, H, W, C, Hd, nb, K, ids, cap, cfg = SHAPES[name]
    (net_s, net_r), scene, g = _pair(hip, name, seed=17)
    renderer = hip.NeRFRenderer(n_coarse=K, lindisp=True, hard_alpha_cap=cap).cuda()
    sampler = hip.PatchRaySampler(ray_batch_size=512, z_near=cfg.d_min, z_far=cfg.d_max, patch_size=8)
    torch.manual_seed(3)
    rays, _ = sampler.sample(scene["images"][:, :1].cuda() * .5 + .5, scene["poses"][:, :1].cuda(), scene["projs"][:, :1].cuda())
    rays = rays.reshape(-1, 8)
    z = renderer.sample_coarse(rays, torch.rand(rays.shape[0], K, device="cuda"))
    c_rgb = torch.randn(rays.shape[0], 3 * len(ids), device="cuda")
    for s in range(4):
        outs, grads = [], []
        for net in (net_s, net_r):
            net.set_scale(s)
            ft = net.native_field()
            assert ft.feat_shift == (s if net is net_s else 0)
            assert tuple(ft.proj_nhwc.shape[1:3]) == ((H >> s, W >> s) if net is net_s else (H, W))
            net.zero_grad(set_to_none=True)
            w, rgb, depth, a, inv, *_ = renderer.composite(net, rays, z, sb=n)
            outs.append((w.detach(), rgb.detach(), depth.detach(), a.detach(), inv.detach()))
            ((rgb * c_rgb).sum() + 0.05 * depth.sum() + 0.01 * (w * w).sum()).backward()
            mc = net.mlp_coarse
            grads.append([p.grad.clone() for p in mc.parameters()] + [net.encoder.feats[s].grad.clone()] +
                         ([net.empty_feature.grad.clone()] if cfg.learn_empty else []))
        for i, (a_, b_) in enumerate(zip(*outs)):
            assert torch.equal(a_, b_), f"scale {s}, output {i}: {(a_ != b_).sum().item()} of {a_.numel()} values differ"
        for i, (a_, b_) in enumerate(zip(*grads)):
            # same per-sample gradient rows; the atomics into one small texel arrive in another order than the resize's backward sums
            # its 4^s texels: 1e-5 of the largest entry
            err = (a_ - b_).abs().max().item() / max(b_.abs().max().item(), 1e-20)
            assert err <= 2e-5, (s, i, err)


def test_query_and_occupancy_profile_on_a_small_scale_map(hip):
    """bts_field_query / bts_occupancy_profile read the same maps: bit-identical between the two hand-overs at every scale."""
    (net_s, net_r), scene, g = _pair(hip, "kitti360", seed=23)
    net_s.eval(), net_r.eval()
    q = O.profile_points(x_range=(-6, 6), y_range=(0, .75), z_range=(14, 3), x_res=64, y_res=64, z_res=32)
    Y = q.shape[0]
    pts = q.reshape(1, -1, 3).expand(2, -1, -1).contiguous().cuda()
    for s in (1, 3):
        res = []
        for net in (net_s, net_r):
            net.set_scale(s)
            with torch.no_grad():
                rgb, inv, sig = net(pts)
                prof, sig_p = net.occupancy_profile(pts, Y, want_sigma=True)
            res.append((rgb, inv, sig, prof, sig_p))
        for i, (a_, b_) in enumerate(zip(*res)):
            assert torch.equal(a_, b_), (s, i)


def test_feat_shift_needs_frame_sizes_that_are_multiples(hip):
    from behindthescenes_amd import native
    spec = native.FieldSpec(C=64, d_hidden=64, n_blocks=0)
    n, H, W = 1, 36, 100                                   # 100 is not a multiple of 8
    proj = torch.zeros(n, H >> 3, W >> 3, 64, device="cuda")
    K_enc, w2c = torch.eye(3, device="cuda")[None].contiguous(), torch.eye(4, device="cuda")[None].contiguous()
    ft = native.FieldTensors(spec, proj, K_enc, w2c, None, None, None, feat_shift=3)
    ft.H, ft.W = H, W
    rays = torch.zeros(64, 8, device="cuda")
    with pytest.raises(native.BtsNativeError, match="feat_shift"):
        native.render_fwd(ft, torch.zeros(spec.mlp_param_count(), device="cuda"), rays, torch.ones(64, 8, device="cuda"), hard_alpha_cap=True)
