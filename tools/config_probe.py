"""Forward / backward kernel times on the other BASELINE shapes: RE10K (C=32, d_hidden=32, one ResnetBlockFC, 256x384, nv=2,
code_mode distance, no hard cap) and KITTI K=32 (configs[0]).   python tools/config_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import behindthescenes_amd as bts
from behindthescenes_amd import native
from behindthescenes_amd import synthetic as S


def run(name, n, V, H, W, C, HD, NB, K, ids_render, intr, z_near, z_far, hard_cap, patch_rays=None, **conf):
    scene = S.synthetic_scene(n, V, H, W, C, seed=3, intrinsics=intr, smooth=True)
    net = bts.BTSNet(S.field_conf(C, HD, NB, H, W, z_near=z_near, z_far=z_far, **conf))
    S.init_mlp_(net.mlp_coarse, seed=7)
    net.encoder = bts.FeatureMapEncoder((H, W), C, num_views=n)
    S.set_feature_map(net, scene["feat"])
    net = net.cuda().train()
    images, projs, poses = scene["images"].cuda(), scene["projs"].cuda(), scene["poses"].cuda()
    net.encode(images, projs, poses, ids_encoder=[0], ids_render=ids_render, images_alt=images * .5 + .5)
    if patch_rays:
        sampler = bts.PatchRaySampler(ray_batch_size=patch_rays, z_near=z_near, z_far=z_far, patch_size=8)
        loss_ids = [i for i in range(V) if i not in ids_render] or [0]
        rays, _ = sampler.sample(images[:, loss_ids] * .5 + .5, poses[:, loss_ids], projs[:, loss_ids])
    else:
        rays, _ = bts.ImageRaySampler(z_near, z_far, H, W).sample(None, poses, projs)
    rays = rays.reshape(-1, 8).contiguous()
    z = native.sample_coarse(rays, torch.rand(rays.shape[0], K, device="cuda"), True)
    ft = net.native_field()
    params = net.mlp_coarse.packed().detach()
    ev = lambda: torch.cuda.Event(enable_timing=True)
    tf, tb = [], []
    for r in range(4):
        a, b, c = ev(), ev(), ev()
        a.record()
        out = native.render_fwd(ft, params, rays, z, hard_alpha_cap=hard_cap, want_weights=True, want_alphas=True, want_invalid=True,
                                want_saved=bool(patch_rays))
        b.record()
        if patch_rays:
            native.render_bwd(ft, params, rays, z, out["sigma_raw"], out["trans"], hard_alpha_cap=hard_cap,
                              g_rgb=torch.randn_like(out["rgb"]), g_depth=torch.randn_like(out["depth"]) * 0.1)
        c.record()
        torch.cuda.synchronize()
        tf.append(a.elapsed_time(b)), tb.append(b.elapsed_time(c))
    f, bw = sorted(tf)[1], sorted(tb)[1]
    B = rays.shape[0]
    print(f"{name:34s} {B:7d} rays x {K:3d}: fwd {f:7.3f} ms ({B / f / 1e3:7.1f} M rays/s)" + (f"   bwd {bw:7.3f} ms" if patch_rays else ""))


run("KITTI K=32 eval (configs[0])", 1, 1, 192, 640, 64, 64, 0, 32, [0], S.K_KITTIRAW, 3.0, 80.0, True)
run("KITTI K=64 eval (configs[1])", 1, 2, 192, 640, 64, 64, 0, 64, [0], S.K_KITTIRAW, 3.0, 80.0, True)
run("RE10K eval 256x384 K=64", 1, 3, 256, 384, 32, 32, 1, 64, [1, 2], S.K_RE10K, 1.0, 100.0, False, code_mode="distance")
run("RE10K train bs24 1024 rays K=48", 24, 3, 256, 384, 32, 32, 1, 48, [1, 2], S.K_RE10K, 1.0, 100.0, False, patch_rays=1024, code_mode="distance")
run("KITTI-360 train bs16 4096 rays K=64", 16, 8, 192, 640, 64, 64, 0, 64, [4, 5, 6, 7], S.K_KITTI360, 3.0, 80.0, True, patch_rays=4096)
