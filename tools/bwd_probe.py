"""Times bts_render_bwd alone on the training shape with parts switched off (no dG atomics / no weight gradients)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import behindthescenes_amd as bts
from behindthescenes_amd import native
from behindthescenes_amd import synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H, W, K, C, HD, V = 192, 640, 64, 64, 64, 8
Z_NEAR, Z_FAR = 3.0, 80.0
scene = S.synthetic_scene(n, V, H, W, C, seed=5, intrinsics=S.K_KITTI360, smooth=True)
net = bts.BTSNet(S.field_conf(C, HD, 0, H, W)); S.init_mlp_(net.mlp_coarse, seed=7)
net.encoder = bts.FeatureMapEncoder((H, W), C, num_views=n)
with torch.no_grad():
    net.encoder.feats[0].data = scene["feat"].clone()
net = net.cuda().eval()
images, projs, poses = scene["images"].cuda(), scene["projs"].cuda(), scene["poses"].cuda()
with torch.no_grad():
    net.encode(images, projs, poses, ids_encoder=[0], ids_render=[4, 5, 6, 7], images_alt=images * .5 + .5)
    sampler = bts.PatchRaySampler(ray_batch_size=4096, z_near=Z_NEAR, z_far=Z_FAR, patch_size=8)
    rays, _ = sampler.sample(images[:, :4] * .5 + .5, poses[:, :4], projs[:, :4])
    rays = rays.reshape(-1, 8).contiguous()
    z = native.sample_coarse(rays, torch.rand(rays.shape[0], K, device="cuda"), True)
    ft = net.native_field()
    params = net.mlp_coarse.packed().detach()
    out = native.render_fwd(ft, params, rays, z, hard_alpha_cap=True, want_saved=True)
    g_rgb = torch.randn_like(out["rgb"]); g_depth = torch.randn_like(out["depth"]) * 0.1
    for name, kw in (("all", dict()), ("no dG", dict(need_proj=False)), ("no d_mlp", dict(need_mlp=False)), ("neither", dict(need_proj=False, need_mlp=False))):
        ts = []
        for r in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            native.render_bwd(ft, params, rays, z, out["sigma_raw"], out["trans"], hard_alpha_cap=True, g_rgb=g_rgb, g_depth=g_depth, **kw)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f"render_bwd {name:10s}: {sorted(ts)[1]:.3f} ms  ({rays.shape[0]} rays)")
