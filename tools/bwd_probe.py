"""Times bts_render_bwd alone on the BASELINE configs[2] shape (bs 16, 4096 patch rays x 64 samples, nv = 4), with and without the
forward's per-sample colours handed to it.   python tools/bwd_probe.py [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import behindthescenes_amd as bts
from behindthescenes_amd import native, synthetic as S

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n, V, H, W, C, K = 16, 8, 192, 640, 64, 64
scene = S.synthetic_scene(n, V, H, W, C, seed=5, intrinsics=S.K_KITTI360, smooth=True)
net = bts.BTSNet(S.field_conf(C, 64, 0, H, W)); S.init_mlp_(net.mlp_coarse, seed=7)
net.encoder = bts.FeatureMapEncoder((H, W), C, num_views=n)
with torch.no_grad():
    net.encoder.feats[0].data = scene["feat"].clone()
net = net.cuda().eval()
images, projs, poses = scene["images"].cuda(), scene["projs"].cuda(), scene["poses"].cuda()
with torch.no_grad():
    net.encode(images, projs, poses, ids_encoder=[0], ids_render=[4, 5, 6, 7])
    ft = net.native_field()
    sampler = bts.PatchRaySampler(ray_batch_size=4096, z_near=3.0, z_far=80.0, patch_size=8)
    rays, _ = sampler.sample(images[:, :4] * .5 + .5, poses[:, :4], projs[:, :4])
    rays = rays.reshape(-1, 8).contiguous()
    z = native.sample_coarse(rays, torch.rand(rays.shape[0], K, device="cuda"), True)
    params = net.mlp_coarse.packed().detach()
    out = native.render_fwd(ft, params, rays, z, hard_alpha_cap=True, want_rgb_samps=True, want_saved=True)
    g_rgb, g_depth = torch.randn_like(out["rgb"]), torch.randn_like(out["depth"]) * 0.1
    for name, rs in (("colours recomputed", None), ("colours from the forward", out["rgb_samps"])):
        ts = []
        for r in range(rounds + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            native.render_bwd(ft, params, rays, z, out["sigma_raw"], out["trans"], hard_alpha_cap=True, g_rgb=g_rgb, g_depth=g_depth, rgb_samps=rs)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts = sorted(ts[1:])
        print(f"bts_render_bwd, {name}: median {ts[len(ts) // 2]:.3f} ms  min {ts[0]:.3f} ms   ({rays.shape[0]} rays x {K})")
