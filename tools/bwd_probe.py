"""Times bts_render_bwd alone on a BASELINE training shape, with and without the forward's per-sample colours handed to it.
    python tools/bwd_probe.py [rounds] [kitti360 | kitti_raw | re10k] [K]
kitti360 (default): configs[2], bs 16, 4096 patch rays x 64 samples, nv = 4 (gate-bit passes); kitti_raw: configs[3], bs 8, 2048 rays,
nv = 2; re10k: configs[4], bs 24, 1024 rays x 48 (or K) samples, C = 32, one ResnetBlockFC, distance code (row passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import behindthescenes_amd as bts
from behindthescenes_amd import native, synthetic as S

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
shape = sys.argv[2] if len(sys.argv) > 2 else "kitti360"
SH = {"kitti360": dict(n=16, V=8, H=192, W=640, C=64, NB=0, K=64, loss=4, render=[4, 5, 6, 7], rays=4096, intr=S.K_KITTI360, z=(3.0, 80.0), cap=True, code="z"),
      "kitti_raw": dict(n=8, V=4, H=192, W=640, C=64, NB=0, K=64, loss=2, render=[2, 3], rays=2048, intr=S.K_KITTIRAW, z=(3.0, 80.0), cap=True, code="z"),
      "re10k": dict(n=24, V=3, H=256, W=384, C=32, NB=1, K=48, loss=1, render=[1, 2], rays=1024, intr=S.K_RE10K, z=(1.0, 100.0), cap=False, code="distance")}[shape]
if len(sys.argv) > 3:
    SH["K"] = int(sys.argv[3])
n, V, H, W, C, K = SH["n"], SH["V"], SH["H"], SH["W"], SH["C"], SH["K"]
scene = S.synthetic_scene(n, V, H, W, C, seed=5, intrinsics=SH["intr"], smooth=True)
net = bts.BTSNet(S.field_conf(C, C, SH["NB"], H, W, z_near=SH["z"][0], z_far=SH["z"][1], code_mode=SH["code"])); S.init_mlp_(net.mlp_coarse, seed=7)
net.encoder = bts.FeatureMapEncoder((H, W), C, num_views=n)
with torch.no_grad():
    net.encoder.feats[0].data = scene["feat"].clone()
net = net.cuda().eval()
images, projs, poses = scene["images"].cuda(), scene["projs"].cuda(), scene["poses"].cuda()
with torch.no_grad():
    net.encode(images, projs, poses, ids_encoder=[0], ids_render=SH["render"])
    ft = net.native_field()
    sampler = bts.PatchRaySampler(ray_batch_size=SH["rays"], z_near=SH["z"][0], z_far=SH["z"][1], patch_size=8)
    nl = SH["loss"]
    rays, _ = sampler.sample(images[:, :nl] * .5 + .5, poses[:, :nl], projs[:, :nl])
    rays = rays.reshape(-1, 8).contiguous()
    z = native.sample_coarse(rays, torch.rand(rays.shape[0], K, device="cuda"), True)
    params = net.mlp_coarse.packed().detach()
    out = native.render_fwd(ft, params, rays, z, hard_alpha_cap=SH["cap"], want_rgb_samps=True, want_saved=True)
    g_rgb, g_depth = torch.randn_like(out["rgb"]), torch.randn_like(out["depth"]) * 0.1
    for name, rs in (("colours recomputed", None), ("colours from the forward", out["rgb_samps"])):
        ts = []
        for r in range(rounds + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            native.render_bwd(ft, params, rays, z, out["sigma_raw"], out["trans"], hard_alpha_cap=SH["cap"], g_rgb=g_rgb, g_depth=g_depth, rgb_samps=rs)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts = sorted(ts[1:])
        print(f"bts_render_bwd [{shape}], {name}: median {ts[len(ts) // 2]:.3f} ms  min {ts[0]:.3f} ms   ({rays.shape[0]} rays x {K})")
