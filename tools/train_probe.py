"""Times the renderer's share of a training step on BASELINE configs[2] shapes (exp_kitti_360.yaml: bs 16, 4096 patch rays per
sample, K = 64, 4 render views; no CNN): hand-over (project), forward with saved activations, backward (render_bwd + project_bwd).
    python tools/train_probe.py [n] [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import behindthescenes_amd as bts
from behindthescenes_amd import native
from behindthescenes_amd import synthetic as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
H, W, K, C, HD, V = 192, 640, 64, 64, 64, 8
Z_NEAR, Z_FAR = 3.0, 80.0
scene = S.synthetic_scene(n, V, H, W, C, seed=5, intrinsics=S.K_KITTI360, smooth=True)
net = bts.BTSNet(S.field_conf(C, HD, 0, H, W)); S.init_mlp_(net.mlp_coarse, seed=7)
net.encoder = bts.FeatureMapEncoder((H, W), C, num_views=n)
with torch.no_grad():
    net.encoder.feats[0].data = scene["feat"].clone()
net = net.cuda().train()
renderer = bts.NeRFRenderer.from_conf(dict(n_coarse=K, lindisp=True, hard_alpha_cap=True)).cuda().train()
crit = bts.ReconstructionLoss({"criterion": "l1+ssim", "invalid_policy": "weight_guided", "lambda_edge_aware_smoothness": 0.001})
sampler = bts.PatchRaySampler(ray_batch_size=4096, z_near=Z_NEAR, z_far=Z_FAR, patch_size=8)
images, projs, poses = scene["images"].cuda(), scene["projs"].cuda(), scene["poses"].cuda()
ids_loss, ids_render = [0, 1, 2, 3], [4, 5, 6, 7]
ev = lambda: torch.cuda.Event(enable_timing=True)
res = {}
for r in range(rounds + 1):
    net.zero_grad(set_to_none=True)
    t = [ev() for _ in range(6)]
    t[0].record()
    net.encode(images, projs, poses, ids_encoder=[0], ids_render=ids_render, images_alt=images * .5 + .5)
    all_rays, gt = sampler.sample(images[:, ids_loss] * .5 + .5, poses[:, ids_loss], projs[:, ids_loss])
    t[1].record()
    ft = net.native_field()                       # project_features (hand-over)
    t[2].record()
    out = renderer.composite(net, all_rays.reshape(-1, 8), renderer.sample_coarse(all_rays.reshape(-1, 8)), sb=n)
    w, rgb, depth, a, inv, _, rs = out
    t[3].record()
    pch = lambda x, *tail: x.reshape(n, 64, 8, 8, *tail)      # PatchRaySampler.reconstruct layout
    level = dict(rgb=pch(rgb, 4, 3), depth=pch(depth), weights=pch(w, K), invalid=pch(inv, K, 4), alphas=pch(a, K))
    loss, _ = crit(dict(coarse=[level], fine=[dict(level)], rgb_gt=pch(gt, 3)))
    t[4].record()
    loss.backward()
    t[5].record()
    torch.cuda.synchronize()
    if r > 0:
        for name, i in (("encode+sample", 0), ("project", 1), ("render fwd (saved, all outputs)", 2), ("photometric loss (fused, incl. its gradient)", 3), ("backward (render_bwd, project_bwd)", 4)):
            res.setdefault(name, []).append(t[i].elapsed_time(t[i + 1]))
B = n * 4096
print(f"n={n}: {B} rays x {K} samples, nv=4")
tot = 0
for k, v in res.items():
    v = sorted(v); m = v[len(v) // 2]; tot += m
    print(f"  {k:45s} {m:8.3f} ms")
print(f"  renderer share of a step: {tot:.3f} ms -> {B / tot / 1e3:.2f} M rays/s")

# the two projection-backward kernels on their own (dG = the gradient the render backward just produced, in spirit: random here)
spec = ft.spec if hasattr(ft, "spec") else None
if spec is not None:
    feat = net.encoder.feats[0].data.contiguous()
    dG = torch.randn(n, H, W, HD, device="cuda")
    mp = net.mlp_coarse.packed()
    if mp is not None:
        for name, kw in (("project_bwd feat only", dict(need_feat=True, need_mlp=False)), ("project_bwd weight only", dict(need_feat=False, need_mlp=True))):
            ts = []
            for r in range(rounds + 1):
                a, b = ev(), ev(); a.record(); native.project_features_bwd(spec, feat, dG, mp.detach(), **kw); b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            print(f"  {name:45s} {sorted(ts[1:])[len(ts[1:]) // 2]:8.3f} ms")
