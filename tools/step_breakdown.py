"""Host-side breakdown of the bench.py step (eager): wall time per stage with a device sync after each, and the un-synced total."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import behindthescenes_amd as bts
from behindthescenes_amd import synthetic as S
H, W, K, C, HD, V = 192, 640, 64, 64, 64, 2
Z_NEAR, Z_FAR = 3.0, 80.0
scene = S.synthetic_scene(1, V, H, W, C, seed=1000, intrinsics=S.K_KITTIRAW)
net = bts.BTSNet(S.field_conf(C, HD, 0, H, W)); S.init_mlp_(net.mlp_coarse, seed=7)
with torch.no_grad():
    net.encoder.feats[0].data = scene["feat"].clone()
net = net.cuda().eval()
wrapped = bts.NeRFRenderer.from_conf(dict(n_coarse=K, lindisp=True, hard_alpha_cap=True)).bind_parallel(net).eval().cuda()
sampler = bts.ImageRaySampler(Z_NEAR, Z_FAR)
images, projs, poses = scene["images"].cuda(), scene["projs"].cuda(), scene["poses"].cuda()
stages = {}
def tick(name, t0):
    torch.cuda.synchronize(); stages.setdefault(name, []).append(time.perf_counter() - t0); return time.perf_counter()
with torch.no_grad():
    for it in range(12):
        t = time.perf_counter()
        net.encode(images, projs, poses, ids_encoder=[0], ids_render=[0]); t = tick("encode", t)
        all_rays, all_rgb_gt = sampler.sample(images * .5 + .5, poses, projs); t = tick("sample", t)
        rd = wrapped(all_rays, want_weights=True, want_alphas=True); t = tick("render (rand + sample_coarse + project + kernel)", t)
        rd["fine"] = dict(rd["coarse"]); rd["rgb_gt"] = all_rgb_gt
        rd = sampler.reconstruct(rd); t = tick("reconstruct", t)
        dz = bts.distance_to_z(rd["coarse"]["depth"], projs); t = tick("distance_to_z", t)
for k, v in stages.items():
    v = sorted(v[2:]); print(f"{k:55s} {1e3 * v[len(v)//2]:.3f} ms")
