"""torch.profiler view of the bench.py --workload train step: which framework ops (copies, fills, index, cat ...) surround the
HIP kernels, with their input shapes.   python tools/train_ops_profile.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import behindthescenes_amd as bts
from behindthescenes_amd import synthetic as S
from torch.profiler import profile, ProfilerActivity

H, W, C, HD, n, Vt, Kt = 192, 640, 64, 64, 16, 8, 64
dev = "cuda"
scene = S.synthetic_scene(n, Vt, H, W, C, seed=2000, intrinsics=S.K_KITTI360, baseline=0.6, smooth=True)
net = bts.BTSNet(S.field_conf(C, HD, 0, H, W)); S.init_mlp_(net.mlp_coarse, seed=7)
net.encoder = bts.FeatureMapEncoder((H, W), C, num_views=n)
S.set_feature_map(net, scene["feat"])
net = net.to(dev).train()
renderer = bts.NeRFRenderer.from_conf(dict(n_coarse=Kt, lindisp=True, hard_alpha_cap=True)).to(dev).train()
sampler = bts.PatchRaySampler(ray_batch_size=4096, z_near=3.0, z_far=80.0, patch_size=8)
images, projs, poses = scene["images"].to(dev), scene["projs"].to(dev), scene["poses"].to(dev)
wrapped = renderer.bind_parallel(net).train()
crit = bts.ReconstructionLoss({"criterion": "l1+ssim", "invalid_policy": "weight_guided", "lambda_edge_aware_smoothness": 0.001})

def step():
    net.zero_grad(set_to_none=True)
    images_ip = images * .5 + .5
    net.encode(images, projs, poses, ids_encoder=[0], ids_render=[4, 5, 6, 7], images_alt=images_ip)
    all_rays, all_rgb_gt = sampler.sample(images_ip[:, :4], poses[:, :4], projs[:, :4])
    rd = wrapped(all_rays, want_weights=True, want_alphas=True, want_rgb_samps=True)
    rd["fine"] = dict(rd["coarse"])
    rd["rgb_gt"], rd["rays"] = all_rgb_gt, all_rays
    rd = sampler.reconstruct(rd)
    crit(dict(coarse=[rd["coarse"]], fine=[rd["fine"]], rgb_gt=rd["rgb_gt"]))[0].backward()

for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=28, max_name_column_width=48, max_shapes_column_width=70))
