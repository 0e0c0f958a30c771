"""Generates the golden fixtures in this directory FROM THE REAL REFERENCE (imported unmodified from
/root/reference through oracle/ref_shim.py).  Run in the build container only:

    python -B tests/golden/gen_golden.py

The reference has no tests of its own, so these input/output pairs -- produced by the reference's
own code on seeded synthetic inputs -- are what pins the oracle (tests/test_oracle_golden.py) and, on the GPU box
where the reference tree does not exist, the HIP path (tests/test_gpu_parity.py).
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from oracle import bts_oracle as O
from oracle.ref_shim import load_reference

torch.set_num_threads(4)


def ref_conf(cfg: O.FieldConfig, n_blocks, d_hidden):
    return dict(z_near=cfg.d_min, z_far=cfg.d_max, inv_z=cfg.inv_z, learn_empty=cfg.learn_empty,
                empty_empty=cfg.empty_empty, code_mode=cfg.code_mode,
                code=dict(num_freqs=cfg.num_freqs, freq_factor=cfg.freq_factor, include_input=cfg.include_input),
                encoder=dict(type="monodepth2"),
                mlp_coarse=dict(type="resnet", n_blocks=n_blocks, d_hidden=d_hidden), mlp_fine=dict(type="empty"))


def load_mlp_into(net, mlp: O.MlpParams):
    with torch.no_grad():
        net.mlp_coarse.lin_in.weight.copy_(mlp.w_in), net.mlp_coarse.lin_in.bias.copy_(mlp.b_in)
        for blk, (w0, b0, w1, b1) in zip(net.mlp_coarse.blocks, mlp.blocks):
            blk.fc_0.weight.copy_(w0), blk.fc_0.bias.copy_(b0), blk.fc_1.weight.copy_(w1), blk.fc_1.bias.copy_(b1)
        net.mlp_coarse.lin_out.weight.copy_(mlp.w_out), net.mlp_coarse.lin_out.bias.copy_(mlp.b_out)


def mlp_arrays(mlp: O.MlpParams):
    d = dict(w_in=mlp.w_in, b_in=mlp.b_in, w_out=mlp.w_out, b_out=mlp.b_out)
    for i, (w0, b0, w1, b1) in enumerate(mlp.blocks):
        d.update({f"blk{i}_w0": w0, f"blk{i}_b0": b0, f"blk{i}_w1": w1, f"blk{i}_b1": b1})
    return d


def run_case(ref, name, *, n, v, H, W, C, Hd, nb, K, ids_render, cfg, hard_cap, intr, rays_per_sample, seed,
             norm_dir=True, with_grad=False, bias_std=0.0, yaw=0.0):
    g = torch.Generator().manual_seed(seed)
    scene = O.synthetic_scene(n, v, H, W, C, seed=seed, intrinsics=intr, yaw_deg=yaw)
    d_in = C + 3 + 6 * cfg.num_freqs
    mlp = O.init_mlp(d_in, Hd, nb, gen=g)
    if bias_std > 0:  # non-zero biases so that a bias bug cannot hide
        mlp.b_in = torch.randn(Hd, generator=g) * bias_std
        mlp.b_out = torch.randn(1, generator=g) * bias_std
        mlp.blocks = [(w0, torch.randn(Hd, generator=g) * bias_std, w1, torch.randn(Hd, generator=g) * bias_std)
                      for (w0, _, w1, _) in mlp.blocks]
    # ---- the real reference objects
    net = ref.make_net(ref_conf(cfg, nb, Hd), [scene["feat"]])
    load_mlp_into(net, mlp)
    empty = None
    if cfg.learn_empty:
        empty = torch.randn(C, generator=g)
        with torch.no_grad():
            net.empty_feature.copy_(empty)
    renderer = ref.NeRFRenderer(n_coarse=K, lindisp=True, hard_alpha_cap=hard_cap)
    net.eval(), renderer.eval()
    net.encode(scene["images"], scene["projs"], scene["poses"], ids_encoder=[0], ids_render=ids_render)
    sampler = ref.ImageRaySampler(cfg.d_min, cfg.d_max, H, W, norm_dir=norm_dir)
    all_rays, _ = sampler.sample(None, scene["poses"], scene["projs"])        # (n, v*H*W, 8)
    # seeded subset of the rays of every sample (keeps fixtures small), always including image-border pixels
    idx = torch.randperm(all_rays.shape[1], generator=g)[:rays_per_sample].sort().values
    rays = all_rays[:, idx].contiguous()                                      # (n, B', 8)
    torch.manual_seed(seed + 1)
    z_samp = renderer.sample_coarse(rays.reshape(-1, 8))                      # reference jitter, CPU RNG
    torch.manual_seed(seed + 1)
    u = torch.rand(rays.shape[0] * rays.shape[1], K)                          # the same stream rand_like consumed

    params = [p for p in net.mlp_coarse.parameters()] + [net.encoder.feats[0]]
    with torch.set_grad_enabled(with_grad):
        out = renderer.composite(net, rays.reshape(-1, 8), z_samp, coarse=True, sb=n)
    weights, rgb, depth, alphas, invalid, _, rgb_samps = out
    arrays = dict(images=scene["images"], feat=scene["feat"], projs=scene["projs"], poses=scene["poses"],
                  rays=rays, z_samp=z_samp, u=u, ray_idx=idx,
                  out_weights=weights, out_rgb=rgb, out_depth=depth, out_alphas=alphas, out_invalid=invalid,
                  out_rgb_samps=rgb_samps, **mlp_arrays(mlp))
    if empty is not None:
        arrays["empty_feature"] = empty
    # field query fixture (BTSNet.forward on raw points, incl. only_density): first 300 points of the ray samples
    pts = (rays.reshape(-1, 8)[:, None, :3] + z_samp.unsqueeze(2) * rays.reshape(-1, 8)[:, None, 3:6]).reshape(n, -1, 3)[:, :300].contiguous()
    with torch.no_grad():
        q_rgb, q_inv, q_sig = net(pts)
        arrays.update(q_pts=pts, q_rgb=q_rgb, q_invalid=q_inv, q_sigma=q_sig)
        if not cfg.empty_empty:  # the reference itself raises IndexError for empty_empty + only_density (models_bts.py:324)
            _, q_inv_d, q_sig_d = net(pts, only_density=True)
            arrays.update(q_invalid_density=q_inv_d, q_sigma_density=q_sig_d)
    if with_grad:
        g_rgb = torch.randn(rgb.shape, generator=g)
        g_depth = torch.randn(depth.shape, generator=g) * 0.1
        loss = (rgb * g_rgb).sum() + (depth * g_depth).sum()
        grads = torch.autograd.grad(loss, params)
        # reference registration order is lin_in, lin_out, blocks (resnetfc.py:87-109): name by named_parameters
        rename = {"lin_in.weight": "g_w_in", "lin_in.bias": "g_b_in", "lin_out.weight": "g_w_out", "lin_out.bias": "g_b_out"}
        for i in range(nb):
            rename.update({f"blocks.{i}.fc_0.weight": f"g_blk{i}_w0", f"blocks.{i}.fc_0.bias": f"g_blk{i}_b0",
                           f"blocks.{i}.fc_1.weight": f"g_blk{i}_w1", f"blocks.{i}.fc_1.bias": f"g_blk{i}_b1"})
        names = [rename[k] for k, _ in net.mlp_coarse.named_parameters()] + ["g_feat"]
        arrays.update(dict(zip(names, grads)))
        arrays.update(gin_rgb=g_rgb, gin_depth=g_depth)
    meta = dict(n=n, v=v, H=H, W=W, C=C, Hd=Hd, nb=nb, K=K, ids_render=list(ids_render), hard_cap=hard_cap,
                d_min=cfg.d_min, d_max=cfg.d_max, inv_z=cfg.inv_z, code_mode=cfg.code_mode, learn_empty=cfg.learn_empty,
                empty_empty=cfg.empty_empty, num_freqs=cfg.num_freqs, freq_factor=cfg.freq_factor, norm_dir=norm_dir)
    np.savez(os.path.join(HERE, f"{name}.npz"), meta=np.array(repr(meta)),
             **{k: (a.detach().numpy() if torch.is_tensor(a) else a) for k, a in arrays.items()})
    print(f"{name}: rays {tuple(rays.shape)} depth[{depth.min().item():.3f},{depth.max().item():.3f}] "
          f"invalid-frac {invalid.mean().item():.3f}")


def misc_fixtures(ref):
    g = torch.Generator().manual_seed(7)
    # gen_rays (util.py:244-273) with and without normalised directions
    scene = O.synthetic_scene(1, 3, 12, 20, 4, seed=3, intrinsics=O.K_KITTIRAW, yaw_deg=7.0)
    poses, projs = scene["poses"][0], scene["projs"][0]
    focal, center = projs[:, [0, 1], [0, 1]], projs[:, [0, 1], [2, 2]]
    rays_n = ref.gen_rays(poses, 20, 12, 3.0, 80.0, focal=focal, c=center, norm_dir=True)
    rays_u = ref.gen_rays(poses, 20, 12, 3.0, 80.0, focal=focal, c=center, norm_dir=False)
    # distance_to_z (projection_operations.py:4-16)
    depths = torch.rand(2, 3, 12, 20, generator=g) * 70 + 3
    projs2 = torch.stack([projs, projs * torch.tensor([[1.1, 1, 1], [1, 0.9, 1], [1, 1, 1]])])
    dz = ref.distance_to_z(depths, projs2)
    # PatchRaySampler (ray_sampler.py:125-162): seeded patch coordinates + rays + gt colours
    images = torch.rand(2, 3, 3, 12, 20, generator=g) * 2 - 1
    ps = ref.PatchRaySampler(ray_batch_size=48, z_near=3.0, z_far=80.0, patch_size=4)
    torch.manual_seed(11)
    p_rays, p_rgb = ps.sample(images, scene["poses"].expand(2, -1, -1, -1), scene["projs"].expand(2, -1, -1, -1))
    np.savez(os.path.join(HERE, "misc.npz"), poses=poses.numpy(), projs=projs.numpy(), rays_norm=rays_n.numpy(),
             rays_unnorm=rays_u.numpy(), depths=depths.numpy(), projs2=projs2.numpy(), dist_to_z=dz.numpy(),
             patch_images=images.numpy(), patch_rays=p_rays.numpy(), patch_rgb=p_rgb.numpy())
    print("misc: gen_rays", tuple(rays_n.shape), "distance_to_z", tuple(dz.shape), "patch rays", tuple(p_rays.shape))


def main():
    ref = load_reference()
    kitti = O.FieldConfig(d_min=3.0, d_max=80.0, inv_z=True, code_mode="z")
    # KITTI-360-like training shape (exp_kitti_360.yaml): C=64, Hd=64, 0 blocks, hard cap, 2 render views, with grads
    run_case(ref, "kitti_train", n=2, v=3, H=16, W=48, C=64, Hd=64, nb=0, K=16, ids_render=[1, 2], cfg=kitti,
             hard_cap=True, intr=O.K_KITTI360, rays_per_sample=320, seed=100, with_grad=True, bias_std=0.1)
    # eval_depth.yaml shape: nv=1 rendered from the encoder view itself, learn_empty default True, both stereo views' rays
    kitti_e = O.FieldConfig(d_min=3.0, d_max=80.0, inv_z=True, code_mode="z", learn_empty=True)
    run_case(ref, "kitti_eval", n=1, v=2, H=16, W=48, C=64, Hd=64, nb=0, K=24, ids_render=[0], cfg=kitti_e,
             hard_cap=True, intr=O.K_KITTIRAW, rays_per_sample=512, seed=200)
    # gen_img_custom.py shape: single view, un-normalised directions, learn_empty false
    run_case(ref, "kitti_single", n=1, v=1, H=16, W=48, C=64, Hd=64, nb=0, K=32, ids_render=[0], cfg=kitti,
             hard_cap=True, intr=O.K_KITTI360, rays_per_sample=256, seed=300, norm_dir=False)
    # RE10K (exp_re10k.yaml): C=32, Hd=32, 1 block, distance code, no hard cap, z in [1,100], with grads
    re10k = O.FieldConfig(d_min=1.0, d_max=100.0, inv_z=True, code_mode="distance")
    run_case(ref, "re10k_train", n=2, v=3, H=16, W=24, C=32, Hd=32, nb=1, K=12, ids_render=[1, 2], cfg=re10k,
             hard_cap=False, intr=O.K_RE10K, rays_per_sample=256, seed=400, with_grad=True, bias_std=0.1, yaw=5.0)
    # off-config: inv_z false + empty_empty, 4 render views
    odd = O.FieldConfig(d_min=2.0, d_max=50.0, inv_z=False, code_mode="z", empty_empty=True)
    run_case(ref, "odd_cfg", n=1, v=5, H=16, W=24, C=32, Hd=32, nb=1, K=8, ids_render=[1, 2, 3, 4], cfg=odd,
             hard_cap=True, intr=O.K_RE10K, rays_per_sample=192, seed=500, yaw=20.0)
    misc_fixtures(ref)


if __name__ == "__main__":
    main()
