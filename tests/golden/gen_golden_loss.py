"""Generates tests/golden/train_step.npz FROM THE REAL REFERENCE: one training-step data flow after the CNN
(models/bts/trainer.py:208-259 + models/bts/model/loss.py): PatchRaySampler.sample -> NeRFRenderer.composite (seeded jitter) ->
_format_outputs -> fine = dict(coarse) -> PatchRaySampler.reconstruct -> ReconstructionLoss -> loss.backward().
Stores the inputs, the render dict the loss consumed, the loss value and its parts, and the gradients of the MLP / feature map.

    python -B tests/golden/gen_golden_loss.py      (build container only)"""
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
from tests.golden.gen_golden import load_mlp_into, mlp_arrays, ref_conf

torch.set_num_threads(4)


def main():
    ref = load_reference()
    cfg = O.FieldConfig(d_min=3.0, d_max=80.0, inv_z=True, code_mode="z")
    n, v, H, W, C, Hd, K, seed = 2, 4, 24, 64, 64, 64, 16, 700
    ids_loss, ids_render = [0, 1], [2, 3]          # exp_kitti_360.yaml "default" mode: frames split in two halves, encoder = frame 0
    g = torch.Generator().manual_seed(seed)
    scene = O.synthetic_scene(n, v, H, W, C, seed=seed, intrinsics=O.K_KITTI360, smooth=True)
    mlp = O.init_mlp(C + 39, Hd, 0, gen=g)
    mlp.b_in = torch.randn(Hd, generator=g) * 0.1
    net = ref.make_net(ref_conf(cfg, 0, Hd), [scene["feat"]])
    load_mlp_into(net, mlp)
    renderer = ref.NeRFRenderer(n_coarse=K, lindisp=True, hard_alpha_cap=True)
    net.train(), renderer.train()
    images_ip = scene["images"] * .5 + .5           # RGBProcessor (image_processor.py:22-29)
    net.encode(scene["images"], scene["projs"], scene["poses"], ids_encoder=[0], ids_render=ids_render, images_alt=images_ip)
    sampler = ref.PatchRaySampler(ray_batch_size=8 * 64, z_near=cfg.d_min, z_far=cfg.d_max, patch_size=8)
    torch.manual_seed(seed + 1)
    all_rays, all_rgb_gt = sampler.sample(images_ip[:, ids_loss], scene["poses"][:, ids_loss], scene["projs"][:, ids_loss])
    torch.manual_seed(seed + 2)
    z_samp = renderer.sample_coarse(all_rays.reshape(-1, 8))
    torch.manual_seed(seed + 2)
    u = torch.rand(all_rays.shape[0] * all_rays.shape[1], K)
    comp = renderer.composite(net, all_rays.reshape(-1, 8), z_samp, coarse=True, sb=n)
    out = renderer._format_outputs(comp, n, want_weights=True, want_alphas=True, want_z_samps=False, want_rgb_samps=True)
    render_dict = dict(coarse=dict(out.toDict() if hasattr(out, "toDict") else out))
    render_dict["fine"] = dict(render_dict["coarse"])
    render_dict["rgb_gt"] = all_rgb_gt
    render_dict = sampler.reconstruct(render_dict)
    data = dict(coarse=[render_dict["coarse"]], fine=[render_dict["fine"]], rgb_gt=render_dict["rgb_gt"])
    crit = ref.ReconstructionLoss(dict(criterion="l1+ssim", invalid_policy="weight_guided", lambda_edge_aware_smoothness=0.001))
    loss, parts = crit(data)
    params = [p for p in net.mlp_coarse.parameters()] + [net.encoder.feats[0]]
    grads = torch.autograd.grad(loss, params)
    names = [{"lin_in.weight": "g_w_in", "lin_in.bias": "g_b_in", "lin_out.weight": "g_w_out", "lin_out.bias": "g_b_out"}[k]
             for k, _ in net.mlp_coarse.named_parameters()] + ["g_feat"]
    c = data["coarse"][0]
    arrays = dict(images=scene["images"], feat=scene["feat"], projs=scene["projs"], poses=scene["poses"], rays=all_rays, rgb_gt=all_rgb_gt,
                  z_samp=z_samp, u=u, out_rgb=c["rgb"], out_depth=c["depth"], out_weights=c["weights"], out_alphas=c["alphas"],
                  out_invalid=c["invalid"], loss=loss.detach().reshape(1), loss_eas=torch.tensor([parts["loss_eas"]]),
                  loss_rgb_coarse=torch.tensor([parts["loss_rgb_coarse"]]), loss_invalid_ratio=torch.tensor([parts["loss_invalid_ratio"]]),
                  **mlp_arrays(mlp), **dict(zip(names, grads)))
    meta = dict(n=n, v=v, H=H, W=W, C=C, Hd=Hd, nb=0, K=K, ids_loss=ids_loss, ids_render=ids_render, hard_cap=True, patch=8, patches=8,
                d_min=cfg.d_min, d_max=cfg.d_max, seed=seed)
    np.savez(os.path.join(HERE, "train_step.npz"), meta=np.array(repr(meta)),
             **{k: a.detach().numpy() for k, a in arrays.items()})
    print("train_step: loss", loss.item(), {k: (float(x) if not torch.is_tensor(x) else x.item()) for k, x in parts.items()})


if __name__ == "__main__":
    main()
