"""Golden fixtures for the two field modes no shipped config uses, FROM THE REAL REFERENCE (imported
unmodified through oracle/ref_shim.py).  Run in the build container only:

    python -B tests/golden/gen_golden_modes.py        ->  tests/golden/modes.npz

* ``combine``: three encoder views, ``combine_ids`` that merge two of them and leave the third a one-member group (the reference's
  feature merge then counts that view twice, models_bts.py:196-209: part of the contract), merged render views;
* ``mlpcolor``: ``sample_color: false`` -- a four-output MLP, sigma = relu(out[0]), colour = sigmoid(out[1:4]).
For each: BTSNet.forward on raw points and NeRFRenderer.composite (eval mode) incl. autograd gradients of a seeded scalar."""
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


def conf(sample_color, C, Hd):
    return dict(z_near=3.0, z_far=80.0, inv_z=True, learn_empty=True, empty_empty=False, code_mode="z", sample_color=sample_color,
                code=dict(num_freqs=6, freq_factor=1.5, include_input=True), encoder=dict(type="monodepth2"),
                mlp_coarse=dict(type="resnet", n_blocks=0, d_hidden=Hd), mlp_fine=dict(type="empty"))


def case(ref, name, out, *, sample_color, ids_encoder, ids_render, combine_ids, seed):
    g = torch.Generator().manual_seed(seed)
    n, v, H, W, C, Hd, K, B = 2, 6, 24, 40, 16, 32, 16, 96
    scene = O.synthetic_scene(n, v, H, W, C, seed=seed, intrinsics=O.K_KITTI360, yaw_deg=9.0, smooth=True)
    nv_enc = len(ids_encoder)
    feats = torch.randn(n * nv_enc, C, H, W, generator=g)
    net = ref.make_net(conf(sample_color, C, Hd), [feats])
    with torch.no_grad():
        for p in net.mlp_coarse.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.3 if p.dim() > 1 else 0.1))
        net.empty_feature.copy_(torch.randn(C, generator=g))
    renderer = ref.NeRFRenderer(n_coarse=K, lindisp=True, hard_alpha_cap=True)
    net.eval(), renderer.eval()
    net.encode(scene["images"], scene["projs"], scene["poses"], ids_encoder=ids_encoder, ids_render=ids_render, combine_ids=combine_ids)
    rays_all = O.image_rays(scene["poses"][:, :1], scene["projs"][:, :1], H, W, 3.0, 80.0)
    rays = rays_all[:, torch.randperm(rays_all.shape[1], generator=g)[:B].sort().values].contiguous()
    z = O.sample_coarse(rays.reshape(-1, 8), K, True, torch.rand(n * B, K, generator=g))
    params = list(net.mlp_coarse.parameters()) + [net.encoder.feats[0], net.empty_feature]
    w, rgb, depth, alphas, invalid, _, rgbs = renderer.composite(net, rays.reshape(-1, 8), z, coarse=True, sb=n)
    g_rgb, g_depth = torch.randn(rgb.shape, generator=g), torch.randn(depth.shape, generator=g) * 0.1
    grads = torch.autograd.grad((rgb * g_rgb).sum() + (depth * g_depth).sum(), params, allow_unused=True)
    pts = (rays.reshape(-1, 8)[:, None, :3] + z.unsqueeze(2) * rays.reshape(-1, 8)[:, None, 3:6]).reshape(n, -1, 3)[:, :200].contiguous()
    with torch.no_grad():
        q_rgb, q_inv, q_sig = net(pts)
    arr = dict(images=scene["images"], projs=scene["projs"], poses=scene["poses"], feats=feats, empty=net.empty_feature, rays=rays, z=z,
               w_in=net.mlp_coarse.lin_in.weight, b_in=net.mlp_coarse.lin_in.bias, w_out=net.mlp_coarse.lin_out.weight,
               b_out=net.mlp_coarse.lin_out.bias, weights=w, rgb=rgb, depth=depth, alphas=alphas, invalid=invalid, rgb_samps=rgbs,
               gin_rgb=g_rgb, gin_depth=g_depth, q_pts=pts, q_rgb=q_rgb, q_invalid=q_inv, q_sigma=q_sig)
    names = [k for k, _ in net.mlp_coarse.named_parameters()] + ["feats", "empty"]
    for k, gr in zip(names, grads):
        arr["g_" + k.replace(".", "_")] = torch.zeros(1) if gr is None else gr
    for k, a in arr.items():
        out[f"{name}_{k}"] = a.detach().numpy()
    out[f"{name}_meta"] = np.array(repr(dict(n=n, v=v, H=H, W=W, C=C, Hd=Hd, K=K, sample_color=sample_color, ids_encoder=ids_encoder,
                                             ids_render=ids_render, combine_ids=combine_ids)))
    print(name, "rgb", tuple(rgb.shape), "invalid frac", invalid.mean().item(), "depth", depth.min().item(), depth.max().item())


if __name__ == "__main__":
    ref = load_reference()
    out = {}
    case(ref, "combine", out, sample_color=True, ids_encoder=[0, 2, 4], ids_render=[1, 3, 5], combine_ids=[(0, 2), (1, 3)], seed=41)
    case(ref, "mlpcolor", out, sample_color=False, ids_encoder=[0], ids_render=[1, 2], combine_ids=None, seed=42)
    np.savez_compressed(os.path.join(HERE, "modes.npz"), **out)
    print("wrote", os.path.join(HERE, "modes.npz"), os.path.getsize(os.path.join(HERE, "modes.npz")) // 1024, "KB")
