"""Golden fixture for the training-mode density noise (nerf.py:279-280) FROM THE REAL REFERENCE.  Run in the build container only:

    python -B tests/golden/gen_golden_noise.py

The reference's NeRFRenderer in train() mode with noise_std > 0 draws `torch.randn_like(sigmas) * noise_std` from the CPU generator
inside composite() (the only draw in there).  Seeding the generator, calling composite, and drawing `torch.randn(B, K)` from the same
seed gives the noise tensor the reference used: noise.npz holds it with the reference's outputs and autograd gradients.  Pins
oracle.composite(sigma_noise=...) (tests/test_oracle_golden.py) and BtsRenderArgs.sigma_noise (tests/test_gpu_protocol.py).
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np
import torch

from oracle import bts_oracle as O
from oracle.ref_shim import load_reference
from gen_golden import ref_conf, load_mlp_into, mlp_arrays

torch.set_num_threads(4)


def main():
    ref = load_reference()
    cfg = O.FieldConfig()
    seed, n, v, H, W, C, Hd, K, B, noise_std = 31, 2, 3, 32, 96, 64, 64, 16, 160, 0.7
    ids_render = [1, 2]
    g = torch.Generator().manual_seed(seed)
    scene = O.synthetic_scene(n, v, H, W, C, seed=seed, intrinsics=O.K_KITTI360, baseline=0.6)
    mlp = O.init_mlp(C + 39, Hd, 0, gen=g)
    mlp.b_in = torch.randn(Hd, generator=g) * 0.1
    mlp.b_out = torch.tensor([-0.3])             # softplus(s) around 0.5: noise of 0.7 pushes a good share of the samples below 0 (relu)
    net = ref.make_net(ref_conf(cfg, 0, Hd), [scene["feat"]])
    load_mlp_into(net, mlp)
    renderer = ref.NeRFRenderer(n_coarse=K, lindisp=True, hard_alpha_cap=True, noise_std=noise_std)
    net.eval(), renderer.train()
    net.encode(scene["images"], scene["projs"], scene["poses"], ids_encoder=[0], ids_render=ids_render)
    sampler = ref.ImageRaySampler(cfg.d_min, cfg.d_max, H, W)
    all_rays, _ = sampler.sample(None, scene["poses"][:, :1], scene["projs"][:, :1])
    idx = torch.randperm(all_rays.shape[1], generator=g)[:B].sort().values
    rays = all_rays[:, idx].contiguous()
    torch.manual_seed(seed + 1)
    z_samp = renderer.sample_coarse(rays.reshape(-1, 8))
    params = [p for p in net.mlp_coarse.parameters()] + [net.encoder.feats[0]]
    torch.manual_seed(seed + 2)
    weights, rgb, depth, alphas, invalid, _, rgb_samps = renderer.composite(net, rays.reshape(-1, 8), z_samp, coarse=True, sb=n)
    torch.manual_seed(seed + 2)
    noise = torch.randn(n * B, K) * noise_std                                  # the draw composite() made
    g_rgb = torch.randn(rgb.shape, generator=g)
    g_depth = torch.randn(depth.shape, generator=g) * 0.1
    grads = torch.autograd.grad((rgb * g_rgb).sum() + (depth * g_depth).sum(), params)
    rename = {"lin_in.weight": "g_w_in", "lin_in.bias": "g_b_in", "lin_out.weight": "g_w_out", "lin_out.bias": "g_b_out"}
    names = [rename[k] for k, _ in net.mlp_coarse.named_parameters()] + ["g_feat"]
    arrays = dict(images=scene["images"], feat=scene["feat"], projs=scene["projs"], poses=scene["poses"], rays=rays, z_samp=z_samp,
                  sigma_noise=noise, out_weights=weights, out_rgb=rgb, out_depth=depth, out_alphas=alphas, out_invalid=invalid,
                  gin_rgb=g_rgb, gin_depth=g_depth, **dict(zip(names, grads)), **mlp_arrays(mlp))
    meta = dict(n=n, v=v, H=H, W=W, C=C, Hd=Hd, nb=0, K=K, ids_render=ids_render, hard_cap=True, d_min=cfg.d_min, d_max=cfg.d_max,
                inv_z=cfg.inv_z, code_mode=cfg.code_mode, learn_empty=cfg.learn_empty, empty_empty=cfg.empty_empty, num_freqs=cfg.num_freqs,
                freq_factor=cfg.freq_factor, norm_dir=True, noise_std=noise_std)
    np.savez_compressed(os.path.join(HERE, "noise.npz"), meta=np.array(repr(meta)),
                        **{k: (a.detach().numpy() if torch.is_tensor(a) else a) for k, a in arrays.items()})
    with torch.no_grad():   # how much of the case the noise actually decides
        _, _, sig = net(((rays.reshape(-1, 8)[:, None, :3] + z_samp.unsqueeze(2) * rays.reshape(-1, 8)[:, None, 3:6]).reshape(n, -1, 3)))
    cut = ((sig.reshape(-1, K) + noise) <= 0).float().mean().item()
    print(f"noise.npz: rays {tuple(rays.shape)} K {K}; samples the noise pushes to relu's zero side: {cut:.3f}; depth [{depth.min().item():.2f}, {depth.max().item():.2f}]")


if __name__ == "__main__":
    main()
