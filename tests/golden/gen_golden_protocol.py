"""Golden fixtures for the PROTOCOL pieces around the hot loop, produced by the REAL reference (imported unmodified through
oracle/ref_shim.py; build container only):   python -B tests/golden/gen_golden_protocol.py   -> protocol.npz

  * ImageRaySampler.sample / .reconstruct          (models/bts/model/ray_sampler.py:224-321)
  * RandomRaySampler.sample / .reconstruct         (ray_sampler.py:15-106), seeded CPU draws
  * NeRFRenderer.sample_coarse_from_dist / sample_fine / sample_fine_depth   (models/common/render/nerf.py:125-208), seeded CPU draws
  * NeRFRenderer.sched_step                        (nerf.py:403-423)
  * composite with white_bkgd=True, outputs and gradients   (nerf.py:301-304)
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
from tests.golden.gen_golden import load_mlp_into, mlp_arrays, ref_conf

torch.set_num_threads(4)


def fake_render_dict(g, n, n_pts, K, nv, with_extras=True):
    part = dict(rgb=torch.randn(n, n_pts, nv * 3, generator=g), weights=torch.rand(n, n_pts, K, generator=g),
                depth=torch.rand(n, n_pts, generator=g), invalid=(torch.rand(n, n_pts, K, nv, generator=g) > 0.5).float())
    if with_extras:
        part.update(alphas=torch.rand(n, n_pts, K, generator=g), z_samps=torch.rand(n, n_pts, K, generator=g),
                    rgb_samps=torch.rand(n, n_pts, K, nv * 3, generator=g))
    return part


def main():
    ref = load_reference()
    g = torch.Generator().manual_seed(2024)
    out = {}
    # ---------------- ImageRaySampler
    n, v, H, W, K, nv = 2, 2, 6, 10, 5, 2
    scene = O.synthetic_scene(n, v, H, W, 4, seed=9, intrinsics=O.K_KITTIRAW, yaw_deg=4.0)
    images = scene["images"]
    s = ref.ImageRaySampler(3.0, 80.0)                       # height / width taken from the images at first use
    rays, gt = s.sample(images, scene["poses"], scene["projs"])
    part = fake_render_dict(g, n, v * H * W, K, nv)
    rd = s.reconstruct(dict(coarse=dict(part), fine=dict(part), rgb_gt=gt))
    out.update(img_images=images, img_poses=scene["poses"], img_projs=scene["projs"], img_rays=rays, img_gt=gt,
               **{f"img_in_{k}": t for k, t in part.items()}, **{f"img_out_{k}": t for k, t in rd["coarse"].items()}, img_out_rgb_gt=rd["rgb_gt"])
    s2 = ref.ImageRaySampler(3.0, 80.0, H, W, norm_dir=False)
    rays2, gt2 = s2.sample(None, scene["poses"], scene["projs"])
    assert gt2 is None
    out["img_rays_unnorm"] = rays2
    # ---------------- RandomRaySampler (CPU generator)
    rs = ref.RandomRaySampler(ray_batch_size=37, z_near=3.0, z_far=80.0)
    torch.manual_seed(123)
    r_rays, r_gt = rs.sample(images, scene["poses"], scene["projs"])
    part_r = fake_render_dict(g, n, 37, K, nv)
    rd_r = rs.reconstruct(dict(coarse=dict(part_r), fine=dict(part_r), rgb_gt=r_gt))
    out.update(rnd_rays=r_rays, rnd_gt=r_gt, **{f"rnd_in_{k}": t for k, t in part_r.items()},
               **{f"rnd_out_{k}": t for k, t in rd_r["coarse"].items()}, rnd_out_rgb_gt=rd_r["rgb_gt"])
    # ---------------- importance sampling helpers (both lindisp settings), seeded CPU draws
    B, Kc = 19, 8
    srays = torch.cat((torch.randn(B, 6, generator=g), torch.full((B, 1), 3.0), torch.full((B, 1), 80.0)), dim=-1)
    wts = torch.rand(B, Kc, generator=g)
    depth = torch.rand(B, generator=g) * 70 + 5
    for lindisp in (True, False):
        r = ref.NeRFRenderer(n_coarse=Kc, n_fine=6, n_fine_depth=2, lindisp=lindisp, depth_std=0.5)
        torch.manual_seed(5)
        zc = r.sample_coarse(srays)
        torch.manual_seed(6)
        z_dist = r.sample_coarse_from_dist(srays, wts, zc)
        torch.manual_seed(7)
        z_fine = r.sample_fine(srays, wts)
        torch.manual_seed(8)
        z_fd = r.sample_fine_depth(srays, depth)
        tag = "lin" if lindisp else "dep"
        out.update({f"smp_{tag}_zc": zc, f"smp_{tag}_dist": z_dist, f"smp_{tag}_fine": z_fine, f"smp_{tag}_fdepth": z_fd})
    out.update(smp_rays=srays, smp_weights=wts, smp_depth=depth)
    # ---------------- sched_step
    r = ref.NeRFRenderer(n_coarse=4, n_fine=0, sched=[[3, 7], [8, 16], [0, 4]])
    trace = []
    for step in range(10):
        r.sched_step(1 if step % 3 else 2)
        trace.append([int(r.iter_idx), int(r.last_sched), r.n_coarse, r.n_fine, int(r.using_fine)])
    out["sched_trace"] = np.array(trace)
    # ---------------- white background: composite outputs + gradients
    cfg = O.FieldConfig()
    n, v, H, W, C, Hd, K = 2, 3, 16, 48, 64, 64, 16
    scene = O.synthetic_scene(n, v, H, W, C, seed=77, intrinsics=O.K_KITTI360)
    mlp = O.init_mlp(C + 39, Hd, 0, gen=g)
    net = ref.make_net(ref_conf(cfg, 0, Hd), [scene["feat"]])
    load_mlp_into(net, mlp)
    renderer = ref.NeRFRenderer(n_coarse=K, lindisp=True, hard_alpha_cap=False, white_bkgd=True)
    net.eval(), renderer.eval()
    net.encode(scene["images"], scene["projs"], scene["poses"], ids_encoder=[0], ids_render=[1, 2])
    all_rays, _ = ref.ImageRaySampler(3.0, 80.0, H, W).sample(None, scene["poses"], scene["projs"])
    idx = torch.randperm(all_rays.shape[1], generator=g)[:200].sort().values
    wrays = all_rays[:, idx].contiguous()
    torch.manual_seed(31)
    wz = renderer.sample_coarse(wrays.reshape(-1, 8))
    params = [p for p in net.mlp_coarse.parameters()] + [net.encoder.feats[0]]
    weights, rgb, depth_o, alphas, invalid, _, rgb_samps = renderer.composite(net, wrays.reshape(-1, 8), wz, coarse=True, sb=n)
    g_rgb = torch.randn(rgb.shape, generator=g)
    grads = torch.autograd.grad((rgb * g_rgb).sum(), params)
    names = {"lin_in.weight": "g_w_in", "lin_in.bias": "g_b_in", "lin_out.weight": "g_w_out", "lin_out.bias": "g_b_out"}
    gnames = [names[k] for k, _ in net.mlp_coarse.named_parameters()] + ["g_feat"]
    out.update(wb_images=scene["images"], wb_feat=scene["feat"], wb_projs=scene["projs"], wb_poses=scene["poses"], wb_rays=wrays, wb_z=wz,
               wb_rgb=rgb, wb_depth=depth_o, wb_weights=weights, wb_gin_rgb=g_rgb, **{f"wb_{k}": t for k, t in mlp_arrays(mlp).items()},
               **{f"wb_{k}": t for k, t in zip(gnames, grads)})
    np.savez(os.path.join(HERE, "protocol.npz"), **{k: (t.detach().numpy() if torch.is_tensor(t) else t) for k, t in out.items()})
    print("protocol.npz:", len(out), "arrays;", "white-bkgd rgb range", float(rgb.min()), float(rgb.max()), "sched trace", trace[-1])


if __name__ == "__main__":
    main()
