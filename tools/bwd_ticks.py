"""Where do the waves of the gate-bit backward (rows_kernel = pass A, scatter_kernel = pass B) spend their cycles?  Needs the diagnostic build
    python -m behindthescenes_amd.build --tag ticks -DBTS_TICKS
    BTS_ALLOW_LIB_OVERRIDE=1 BTS_RENDER_LIB=behindthescenes_amd/variants/libbts_ticks.so python tools/bwd_ticks.py [kitti360 | kitti_raw]
rows_kernel, cycles per ray iteration: 0 head (ray record, camera, per-sample loads issued, geometry, taps, tile broadcast, first gather blocks
out)   1 upstream weight gradient + compositing gradient (waits for the per-sample loads)   2 forward pipeline (gather, encoding, lin_in)
3 gate masks + dw_out.
scatter_kernel, cycles per step: 0 step inputs + geometry + taps   1 footprint (four wave minima), window move + flushes   2 slots, conflict
test, table   3 read-modify-write rounds; per wave: 12 set-up, 13 final flush."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import behindthescenes_amd as bts
from behindthescenes_amd import native, synthetic as S

shape = sys.argv[1] if len(sys.argv) > 1 else "kitti360"
SH = {"kitti360": dict(n=16, V=8, loss=4, render=[4, 5, 6, 7], rays=4096, intr=S.K_KITTI360),
      "kitti_raw": dict(n=8, V=4, loss=2, render=[2, 3], rays=2048, intr=S.K_KITTIRAW)}[shape]
n, V, H, W, C, K = SH["n"], SH["V"], 192, 640, 64, 64
scene = S.synthetic_scene(n, V, H, W, C, seed=5, intrinsics=SH["intr"], smooth=True)
net = bts.BTSNet(S.field_conf(C, C, 0, H, W, z_near=3.0, z_far=80.0)); S.init_mlp_(net.mlp_coarse, seed=7)
net.encoder = bts.FeatureMapEncoder((H, W), C, num_views=n)
with torch.no_grad():
    net.encoder.feats[0].data = scene["feat"].clone()
net = net.cuda().eval()
images, projs, poses = scene["images"].cuda(), scene["projs"].cuda(), scene["poses"].cuda()
dbg = torch.zeros(2 * 4096 * 16, dtype=torch.int64, device="cuda")
os.environ["BTS_DBG_PTR"] = str(dbg.data_ptr())
with torch.no_grad():
    net.encode(images, projs, poses, ids_encoder=[0], ids_render=SH["render"])
    ft = net.native_field()
    sampler = bts.PatchRaySampler(ray_batch_size=SH["rays"], z_near=3.0, z_far=80.0, patch_size=8)
    nl = SH["loss"]
    rays, _ = sampler.sample(images[:, :nl] * .5 + .5, poses[:, :nl], projs[:, :nl])
    rays = rays.reshape(-1, 8).contiguous()
    z = native.sample_coarse(rays, torch.rand(rays.shape[0], K, device="cuda"), True)
    params = net.mlp_coarse.packed().detach()
    out = native.render_fwd(ft, params, rays, z, hard_alpha_cap=True, want_rgb_samps=True, want_saved=True)
    g_rgb, g_depth = torch.randn_like(out["rgb"]), torch.randn_like(out["depth"]) * 0.1
    for r in range(3):
        dbg.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        native.render_bwd(ft, params, rays, z, out["sigma_raw"], out["trans"], hard_alpha_cap=True, g_rgb=g_rgb, g_depth=g_depth, rgb_samps=out["rgb_samps"])
        e1.record()
        torch.cuda.synchronize()
    print(f"{shape}: bts_render_bwd {e0.elapsed_time(e1):.3f} ms (instrumented)")
    for name, d, cols in (("rows_kernel", dbg[:4096 * 16], 4), ("scatter_kernel", dbg[4096 * 16:], 4)):
        d = d.view(-1, 16).double().cpu()
        d = d[d[:, 14] > 0]
        if not d.shape[0]:
            print(name, "no records (library built without -DBTS_TICKS?)")
            continue
        per = d[:, :cols].sum(0) / d[:, 14].sum()
        print(f"{name}: {d.shape[0]} waves recorded, {d[:, 14].mean():.1f} iterations each, wave lifetime {d[:, 15].mean():.0f} ticks (max {d[:, 15].max():.0f})")
        print("   cycles per iteration by section:", " ".join(f"{i}:{x:.0f}" for i, x in enumerate(per.tolist())), f" sum {per.sum():.0f}")
        if name == "scatter_kernel":
            print(f"   per wave: set-up {d[:, 12].mean():.0f}, final flush {d[:, 13].mean():.0f}")
