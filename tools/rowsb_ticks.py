"""Where does a wave of rowsb_kernel (pass A of the row backward) spend its cycles?  Needs the diagnostic build
    python -m behindthescenes_amd.build --tag ticks -DBTS_TICKS
    BTS_ALLOW_LIB_OVERRIDE=1 BTS_RENDER_LIB=behindthescenes_amd/variants/libbts_ticks.so python tools/rowsb_ticks.py [K]
Sections (cycles per ray-chunk iteration, averaged over the waves):
 0 top: loads issued, geometry, taps, table, first blocks out   1 compositing gradient   2 forward pipeline (gather, encoding, lin_in)
 3 block forward   4 dw_out   5 v, vn = mn.W1^T v   6 dW1 tiles   7 t2 = W0^T vn   8 dW0 tiles + v update   9 u0 row stores issued
 10 - 12 split section 0 further: ray scalars in | per-sample loads issued | projection, taps, tile broadcast | (0 itself: table + first blocks)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import behindthescenes_amd as bts
from behindthescenes_amd import native, synthetic as S

K = int(sys.argv[1]) if len(sys.argv) > 1 else 48
n, V, H, W, C = 24, 3, 256, 384, 32
scene = S.synthetic_scene(n, V, H, W, C, seed=5, intrinsics=S.K_RE10K, smooth=True)
net = bts.BTSNet(S.field_conf(C, C, 1, H, W, z_near=1.0, z_far=100.0, code_mode="distance")); S.init_mlp_(net.mlp_coarse, seed=7)
net.encoder = bts.FeatureMapEncoder((H, W), C, num_views=n)
with torch.no_grad():
    net.encoder.feats[0].data = scene["feat"].clone()
net = net.cuda().eval()
images, projs, poses = scene["images"].cuda(), scene["projs"].cuda(), scene["poses"].cuda()
dbg = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda")
os.environ["BTS_DBG_PTR"] = str(dbg.data_ptr())
with torch.no_grad():
    net.encode(images, projs, poses, ids_encoder=[0], ids_render=[1, 2])
    ft = net.native_field()
    sampler = bts.PatchRaySampler(ray_batch_size=1024, z_near=1.0, z_far=100.0, patch_size=8)
    rays, _ = sampler.sample(images[:, :1] * .5 + .5, poses[:, :1], projs[:, :1])
    rays = rays.reshape(-1, 8).contiguous()
    z = native.sample_coarse(rays, torch.rand(rays.shape[0], K, device="cuda"), True)
    params = net.mlp_coarse.packed().detach()
    out = native.render_fwd(ft, params, rays, z, hard_alpha_cap=False, want_rgb_samps=True, want_saved=True)
    g_rgb, g_depth = torch.randn_like(out["rgb"]), torch.randn_like(out["depth"]) * 0.1
    for r in range(3):
        dbg.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        native.render_bwd(ft, params, rays, z, out["sigma_raw"], out["trans"], hard_alpha_cap=False, g_rgb=g_rgb, g_depth=g_depth, rgb_samps=out["rgb_samps"])
        e1.record()
        torch.cuda.synchronize()
    d = dbg.view(-1, 16).double().cpu()
    d = d[d[:, 14] > 0]
    per = d[:, :13].sum(0) / d[:, 14].sum()
    life = d[:, 15].mean()
    print(f"bts_render_bwd {e0.elapsed_time(e1):.3f} ms; {d.shape[0]} waves, {d[:, 14].mean():.1f} iterations each, wave lifetime {life:.0f} ticks (max {d[:, 15].max():.0f})")
    print("ticks per iteration by section:", " ".join(f"{i}:{x:.0f}" for i, x in enumerate(per.tolist())), f" sum {per.sum():.0f}")
    print("share:                         ", " ".join(f"{i}:{100 * x / per.sum():.1f}%" for i, x in enumerate(per.tolist())))
