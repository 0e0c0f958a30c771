"""Per-section cycle counters of the render kernel's TRAINING instantiation (probe build), at a BASELINE training shape:
    python tools/section_probe_train.py [kitti360 | kitti_raw | re10k]
lean training outputs (per-ray reductions from the epilogue, saved activations, rgb_samps for the backward).  Sections as in
tools/section_probe.py: 0 geometry + colour issue, 1 gather + encoding + MFMA, 2 lin_out + softplus, 3 colour blend + compositing scan,
4 per-sample stores, 5 per-ray sums + stores."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BTS_RENDER_LIB", os.path.join(ROOT, "behindthescenes_amd", "variants", "libbts_probe.so"))
os.environ.setdefault("BTS_ALLOW_LIB_OVERRIDE", "1")
import torch
import behindthescenes_amd as bts
from behindthescenes_amd import native, synthetic as S

shape = sys.argv[1] if len(sys.argv) > 1 else "kitti360"
SH = {"kitti360": dict(n=16, V=8, H=192, W=640, C=64, NB=0, K=64, loss=4, render=[4, 5, 6, 7], rays=4096, intr=S.K_KITTI360, z=(3.0, 80.0), cap=True, code="z"),
      "kitti_raw": dict(n=8, V=4, H=192, W=640, C=64, NB=0, K=64, loss=2, render=[2, 3], rays=2048, intr=S.K_KITTIRAW, z=(3.0, 80.0), cap=True, code="z"),
      "re10k": dict(n=24, V=3, H=256, W=384, C=32, NB=1, K=48, loss=1, render=[1, 2], rays=1024, intr=S.K_RE10K, z=(1.0, 100.0), cap=False, code="distance")}[shape]
n, V, H, W, C, K = SH["n"], SH["V"], SH["H"], SH["W"], SH["C"], SH["K"]
scene = S.synthetic_scene(n, V, H, W, C, seed=5, intrinsics=SH["intr"], smooth=True)
net = bts.BTSNet(S.field_conf(C, C, SH["NB"], H, W, z_near=SH["z"][0], z_far=SH["z"][1], code_mode=SH["code"])); S.init_mlp_(net.mlp_coarse, seed=7)
net.encoder = bts.FeatureMapEncoder((H, W), C, num_views=n)
with torch.no_grad():
    net.encoder.feats[0].data = scene["feat"].clone()
net = net.cuda().eval()
images, projs, poses = scene["images"].cuda(), scene["projs"].cuda(), scene["poses"].cuda()
with torch.no_grad():
    net.encode(images, projs, poses, ids_encoder=[0], ids_render=SH["render"])
    ft = net.native_field()
    sampler = bts.PatchRaySampler(ray_batch_size=SH["rays"], z_near=SH["z"][0], z_far=SH["z"][1], patch_size=8)
    nl = SH["loss"]
    rays, _ = sampler.sample(images[:, :nl] * .5 + .5, poses[:, :nl], projs[:, :nl])
    rays = rays.reshape(-1, 8).contiguous()
    u = torch.rand(rays.shape[0], K, device="cuda")
    params = net.mlp_coarse.packed().detach()
    dbg = torch.zeros(512 * 4 * 8, dtype=torch.int64, device="cuda")
    os.environ["BTS_DBG_PTR"] = str(dbg.data_ptr())
    os.environ["BTS_ABLATE"] = "128"
    for _ in range(3):
        dbg.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        native.render_fwd(ft, params, rays, None, jitter=u, lindisp=True, want_z=True, hard_alpha_cap=SH["cap"], want_invalid=False, want_rgb_samps=True,
                          want_saved=True, want_invalid_sums=True)
        e1.record()
        torch.cuda.synchronize()
    d = dbg.view(-1, 8).double().cpu()
    d = d[d[:, 6] > 0]
    iters = rays.shape[0] / d.shape[0]
    per = d[:, :6].mean(0) / iters
    print(f"{shape}: {e0.elapsed_time(e1):.3f} ms (instrumented), waves {d.shape[0]}, iterations/wave {iters:.1f}")
    print("   ticks per iteration by section:", " ".join(f"{x:8.1f}" for x in per.tolist()), " sum", f"{per.sum():.1f}")
