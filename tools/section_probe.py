"""Per-section cycle counters of the pipelined render kernel (probe build): where does a wave's time go?
    python tools/section_probe.py
Sections (s_memtime deltas accumulated per wave, averaged over waves, per ray iteration):
 0 geometry (ray/camera loads, projection, taps, colour issue, tile broadcast)   1 gather + encoding + MFMA
 2 lin_out + softplus   3 colour blend + compositing scan   4 per-sample stores   5 per-ray sums + stores"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BTS_RENDER_LIB", os.path.join(ROOT, "behindthescenes_amd", "variants", "libbts_probe.so"))
os.environ.setdefault("BTS_ALLOW_LIB_OVERRIDE", "1")
import torch
import behindthescenes_amd as bts
from behindthescenes_amd import native
from behindthescenes_amd import synthetic as S

H, W, K, V = 192, 640, 64, 2
scene = S.synthetic_scene(1, V, H, W, 64, seed=1, intrinsics=S.K_KITTIRAW)
net = S.build_net(scene, 64, 0, [0])
ft = net.native_field()
params = net.mlp_coarse.packed().detach()
rays = bts.ImageRaySampler(3.0, 80.0, H, W).sample(None, scene["poses"].cuda(), scene["projs"].cuda())[0].reshape(-1, 8).contiguous()
z = native.sample_coarse(rays, torch.rand(rays.shape[0], K, device="cuda"), True)
dbg = torch.zeros(512 * 4 * 8, dtype=torch.int64, device="cuda")
os.environ["BTS_DBG_PTR"] = str(dbg.data_ptr())
half = rays.shape[0] // 2
sets = {"both": (rays, z), "view0": (rays[:half].contiguous(), z[:half].contiguous()), "view1": (rays[half:].contiguous(), z[half:].contiguous())}
for name, (r_, z_) in sets.items():
    for extra in ((0, 2, 4, 1, 2 | 4 | 1) if name != "both" else (0,)):
        os.environ["BTS_ABLATE"] = str(128 | extra)
        for _ in range(2):
            dbg.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            native.render_fwd(ft, params, r_, z_, hard_alpha_cap=True, want_weights=True, want_alphas=True, want_invalid=True)
            e1.record()
            torch.cuda.synchronize()
        d = dbg.view(-1, 8).double().cpu()
        d = d[d[:, 6] > 0]
        iters = r_.shape[0] / d.shape[0]
        per = d[:, :6].mean(0) / iters
        print(f"{name} ablate={extra}: {e0.elapsed_time(e1):.3f} ms, waves {d.shape[0]}, iterations/wave {iters:.1f}, wave lifetime {d[:,6].mean():.0f} ticks (max {d[:,6].max():.0f})")
        print("   ticks per iteration by section:", " ".join(f"{x:8.1f}" for x in per.tolist()), " sum", f"{per.sum():.1f}")
