"""Section-ablation timing of the fused forward (probe build, -DBTS_PROBE) on the BASELINE configs[1] workload.
    BTS_ALLOW_LIB_OVERRIDE=1 BTS_RENDER_LIB=behindthescenes_amd/variants/libbts_probe.so python tools/ablate_probe.py [rounds]
Bits: 1 = no G gather/blend, 2 = no sincos, 4 = no MFMA, 8 = no colour taps, 16 = no per-sample stores, 32 = no lin_out.
Marginal cost of a section = t(0) - t(bit); results are NOT valid renders."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BTS_RENDER_LIB", os.path.join(ROOT, "behindthescenes_amd", "variants", "libbts_probe.so"))
os.environ.setdefault("BTS_ALLOW_LIB_OVERRIDE", "1")
import torch
import behindthescenes_amd as bts
from behindthescenes_amd import native
from behindthescenes_amd import synthetic as S

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
H, W, K, V = 192, 640, 64, 2
scene = S.synthetic_scene(1, V, H, W, 64, seed=1, intrinsics=S.K_KITTIRAW)
net = S.build_net(scene, 64, 0, [0])
ft = net.native_field()
params = net.mlp_coarse.packed().detach()
rays_all = bts.ImageRaySampler(3.0, 80.0, H, W).sample(None, scene["poses"].cuda(), scene["projs"].cuda())[0].reshape(V, -1, 8)
sets = {"both": rays_all.reshape(-1, 8).contiguous(), "view0": rays_all[0].contiguous(), "view1": rays_all[1].contiguous()}
zs = {k: native.sample_coarse(r, torch.rand(r.shape[0], K, device="cuda"), True) for k, r in sets.items()}
masks = [0, 1, 2, 4, 8, 16, 32, 1 | 2, 1 | 4, 2 | 4, 1 | 2 | 4, 1 | 2 | 4 | 8 | 16 | 32]
res = {}
for r in range(rounds + 1):
    for name in sets:
        for m in masks:
            if name != "both" and m not in (0, 1, 2, 4, 1 | 2 | 4):
                continue
            os.environ["BTS_ABLATE"] = str(m)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            native.render_fwd(ft, params, sets[name], zs[name], hard_alpha_cap=True, want_weights=True, want_alphas=True, want_invalid=True)
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                res.setdefault((name, m), []).append(e0.elapsed_time(e1))
for (name, m), ts in res.items():
    ts = sorted(ts)
    print(f"{name:6s} ablate={m:3d}: median {ts[len(ts)//2]:.3f} ms  min {ts[0]:.3f} ms  ({sets[name].shape[0]} rays)")
