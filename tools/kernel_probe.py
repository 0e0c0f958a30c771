"""Times variants of the fused forward on the BASELINE configs[1] workload (one process, interleaved rounds).
usage: python tools/kernel_probe.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import behindthescenes_amd as bts
from behindthescenes_amd import native
from behindthescenes_amd import synthetic as S

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
only = sys.argv[2] if len(sys.argv) > 2 else None
H, W, K, V = 192, 640, 64, 2
scene = S.synthetic_scene(1, V, H, W, 64, seed=1, intrinsics=S.K_KITTIRAW)
net = S.build_net(scene, 64, 0, [0])
ft = net.native_field()
feat_nhwc = native.nchw_to_nhwc(net.grid_f_features[0][:, 0].detach().contiguous())
ft_direct = native.FieldTensors(net.spec, None, ft.K_enc, ft.w2c_enc, ft.imgs_nhwc4, ft.K_r, ft.w2c_r, None, feat_nhwc=feat_nhwc)
params = net.mlp_coarse.packed().detach()
rays = bts.ImageRaySampler(3.0, 80.0, H, W).sample(None, scene["poses"].cuda(), scene["projs"].cuda())[0].reshape(-1, 8).contiguous()
z = native.sample_coarse(rays, torch.rand(rays.shape[0], K, device="cuda"), True)
variants = {
    "proj__all-out": (ft, dict(want_weights=True, want_alphas=True, want_invalid=True)),
    "v1____all-out": (ft, dict(want_weights=True, want_alphas=True, want_invalid=True)),
    "f32mfma_all  ": (ft, dict(want_weights=True, want_alphas=True, want_invalid=True)),
    "proj__no-wa  ": (ft, dict(want_invalid=True)),
    "proj__no-out ": (ft, dict(want_invalid=False)),
    "direct all   ": (ft_direct, dict(want_weights=True, want_alphas=True, want_invalid=True)),
    "direct no out": (ft_direct, dict(want_invalid=False)),
}
if only:
    variants = {k: v for k, v in variants.items() if k.startswith(only)}
res = {k: [] for k in variants}
for r in range(rounds + 1):
    for name, (f, kw) in variants.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for key, pref in (("BTS_RENDER_V1", "v1"), ("BTS_RENDER_F32MFMA", "f32mfma")):
            if name.startswith(pref):
                os.environ[key] = "1"
            else:
                os.environ.pop(key, None)
        e0.record()
        native.render_fwd(f, params, rays, z, hard_alpha_cap=True, **kw)
        e1.record()
        torch.cuda.synchronize()
        if r > 0:
            res[name].append(e0.elapsed_time(e1))
for name, ts in res.items():
    ts = sorted(ts)
    print(f"{name}: median {ts[len(ts)//2]:.3f} ms  min {ts[0]:.3f} ms   -> {rays.shape[0] / ts[len(ts)//2] / 1e3:.1f} M rays/s")
