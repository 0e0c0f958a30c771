"""The timed kernel of bench.py on its own: bts_render_fwd on the BASELINE configs[1] workload (192x640, 2 views, 245 760 rays x 64
samples, learn_empty=True as eval_depth.yaml runs it, want_weights + alphas + invalid) -- the process tools/profile.sh wraps in
rocprofv3.   usage: python tools/kernel_probe.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import behindthescenes_amd as bts
from behindthescenes_amd import native
from behindthescenes_amd import synthetic as S

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
H, W, K, V = 192, 640, 64, 2
torch.manual_seed(4242)
scene = S.synthetic_scene(1, V, H, W, 64, seed=1000, intrinsics=S.K_KITTIRAW)
net = S.build_net(scene, 64, 0, [0], learn_empty=True)
ft = net.native_field()
params = net.mlp_coarse.packed().detach()
rays = bts.ImageRaySampler(3.0, 80.0, H, W).sample(None, scene["poses"].cuda(), scene["projs"].cuda())[0].reshape(-1, 8).contiguous()
u = torch.rand(rays.shape[0], K, device="cuda")   # sample_coarse runs inside the kernel (BtsRenderArgs.jitter), as bench.py's step does
ts = []
for r in range(rounds + 1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    native.render_fwd(ft, params, rays, None, jitter=u, lindisp=True, hard_alpha_cap=True, want_weights=True, want_alphas=True, want_invalid=True)
    e1.record()
    torch.cuda.synchronize()
    if r > 0:
        ts.append(e0.elapsed_time(e1))
ts.sort()
print(f"render_fwd: median {ts[len(ts) // 2]:.3f} ms  min {ts[0]:.3f} ms  -> {rays.shape[0] / ts[len(ts) // 2] / 1e3:.1f} M rays/s")
