import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from behindthescenes_amd import native
from tests._cases import Case
from tests._hip_helpers import net_from_case
name = sys.argv[1] if len(sys.argv) > 1 else "kitti_train"
c = Case(name)
net = net_from_case(c)
ft = net.native_field()
mlp = net.mlp_coarse.packed().detach()
rays, z = c.rays.reshape(-1, 8).cuda(), c.z_samp.cuda()
kw = dict(hard_alpha_cap=c.hard_cap, want_weights=True, want_alphas=True, want_saved=True, want_rgb_samps=True)
runs = [native.render_fwd(ft, mlp, rays, z, **kw) for _ in range(4)]
torch.cuda.synchronize()
for k in ("sigma_raw", "trans", "rgb_samps", "invalid"):
    d = torch.stack([(runs[0][k] - r[k]).abs().flatten(1).max(1).values for r in runs[1:]]).max(0).values
    print(f"{k:10s}: rays differing between runs: {(d > 0).sum().item()} (first {(d > 0).nonzero()[:10, 0].tolist()})")
os.environ["BTS_RENDER_V1"] = "1"
ref = native.render_fwd(ft, mlp, rays, z, **kw)
for i, r in enumerate(runs):
    print("run", i, "bad-vs-v1 rays:", ((r["sigma_raw"] - ref["sigma_raw"]).abs() > 1e-3).any(1).sum().item(), " early-stored s_raw (in trans) bad rays:", ((r["trans"] - ref["sigma_raw"]).abs() > 1e-3).any(1).sum().item())
