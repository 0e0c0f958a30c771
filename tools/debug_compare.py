import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import behindthescenes_amd as bts
from behindthescenes_amd import native
from tests._cases import Case
from tests._hip_helpers import net_from_case
name = sys.argv[1] if len(sys.argv) > 1 else "kitti_eval"
c = Case(name)
net = net_from_case(c)
ft = net.native_field()
mlp = net.mlp_coarse.packed().detach()
rays, z = c.rays.reshape(-1, 8).cuda(), c.z_samp.cuda()
kw = dict(hard_alpha_cap=c.hard_cap, want_weights=True, want_alphas=True, want_saved=True)
os.environ.pop("BTS_RENDER_V1", None)
a = native.render_fwd(ft, mlp, rays, z, **kw)
os.environ["BTS_RENDER_V1"] = "1"
b = native.render_fwd(ft, mlp, rays, z, **kw)
print(name, "rays", rays.shape, "K", z.shape[1], "nv", ft.nv, "n", ft.n)
for k in ("depth", "rgb", "weights", "alphas", "sigma_raw", "trans", "invalid"):
    d = (a[k] - b[k]).abs()
    print(f"{k:10s} max diff {d.max().item():.3e}  frac>1e-4 {(d > 1e-4).float().mean().item():.4f}")
d = (a["sigma_raw"] - b["sigma_raw"]).abs()
bad = (d > 1e-3).nonzero()
print("bad sigma samples:", bad.shape[0], "of", d.numel())
if bad.shape[0]:
    r, k = bad[:, 0], bad[:, 1]
    print(" ray idx (first 20):", r[:20].tolist())
    print(" k idx   (first 20):", k[:20].tolist())
    print(" k histogram (lo32 / hi32):", (k < 32).sum().item(), (k >= 32).sum().item())
    print(" k histogram by 8:", torch.bincount(k // 8, minlength=8).tolist())
    rr = r.unique()
    print(" bad rays mod 64 hist by 8:", torch.bincount((rr % 64) // 8, minlength=8).tolist(), " first bad rays:", rr[:24].tolist())
    print(" distinct rays:", r.unique().numel())
    i = bad[0]
    print(" example:", a["sigma_raw"][i[0], i[1]].item(), b["sigma_raw"][i[0], i[1]].item())
