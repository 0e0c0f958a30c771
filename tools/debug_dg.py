"""Debug: dG of the RE10K golden case through bts_render_bwd; saves it to argv[1] (run once per BTS_BWD_DIRECT_ATOMICS setting)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from behindthescenes_amd import native
from tests._cases import Case
from tests._hip_helpers import net_from_case
c = Case(sys.argv[2] if len(sys.argv) > 2 else "re10k_train")
net = net_from_case(c, train=True)
ft = net.native_field()
params = net.mlp_coarse.packed().detach()
rays = c.rays.reshape(-1, 8).cuda().contiguous(); z = c.z_samp.cuda().contiguous()
out = native.render_fwd(ft, params, rays, z, hard_alpha_cap=c.hard_cap, want_saved=True)
d_proj, d_mlp, _ = native.render_bwd(ft, params, rays, z, out["sigma_raw"], out["trans"], hard_alpha_cap=c.hard_cap,
                                     g_rgb=c.t["gin_rgb"].cuda().contiguous(), g_depth=c.t["gin_depth"].cuda().contiguous())
torch.cuda.synchronize()
torch.save(dict(d_proj=d_proj.cpu(), d_mlp=d_mlp.cpu()), sys.argv[1])
if len(sys.argv) > 3:
    a = torch.load(sys.argv[3])
    d = (a["d_proj"] - d_proj.cpu())
    print("max |d_proj| ref", a["d_proj"].abs().max().item(), "max diff", d.abs().max().item())
    bad = (d.abs() > 1e-4 * a["d_proj"].abs().max()).nonzero()
    print("bad entries", bad.shape[0], "of", d.numel())
    print(bad[:40].tolist())
    n, H, W, HD = d.shape
    per_texel = d.abs().amax(dim=-1)
    print("texels with error:", (per_texel > 1e-4 * a["d_proj"].abs().max()).nonzero()[:40].tolist())
    for idx in bad[:12].tolist():
        print(idx, "ref", a["d_proj"][tuple(idx)].item(), "new", d_proj.cpu()[tuple(idx)].item())
    print("sum ref", a["d_proj"].sum().item(), "sum new", d_proj.sum().item())
