"""Debug helper: gradients of the kitti_train golden case against the reference's, per block of w_in columns, plus a host-side check
of the backward's workspace (the per-channel gate masks must be the transpose of the per-sample masks)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import behindthescenes_amd as bts
from behindthescenes_amd import native
from tests._cases import Case
from tests._hip_helpers import net_from_case

c = Case("kitti_train")
net = net_from_case(c, train=True)
net.encode(c.scene["images"].cuda(), c.scene["projs"].cuda(), c.scene["poses"].cuda(), ids_encoder=[0], ids_render=c.meta["ids_render"])
K = c.meta["K"]
renderer = bts.NeRFRenderer(n_coarse=K, lindisp=True, hard_alpha_cap=c.hard_cap).cuda()
g_rgb, g_depth = c.t["gin_rgb"].cuda(), c.t["gin_depth"].cuda()
params = [net.mlp_coarse.lin_in.weight, net.mlp_coarse.lin_in.bias, net.mlp_coarse.lin_out.weight, net.mlp_coarse.lin_out.bias]
rays = c.rays.reshape(-1, 8).cuda()
w, rgb, depth, a, inv, _, rs = renderer.composite(net, rays, c.z_samp.cuda(), sb=c.rays.shape[0], want_rgb_samps=True)
((rgb * g_rgb).sum() + (depth * g_depth).sum()).backward()
torch.cuda.synchronize()
gw = params[0].grad.cpu()
ref = c.t["g_w_in"].view_as(gw)
C = gw.shape[1] - 39
mx = ref.abs().max()
print("w_in grad: features", ((gw[:, :C] - ref[:, :C]).abs().max() / mx).item())
for j in range(39):
    e = ((gw[:, C + j] - ref[:, C + j]).abs().max() / mx).item()
    if e > 1e-4:
        print(f"  pe column {j}: err {e:.3e}   ours {gw[:3, C + j].tolist()}  ref {ref[:3, C + j].tolist()}")
print("b_in", ((params[1].grad.cpu() - c.t["g_b_in"].view_as(params[1].grad.cpu())).abs().max() / c.t["g_b_in"].abs().max()).item())
print("w_out", ((params[2].grad.cpu() - c.t["g_w_out"].view_as(params[2].grad.cpu())).abs().max() / c.t["g_w_out"].abs().max()).item())
# workspace: gs (rays, K) | masks (rays, HT, K) | pmask (rays, HD, 2)
ws = list(native._WS.values())[0]
R, HD = rays.shape[0], 64
HT = HD // 32
gs = ws[:R * K].view(R, K).cpu()
masks = ws[R * K:R * K * (1 + HT)].view(torch.int32).view(R, HT, K).cpu()
pm = ws[R * K * (1 + HT):R * K * (1 + HT) + R * HD * 2].view(torch.int32).view(R, HD, 2).cpu()
bad = 0
for r in range(min(R, 64)):
    for ch in range(HD):
        for k in range(K):
            bit_s = (int(masks[r, ch // 32, k]) >> (ch % 32)) & 1
            bit_c = (int(pm[r, ch, k // 32]) >> (k % 32)) & 1
            bad += bit_s != bit_c
print("mask / pmask mismatches in the first rays:", bad, " (gs nonzero:", int((gs != 0).sum()), "of", gs.numel(), ")")
