"""Debug: the pipelined query kernel against the oracle and against the profile-mode launch, error pattern by lane."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import bts_oracle as O
from tests._cases import Case
from tests._hip_helpers import net_from_case, build_net

for name in ("kitti_train", "re10k_train"):
    c = Case(name)
    net = net_from_case(c)
    pts = c.t["q_pts"].cuda()
    rgb, inv, sig = net(pts)
    ref = c.t["q_sigma"]
    e = (sig.cpu() - ref).abs().reshape(pts.shape[0], -1)
    print(name, "pts", tuple(pts.shape), "max err", float(e.max()), "n bad", int((e > 1e-4 * ref.abs().reshape(e.shape) + 1e-6).sum()))
    bad = (e > 1e-4 * ref.abs().reshape(e.shape) + 1e-6).nonzero()
    print("  bad idx (sample, point) first 20:", bad[:20].tolist(), " lanes:", sorted(set((bad[:, 1] % 64).tolist()))[:64])
    print("  inv equal frac", float((inv.cpu() == c.t["q_invalid"]).float().mean()), "rgb max err", float((rgb.cpu() - c.t["q_rgb"]).abs().max()))

cfg = O.FieldConfig(learn_empty=True)
g = torch.Generator().manual_seed(41)
scene = O.synthetic_scene(1, 2, 192, 640, 64, seed=41, intrinsics=O.K_KITTI360, baseline=0.6, smooth=True)
mlp = O.init_mlp(103, 64, 0, gen=g)
mlp.b_out = torch.tensor([-2.0])
empty = torch.randn(64, generator=g)
q = O.profile_points(x_res=64, z_res=32)
Y, Z, X, _ = q.shape
net = build_net(cfg, mlp, scene, [0, 1], empty_feature=empty)
pts = q.reshape(1, -1, 3).cuda().contiguous()
for rep in range(3):
    prof, sigma = net.occupancy_profile(pts, Y, want_sigma=True)
    _, _, sig_q = net(pts)
    _, _, sig_q2 = net(pts)
    d = (sig_q.reshape(-1) != sigma.reshape(-1))
    d2 = (sig_q.reshape(-1) != sig_q2.reshape(-1))
    print(f"rep {rep}: plain vs profile differ {int(d.sum())} of {d.numel()}; plain vs plain differ {int(d2.sum())}; max |diff| {float((sig_q.reshape(-1) - sigma.reshape(-1)).abs().max()):.3e}")
    if d.any():
        idx = d.nonzero()[:, 0]
        print("   first idx", idx[:16].tolist(), "lanes (plain)", sorted(set((idx % 64).tolist()))[:20], "levels (profile lane)", sorted(set((idx // (Z * X)).tolist()))[:20])
st = O.make_state(scene, [0, 1], cfg, empty)
with torch.no_grad():
    o_prof, o_sigma, o_inv = O.occupancy_profile(q, st, mlp, cfg)
for nme, s_ in (("plain", sig_q), ("profile", sigma)):
    e = (s_.reshape(-1).cpu() - o_sigma).abs()
    print(nme, "vs oracle: max", float(e.max()), "bad", int((e > 1e-4 * o_sigma.abs() + 1e-6).sum()))
