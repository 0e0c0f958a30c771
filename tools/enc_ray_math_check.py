"""CPU check of the MATH behind the experimental encoder-camera ray path (csrc/bts_render_kernel.h, -DBTS_ENC_RAY): for rays through
the encoder camera, lin_in's output h_k = per-ray constant + six first-order rows + the depth-code slice, emulated here in fp32 torch,
against the oracle's per-sample evaluation (fp32) and an fp64 evaluation.  No GPU, no kernel: it answers "is the first-order
form as accurate as the reference's own fp32 arithmetic?" before GPU minutes are spent on the kernel.
    python tools/enc_ray_math_check.py"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import bts_oracle as O   # a probe about the oracle's numbers, not product code

torch.manual_seed(0)
H, W, C, HD, K = 192, 640, 64, 64, 64
cfg = O.FieldConfig()
scene = O.synthetic_scene(1, 2, H, W, C, seed=21, intrinsics=O.K_KITTIRAW, smooth=True)
mlp = O.init_mlp(C + 39, HD, 0, gen=torch.Generator().manual_seed(7))
st = O.make_state(scene, [0], cfg)
rays = O.image_rays(scene["poses"][:, :1], scene["projs"][:, :1], H, W, cfg.d_min, cfg.d_max)[0]      # view 0 = the encoder camera
sel = torch.randperm(rays.shape[0], generator=torch.Generator().manual_seed(1))[:4000]
rays = rays[sel]
z = O.sample_coarse(rays, K, True, torch.rand(rays.shape[0], K))
xyz = (rays[:, None, :3] + z[..., None] * rays[:, None, 3:6]).reshape(1, -1, 3)


def lin_in(dtype):
    s = O.FieldState(*[t.to(dtype) for t in (st.feat, st.K_enc, st.w2c_enc, st.imgs, st.K_r, st.w2c_r)])
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        x, inv = O.sample_features(xyz.to(dtype), s, cfg)
    finally:
        torch.set_default_dtype(old)
    return torch.nn.functional.linear(x, mlp.w_in.to(dtype), mlp.b_in.to(dtype))[0].view(-1, K, HD), x[0].view(-1, K, C + 39), inv[0].view(-1, K)


h32, x32, inv = lin_in(torch.float32)
h64, _, _ = lin_in(torch.float64)

# ---- emulation in fp32
xy = x32[..., C:C + 2]                    # per-sample image coordinates as the reference computed them
zn = x32[..., C + 2]
ff = cfg.freq_factor
ix = (((xy[..., 0] + 1) * W - 1) / 2).clamp(0, W - 1)
iy = (((xy[..., 1] + 1) * H - 1) / 2).clamp(0, H - 1)
G = torch.einsum("chw,oc->hwo", st.feat[0], mlp.w_in[:, :C])             # projected feature map (H, W, HD)
w_pe = mlp.w_in[:, C:]                                                    # reference order: x, y, z, then per octave sin(3) cos(3)
ok = (~inv.any(1)) & ((ix - ix[:, :1]).abs().amax(1) <= 2 ** -11) & ((iy - iy[:, :1]).abs().amax(1) <= 2 ** -11) \
    & ((xy[..., 0] - xy[:, :1, 0]).abs().amax(1) <= 2 ** -18) & ((xy[..., 1] - xy[:, :1, 1]).abs().amax(1) <= 2 ** -18) \
    & (ix[:, 0] >= 1.5) & (ix[:, 0] <= W - 2.5) & (iy[:, 0] >= 1.5) & (iy[:, 0] <= H - 2.5)
print(f"rays taking the path: {int(ok.sum())} of {ok.numel()}  (spread ix max {float((ix - ix[:, :1]).abs().max()):.2e} px, x {float((xy[..., 0] - xy[:, :1, 0]).abs().max()):.2e})")


def axis_ref(pr):
    nb = torch.round(pr)
    near = (pr - nb).abs() <= 2 ** -10
    fl = torch.floor(pr)
    fr = pr - fl
    c0 = torch.where(near, nb - 1, fl).long()
    p0 = torch.where(near, nb, pr)
    zero, one = torch.zeros_like(pr), torch.ones_like(pr)
    V = torch.stack([torch.where(near, zero, 1 - fr), torch.where(near, one, fr), zero], -1)
    DL = torch.stack([-one, one, zero], -1)
    DR = torch.stack([torch.where(near, zero, -one), torch.where(near, -one, one), torch.where(near, one, zero)], -1)
    return c0, p0, V, DL, DR, near


cx, px0, Vx, DLx, DRx, nearx = axis_ref(ix[:, 0])
cy, py0, Vy, DLy, DRy, neary = axis_ref(iy[:, 0])
print(f"reference point on a texel boundary: x {float(nearx[ok].float().mean()):.2f}, y {float(neary[ok].float().mean()):.2f} of the rays")
B = rays.shape[0]
g = torch.stack([torch.stack([G[(cy + a).clamp(0, H - 1), (cx + b).clamp(0, W - 1)] for b in range(3)], 1) for a in range(3)], 1)  # (B,3,3,HD)
rowV = (Vx[:, None, :, None] * g).sum(2)
rowL = (DLx[:, None, :, None] * g).sum(2)
rowR = (DRx[:, None, :, None] * g).sum(2)
f0 = (Vy[..., None] * rowV).sum(1)
uxm, uxp = (Vy[..., None] * rowL).sum(1), (Vy[..., None] * rowR).sum(1)
uym, uyp = (DLy[..., None] * rowV).sum(1), (DRy[..., None] * rowV).sum(1)
xr, yr = xy[:, 0, 0], xy[:, 0, 1]
p0 = mlp.b_in[None] + xr[:, None] * w_pe[:, 0][None] + yr[:, None] * w_pe[:, 1][None]
vx = (w_pe[:, 0] / ff)[None].repeat(B, 1)
vy = (w_pe[:, 1] / ff)[None].repeat(B, 1)
Pc = torch.tensor(math.pi * 0.5, dtype=torch.float32)
for oct in range(6):
    po = 2.0 ** oct
    for axis, (a0, v) in enumerate(((xr, vx), (yr, vy))):
        arg = a0 * (ff * po)
        w_s, w_c = w_pe[:, 3 + 6 * oct + axis], w_pe[:, 3 + 6 * oct + 3 + axis]
        p0 = p0 + torch.sin(arg)[:, None] * w_s[None] + torch.sin(arg + Pc)[:, None] * w_c[None]
        v += po * (torch.cos(arg)[:, None] * w_s[None] - torch.sin(arg)[:, None] * w_c[None])
base = f0 + p0                                                                        # (B, HD)
dx, dy = ix - px0[:, None], iy - py0[:, None]
dax = xy[..., 0] * ff - (xr * ff)[:, None]
day = xy[..., 1] * ff - (yr * ff)[:, None]
corr = dx.clamp(max=0)[..., None] * uxm[:, None] + dx.clamp(min=0)[..., None] * uxp[:, None] \
    + dy.clamp(max=0)[..., None] * uym[:, None] + dy.clamp(min=0)[..., None] * uyp[:, None] \
    + dax[..., None] * vx[:, None] + day[..., None] * vy[:, None]
zpart = zn[..., None] * w_pe[:, 2][None, None]
for oct in range(6):
    arg = zn * (ff * 2.0 ** oct)
    zpart = zpart + torch.sin(arg)[..., None] * w_pe[:, 3 + 6 * oct + 2] + torch.sin(arg + Pc)[..., None] * w_pe[:, 3 + 6 * oct + 5]
h_em = base[:, None] + corr + zpart
h_nocorr = base[:, None] + zpart

for name, hh in (("oracle fp32 (the reference's arithmetic)", h32), ("first-order emulation", h_em), ("per-ray constant only (the backed-out shortcut)", h_nocorr)):
    e = (hh.double() - h64)[ok].abs()
    print(f"{name:50s} vs fp64: max {float(e.max()):.2e}  rms {float(e.square().mean().sqrt()):.2e}")
e = (h_em - h32)[ok].abs()
print(f"{'first-order emulation vs oracle fp32':50s}         max {float(e.max()):.2e}  rms {float(e.square().mean().sqrt()):.2e}")
e = (h_nocorr - h32)[ok].abs()
print(f"{'per-ray constant only vs oracle fp32':50s}         max {float(e.max()):.2e}  rms {float(e.square().mean().sqrt()):.2e}")

# where are the largest deviations?
d = (h_em - h32).abs() * ok[:, None, None]
top = torch.topk(d.flatten(), 8)
for v, i in zip(top.values.tolist(), top.indices.tolist()):
    r, k, hid = i // (K * HD), (i // HD) % K, i % HD
    print(f"  err {v:.2e} ray {r} k {k} hid {hid}: ix0 {float(px0[r]):.5f} dx {float(dx[r, k]):.2e} dy {float(dy[r, k]):.2e} dax {float(dax[r, k]):.2e} "
          f"|uxp| {float(uxp[r].abs().max()):.2f} |h| {float(h32[r, k, hid]):.2f} oracle-vs-fp64 here {float((h32[r, k, hid].double() - h64[r, k, hid]).abs()):.2e}")
