"""GPU: gradients of the HIP renderer (bts_render_bwd + bts_project_features_bwd through torch.autograd.Function) against
(a) the gradients torch.autograd produced for the REAL reference (tests/golden/*_train.npz) and (b) autograd through the
CPU oracle on fresh inputs.  Tolerance: 1e-4 of the largest gradient entry of each tensor; float atomics
make the summation order run-to-run non-deterministic, like the reference's own CUDA grid_sample backward."""
import pytest
import torch

from oracle import bts_oracle as O
from tests._cases import Case

pytestmark = pytest.mark.gpu
GRAD_RTOL = 1e-4
DEPTH_RTOL_MS = 1e-4


@pytest.fixture(scope="module")
def hip():
    import behindthescenes_amd as bts
    from behindthescenes_amd import _lib
    assert torch.cuda.is_available()
    _lib.load()
    return bts


def _rel_to_max(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


def _hip_grads(hip, net, renderer, rays, z, sb, loss_fn, want_rgb_samps=True):
    params = [net.mlp_coarse.lin_in.weight, net.mlp_coarse.lin_in.bias]
    for blk in net.mlp_coarse.blocks:
        params += [blk.fc_0.weight, blk.fc_0.bias, blk.fc_1.weight, blk.fc_1.bias]
    params += [net.mlp_coarse.lin_out.weight, net.mlp_coarse.lin_out.bias, net.encoder.feats[0]]
    if net.learn_empty:
        params.append(net.empty_feature)
    for p in params:
        p.grad = None
    # want_rgb_samps: the backward then reads the forward's per-sample colours instead of re-tapping the colour frames
    w, rgb, depth, a, inv, _, rs = renderer.composite(net, rays, z, sb=sb, want_rgb_samps=want_rgb_samps)
    loss_fn(w, rgb, depth, a).backward()
    return [p.grad for p in params]


@pytest.mark.parametrize("keep_colours", [True, False])
@pytest.mark.parametrize("name", ["kitti_train", "re10k_train"])
def test_gradients_vs_reference_golden(hip, name, keep_colours):
    """kitti_train: lin_in -> lin_out; re10k_train: one ResnetBlockFC in between (fc_0 / fc_1 weight and bias gradients too)."""
    from tests._hip_helpers import net_from_case
    c = Case(name)
    net = net_from_case(c, train=True)
    net.encode(c.scene["images"].cuda(), c.scene["projs"].cuda(), c.scene["poses"].cuda(), ids_encoder=[0], ids_render=c.meta["ids_render"])
    renderer = hip.NeRFRenderer(n_coarse=c.meta["K"], lindisp=True, hard_alpha_cap=c.hard_cap).cuda()
    g_rgb, g_depth = c.t["gin_rgb"].cuda(), c.t["gin_depth"].cuda()
    grads = _hip_grads(hip, net, renderer, c.rays.reshape(-1, 8).cuda(), c.z_samp.cuda(), c.rays.shape[0],
                       lambda w, rgb, depth, a: (rgb * g_rgb).sum() + (depth * g_depth).sum(), want_rgb_samps=keep_colours)
    nb = c.meta["nb"]
    names = ["g_w_in", "g_b_in"] + sum([[f"g_blk{i}_w0", f"g_blk{i}_b0", f"g_blk{i}_w1", f"g_blk{i}_b1"] for i in range(nb)], []) \
        + ["g_w_out", "g_b_out", "g_feat"]
    assert len(grads) == len(names)
    for g, nme in zip(grads, names):
        assert g is not None, nme
        assert _rel_to_max(g, c.t[nme].view_as(g.cpu())) <= GRAD_RTOL, (nme, _rel_to_max(g, c.t[nme].view_as(g.cpu())))


def _oracle_grads(scene, mlp, cfg, ids_render, rays, z, n, hard_cap, loss_fn, empty=None, dtype=torch.float32, device="cpu",
                  loss_fns=None):
    """torch.autograd through the oracle.  dtype=float64: the same formulas in double, the arbiter between two fp32 evaluations;
    device="cuda": the oracle's torch ops run eagerly on the GPU (used for the fp64 arbiter at the real training shapes, where a
    CPU fp64 pass would take minutes -- as "truth" its device does not matter).  `loss_fns` (a list) returns one gradient tuple per
    loss from ONE forward graph."""
    if dtype == torch.float64:   # fp64 evaluation of the same formulas: the arbiter
        torch.set_default_dtype(torch.float64)
        try:
            scene64 = {k: v.double() for k, v in scene.items()}
            mlp64 = O.MlpParams(mlp.w_in.double(), mlp.b_in.double(), [tuple(t.double() for t in b) for b in mlp.blocks],
                                mlp.w_out.double(), mlp.b_out.double())
            return _oracle_grads(scene64, mlp64, cfg, ids_render, rays.double(), z.double(), n, hard_cap, loss_fn,
                                 None if empty is None else empty.double(), dtype=None, device=device, loss_fns=loss_fns)
        finally:
            torch.set_default_dtype(torch.float32)
    params = [t.clone().to(device).requires_grad_(True) for t in mlp.tensors()]
    feat = scene["feat"].clone().to(device).requires_grad_(True)
    nb = len(mlp.blocks)
    blocks = [tuple(params[2 + 4 * i: 6 + 4 * i]) for i in range(nb)]
    m = O.MlpParams(params[0], params[1], blocks, params[-2], params[-1])
    st = O.make_state(scene, ids_render, cfg)
    e = None if empty is None else empty.clone().to(device).requires_grad_(True)
    st = O.FieldState(feat, *[t.to(device) for t in (st.K_enc, st.w2c_enc, st.imgs, st.K_r, st.w2c_r)], e)
    w, rgb, depth, a, *_ = O.composite(rays.reshape(-1, 8).to(device), z.to(device), n, st, m, cfg, hard_alpha_cap=hard_cap)
    wanted = params + [feat] + ([e] if e is not None else [])
    if loss_fns is None:
        return tuple(g.cpu() for g in torch.autograd.grad(loss_fn(w, rgb, depth, a), wanted))
    return [tuple(g.cpu() for g in torch.autograd.grad(f(w, rgb, depth, a), wanted, retain_graph=i + 1 < len(loss_fns)))
            for i, f in enumerate(loss_fns)]


def _gate_safe_rays(scene, mlp, cfg, ids_render, rays, z, empty, margin, stats=None):
    """Rays none of whose samples has a hidden pre-activation within its fp32 uncertainty of the relu kink (fp64 evaluation).  On
    every other ray two correct fp32 evaluations may gate a unit differently, and ONE flipped gate moves a few gradient entries by
    ~1e-4..1e-3 of the largest entry -- an effect of the test point, not of either implementation.  Measured on the KITTI-360
    training shape against an fp64 evaluation: the fp32 ORACLE is 4e-4 off at one such point (sample 4), the HIP path 2e-3 at
    another (sample 14, ray 1573, k = 17: unit 63 has h = -3.3e-5 there), every other ray agrees to 1e-6.
    What makes h uncertain in ANY fp32 evaluation:
      * the tap position: ix = ((x + 1) W - 1) / 2 carries a few ulp of (x + 1) W / 2, i.e. ~4e-5 px at W = 640, and h moves by
        |dh / d ix| times that -- the dominant term wherever the feature map has a slope (3e-5 at the point above);
      * the encoding: sin(f x) up to f = 48 with x = q.x / q.z good to ~2e-7 |x|: ~6e-6 |x|, large only far outside the frustum.
    margin_unit = margin * (1 + max(|x|, |y|, |code|)) + |W_f (f(ix + e, iy) - f(ix, iy))| + |W_f (f(ix, iy + e) - f)|, e = 6e-5 px,
    pushed (in quadrature) through the ResnetBlockFC layers.   Returns (mask (n, B'), #rays excluded)."""
    import torch.nn.functional as F
    torch.set_default_dtype(torch.float64)
    try:
        st = O.make_state({k: v.double() for k, v in scene.items()}, ids_render, cfg, None if empty is None else empty.double())
        n, Bp = rays.shape[:2]
        C, Hh, Ww = st.feat.shape[1:]
        r = rays.double().reshape(-1, 8)
        pts = (r[:, None, :3] + z.double().unsqueeze(2) * r[:, None, 3:6]).reshape(n, -1, 3)
        x, _ = O.sample_features(pts, st, cfg)
        xy = O.project(pts, st.w2c_enc.unsqueeze(1), st.K_enc.unsqueeze(1))[0][:, 0]   # (n, P, 2)
        e_px = 6e-5
        f0 = O._bilinear_border(st.feat, xy)
        w_f = mlp.w_in.double()[:, :C]
        dh = F.linear(O._bilinear_border(st.feat, xy + torch.tensor([2 * e_px / Ww, 0.0])) - f0, w_f).abs() \
            + F.linear(O._bilinear_border(st.feat, xy + torch.tensor([0.0, 2 * e_px / Hh])) - f0, w_f).abs()
        m = margin * (1.0 + x[..., C:C + 3].abs().amax(-1, keepdim=True)) + dh          # (n, B'*K, Hd)
        h = F.linear(x, mlp.w_in.double(), mlp.b_in.double())
        near = torch.zeros(h.shape[:-1], dtype=torch.bool)
        n_units = n_near = 0
        for (w0, b0, w1, b1) in mlp.blocks:
            t = F.linear(torch.relu(h), w0.double(), b0.double())
            mt = F.linear(m * m, w0.double() ** 2).sqrt() + margin      # independent errors add in quadrature
            near |= (h.abs() < m).any(-1) | (t.abs() < mt).any(-1)
            n_units += h.numel() + t.numel()
            n_near += int((h.abs() < m).sum()) + int((t.abs() < mt).sum())
            h = h + F.linear(torch.relu(t), w1.double(), b1.double())
            m = (m * m + F.linear(mt * mt, w1.double() ** 2)).sqrt()
        near |= (h.abs() < m).any(-1)
        n_units += h.numel()
        n_near += int((h.abs() < m).sum())
        if stats is not None:   # how many relu inputs (unit x sample) sit inside their uncertainty band
            stats["units"] = stats.get("units", 0) + n_units
            stats["near"] = stats.get("near", 0) + n_near
    finally:
        torch.set_default_dtype(torch.float32)
    safe = ~near.view(n, Bp, -1).any(-1)
    return safe, int((~safe).sum())


@pytest.mark.parametrize("C,K", [(64, 64), (64, 80), (32, 70), (64, 130)])
@pytest.mark.parametrize("learn_empty", [False, True])
def test_gradients_vs_oracle_autograd(hip, learn_empty, C, K):
    """Training-like shape in the small (n=3, nv=4, ragged ray count), a loss that also feeds gradient into `weights` and
    `alphas` (the alpha / entropy regularisers of loss.py use them), with and without learn_empty.  K = 64: the gate-bit passes;
    K = 80 / 70 / 130 (plain MLP of width 64 / 32, one and two ragged extra chunks): the row passes of bts_bwd_blocks.hip with the
    suffix sum of the compositing gradient carried from chunk to chunk.  The strict bound -- every
    entry within 1e-4 of the largest one -- is asserted against an fp64 evaluation on rays that keep a margin from every relu kink
    (round 1 had ratcheted this test to 20e-4 "because of gate flips"; with those rays identified and set aside, and the two
    round-2 root causes fixed, the strict bound holds)."""
    from tests._hip_helpers import build_net
    from tests._cases import robust_ray_mask
    cfg = O.FieldConfig(learn_empty=learn_empty)
    g = torch.Generator().manual_seed(77)
    n, v, H, W, NR = 3, 5, 48, 160, 700
    scene = O.synthetic_scene(n, v, H, W, C, seed=77, intrinsics=O.K_KITTI360, smooth=True)
    mlp = O.init_mlp(C + 39, C, 0, gen=g)
    mlp.b_in = torch.randn(C, generator=g) * 0.1
    empty = torch.randn(C, generator=g) if learn_empty else None
    rays = O.image_rays(scene["poses"], scene["projs"], H, W, 3.0, 80.0)
    rays = rays[:, torch.randperm(rays.shape[1], generator=g)[:1600].sort().values].contiguous()
    z = O.sample_coarse(rays.reshape(-1, 8), K, True, torch.rand(n * 1600, K, generator=g))
    # keep rays whose samples stay clear of every frustum border (a flipped `invalid` flag -- 1-ulp effect on exact-border pixels,
    # see test_gpu_parity -- swaps the whole feature vector of that sample with learn_empty) and of every relu kink
    keep = robust_ray_mask(O.make_state(scene, [1, 2, 3, 4], cfg, empty), rays, z).view(n, -1)
    safe, n_gate = _gate_safe_rays(scene, mlp, cfg, [1, 2, 3, 4], rays, z, empty, margin=2e-5)
    print(f"rays excluded for a hidden unit within 2e-5 of the relu kink: {n_gate} of {n * 1600}")
    assert n_gate <= 0.6 * n * 1600
    keep = keep & safe
    NR = min(NR, int(keep.sum(1).min()))      # (long rays meet more kinks: fewer rays survive)
    assert NR >= 300, NR
    idx = torch.stack([torch.nonzero(keep[i])[:NR, 0] for i in range(n)])
    assert idx.shape == (n, NR)
    rays = torch.gather(rays, 1, idx.unsqueeze(-1).expand(-1, -1, 8)).contiguous()
    z = torch.gather(z.view(n, -1, K), 1, idx.unsqueeze(-1).expand(-1, -1, K)).reshape(-1, K).contiguous()
    c_rgb = torch.randn(n * NR, 12, generator=g)
    c_w = torch.randn(n * NR, K, generator=g) * 0.1

    def loss_fn(w, rgb, depth, a):
        dev, dt = rgb.device, rgb.dtype
        return (rgb * c_rgb.to(dev, dt)).sum() + 0.05 * depth.sum() + (w * c_w.to(dev, dt)).sum() + 0.01 * (a ** 2).sum()

    ref = _oracle_grads(scene, mlp, cfg, [1, 2, 3, 4], rays, z, n, True, loss_fn, empty)
    truth = _oracle_grads(scene, mlp, cfg, [1, 2, 3, 4], rays, z, n, True, loss_fn, empty, dtype=torch.float64)
    net = build_net(cfg, mlp, scene, [1, 2, 3, 4], empty_feature=empty, train=True)
    renderer = hip.NeRFRenderer(n_coarse=K, lindisp=True, hard_alpha_cap=True).cuda()
    ours = _hip_grads(hip, net, renderer, rays.reshape(-1, 8).cuda(), z.cuda(), n, loss_fn)
    names = ["w_in", "b_in", "w_out", "b_out", "feat"] + (["empty_feature"] if learn_empty else [])
    for a, b, t, nme in zip(ours, ref, truth, names):
        assert a is not None, nme
        t = t.view_as(a.cpu())
        mx_hip = (a.cpu().double() - t).abs().max().item() / t.abs().max().item()
        mx_ref = (b.view_as(a.cpu()).double() - t).abs().max().item() / t.abs().max().item()
        l2_hip = ((a.cpu().double() - t).norm() / t.norm()).item()
        l2_ref = ((b.view_as(a.cpu()).double() - t).norm() / t.norm()).item()
        print(f"{nme}: max err / max entry  HIP {mx_hip:.2e}  fp32 oracle {mx_ref:.2e};  L2  HIP {l2_hip:.2e}  fp32 oracle {l2_ref:.2e}")
        assert mx_hip <= 0.2 * GRAD_RTOL, (nme, mx_hip, mx_ref)     # measured: 4e-7 .. 6e-6 (the fp32 oracle: 1e-7 .. 6e-6)
        assert l2_hip <= max(2.0 * l2_ref, 0.1 * GRAD_RTOL), (nme, l2_hip, l2_ref)


def _patch_rays(scene, cfg, ids_loss, n_patches, K, g):
    """`n_patches` random 8x8 patches per sample in PatchRaySampler order (the backward's dG scatter relies on a 64-ray group being
    one patch) -> rays (n, n_patches*64, 8), z (n*n_patches*64, K)"""
    n, v, _, H, W = scene["images"].shape
    nl = len(ids_loss)
    all_rays = O.image_rays(scene["poses"][:, ids_loss], scene["projs"][:, ids_loss], H, W, cfg.d_min, cfg.d_max).view(n, nl, H, W, 8)
    pv = torch.randint(0, nl, (n, n_patches), generator=g)
    py = torch.randint(0, H - 8, (n, n_patches), generator=g)
    px = torch.randint(0, W - 8, (n, n_patches), generator=g)
    rays = torch.stack([torch.cat([all_rays[i, pv[i, j], py[i, j]:py[i, j] + 8, px[i, j]:px[i, j] + 8].reshape(64, 8) for j in range(n_patches)])
                        for i in range(n)]).contiguous()                               # (n, n_patches*64, 8)
    z = O.sample_coarse(rays.reshape(-1, 8), K, True, torch.rand(n * n_patches * 64, K, generator=g))
    return rays, z.contiguous()


def _patch_rays_with_kink_mask(scene, mlp, cfg, ids_loss, ids_render, n_patches, K, g, margin, stats=None):
    """_patch_rays + a per-ray mask that is 0 where some sample of the ray has a hidden unit within `margin` of the relu kink in an fp64
    evaluation (see _gate_safe_rays).  The masked tests multiply each ray's upstream gradient by the mask, so such rays contribute to
    NEITHER path's gradient while the kernels still run on the full, real ray layout.
    -> rays (n, n_patches*64, 8), z (n*n_patches*64, K), mask (n*n_patches*64,)"""
    n = scene["images"].shape[0]
    rays, z = _patch_rays(scene, cfg, ids_loss, n_patches, K, g)
    masks = []
    for i in range(n):   # one sample at a time: fp64 activations
        sc = {k: t[i:i + 1] for k, t in scene.items()}
        safe, _ = _gate_safe_rays(sc, mlp, cfg, ids_render, rays[i:i + 1], z.view(n, -1, K)[i], None, margin, stats)
        masks.append(safe.view(-1))
    mask = torch.cat(masks).float()
    print(f"rays whose upstream gradient is zeroed (hidden unit within {margin:g} of the relu kink): {int((1 - mask).sum())} of {mask.numel()}")
    return rays, z, mask


# ---------------------------------------------------------------------------------------------------------------
# Gradient parity at the REAL training shapes, two instruments on the same inputs (built once per shape):
#   * masked, strict: rays with a hidden unit inside its fp32 uncertainty of the relu kink get a zero upstream gradient in BOTH
#     paths; every entry of every gradient tensor must then agree with oracle autograd to 1e-4 of the tensor's largest entry;
#   * unmasked, fp64-arbitrated: EVERY ray keeps its upstream gradient; an fp64 evaluation of the same formulas
#     is the truth, and the HIP path may not be further from it than the fp32 oracle (= what the reference computes) is -- in the
#     max norm, in L2, and in the number of feature-map texels that are off by more than 1e-4 of the largest entry.  A wrong
#     scatter slot, a stale gate bit or a bad pair-collision round on ANY ray shows up here.
# ---------------------------------------------------------------------------------------------------------------
REAL_SHAPES = {
    # BASELINE configs[2]: bs 16, 8 frames (4 loss + 4 render views), 64 patches of 8x8 = 4096 rays per sample, K = 64, nv = 4
    "kitti360": dict(n=16, v=8, H=192, W=640, C=64, Hd=64, nb=0, K=64, ids_loss=[0, 1, 2, 3], ids_render=[4, 5, 6, 7], patches=64,
                     cfg=dict(), hard_cap=True, intr="K_KITTI360", baseline=0.6, seed=303, min_kept=0.4, b_in=True),
    # exp_re10k.yaml field (C = 32, one ResnetBlockFC of width 32, distance code, no alpha cap): K = 48 (the yaml) and 128 (BASELINE.json),
    # 256x384 frames, 1024 patch rays per sample, nv = 2
    "re10k_k48": dict(n=3, v=3, H=256, W=384, C=32, Hd=32, nb=1, K=48, ids_loss=[0], ids_render=[1, 2], patches=16,
                      cfg=dict(d_min=1.0, d_max=100.0, code_mode="distance"), hard_cap=False, intr="K_RE10K", baseline=0.2, seed=548,
                      min_kept=0.25, b_in=False),
    "re10k_k128": dict(n=3, v=3, H=256, W=384, C=32, Hd=32, nb=1, K=128, ids_loss=[0], ids_render=[1, 2], patches=16,
                       cfg=dict(d_min=1.0, d_max=100.0, code_mode="distance"), hard_cap=False, intr="K_RE10K", baseline=0.2, seed=628,
                       min_kept=0.25, b_in=False),
}
_REAL = {}


def _real_case(name):
    """Inputs, kink mask, oracle gradients (fp32 on the CPU, masked and unmasked upstream gradient from one graph) of a real shape."""
    if name in _REAL:
        return _REAL[name]
    s = REAL_SHAPES[name]
    cfg = O.FieldConfig(**s["cfg"])
    g = torch.Generator().manual_seed(s["seed"])
    scene = O.synthetic_scene(s["n"], s["v"], s["H"], s["W"], s["C"], seed=s["seed"], intrinsics=getattr(O, s["intr"]), baseline=s["baseline"],
                              smooth=True)
    mlp = O.init_mlp(s["C"] + 39, s["Hd"], s["nb"], gen=g)
    if s["b_in"]:
        mlp.b_in = torch.randn(s["Hd"], generator=g) * 0.1
    stats = {}
    rays, z, mask = _patch_rays_with_kink_mask(scene, mlp, cfg, s["ids_loss"], s["ids_render"], s["patches"], s["K"], g, margin=2e-5, stats=stats)
    B = s["n"] * s["patches"] * 64
    nv = len(s["ids_render"])
    c_rgb = torch.randn(B, nv * 3, generator=g)

    def make_loss(m):
        cm = c_rgb * m.unsqueeze(-1)

        def loss_fn(w, rgb, depth, a):
            return (rgb * cm.to(rgb.device, rgb.dtype)).sum() + 0.05 * (depth * m.to(depth.device, depth.dtype)).sum()
        return loss_fn

    masked, full = make_loss(mask), make_loss(torch.ones_like(mask))
    ref_masked, ref_full = _oracle_grads(scene, mlp, cfg, s["ids_render"], rays, z, s["n"], s["hard_cap"], None, loss_fns=[masked, full])
    names = ["w_in", "b_in"] + (["fc_0.w", "fc_0.b", "fc_1.w", "fc_1.b"] if s["nb"] else []) + ["w_out", "b_out", "feat"]
    _REAL[name] = dict(s=s, cfg=cfg, scene=scene, mlp=mlp, rays=rays, z=z, mask=mask, masked=masked, full=full, ref_masked=ref_masked,
                       ref_full=ref_full, names=names, stats=stats)
    return _REAL[name]


def _hip_real(hip, c, loss_fn):
    from tests._hip_helpers import build_net
    s = c["s"]
    net = build_net(c["cfg"], c["mlp"], c["scene"], s["ids_render"], train=True)
    renderer = hip.NeRFRenderer(n_coarse=s["K"], lindisp=True, hard_alpha_cap=s["hard_cap"]).cuda()
    return _hip_grads(hip, net, renderer, c["rays"].reshape(-1, 8).cuda(), c["z"].cuda(), s["n"], loss_fn)


@pytest.mark.parametrize("name", list(REAL_SHAPES))
def test_gradients_at_the_real_training_shapes_masked(hip, name):
    """bts_render_bwd (every pass) + bts_project_features_bwd at the real shapes against torch.autograd through the CPU oracle, rays
    with a relu-kink sample set aside (see above).  Bound: 1e-4 of the largest entry of each tensor.  The fraction of rays set aside
    is checked against what the geometry predicts and recorded (gpurun_out/grad_real_<name>.json)."""
    c = _real_case(name)
    s, mask = c["s"], c["mask"]
    kept = float(mask.mean())
    assert mask.numel() == s["n"] * s["patches"] * 64 and kept > s["min_kept"], kept
    # Cross-check of the masked fraction against the unit-level event it is built from: with p = P(a relu input sits inside its fp32
    # uncertainty band) measured over all (unit, sample) pairs and U relu inputs per ray, independent events would zero
    # 1 - (1 - p)^U of the rays; the events cluster along a ray (neighbouring samples see almost the same activations), so the
    # mask may only zero FEWER rays than that -- anything above it means rays are being set aside for another reason.
    p_unit = c["stats"]["near"] / c["stats"]["units"]
    units_per_ray = c["stats"]["units"] / mask.numel()
    bound = 1.0 - (1.0 - p_unit) ** units_per_ray
    print(f"{name}: P(relu input inside its band) {p_unit:.2e}, {units_per_ray:.0f} relu inputs per ray -> at most {bound:.3f} of the rays "
          f"zeroed if independent; zeroed: {1 - kept:.3f}")
    assert 1.0 - kept <= bound + 0.01, (1.0 - kept, bound)
    ours = _hip_real(hip, c, c["masked"])
    errs = {}
    for a, b, nme in zip(ours, c["ref_masked"], c["names"]):
        errs[nme] = _rel_to_max(a, b.view_as(a.cpu()))
        print(f"{name} {nme}: max err / max entry {errs[nme]:.2e}")
    _record(name, dict(masked=dict(rays=int(mask.numel()), kept_fraction=kept, zeroed_fraction_bound_if_independent=bound,
                                   p_relu_input_inside_band=p_unit, relu_inputs_per_ray=units_per_ray, max_err_over_max_entry=errs)))
    for nme, err in errs.items():
        assert err <= GRAD_RTOL, (nme, err)


def _record(name, d):
    """Numbers of the real-shape gradient tests for profiles/ (the test log itself is captured by pytest -q)."""
    import json
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(root, exist_ok=True)
    path = os.path.join(root, f"grad_real_{name}.json")
    old = json.load(open(path)) if os.path.exists(path) else {}
    old.update(d)
    json.dump(old, open(path, "w"), indent=1)


ARB_FACTOR = 1.5      # HIP may be at most this much further from the fp64 truth than the fp32 oracle is (max norm and L2) ...
ARB_EPS_MAX = 2e-5    # ... plus this fraction of the tensor's largest entry (max norm)
ARB_EPS_L2 = 1e-5     # ... / of the tensor's norm (L2): where both are at rounding level the ratio means nothing
ARB_COUNT = 1.25      # feature-map texels off by > 1e-4 of the largest entry: at most this many times the oracle's own count ...
ARB_COUNT_SIGMA = 3   # ... + 3 sigma of the count's own scatter: the off texels come in clusters -- a flipped gate of lin_in's output moves
                      # the 4 taps of ONE channel of dG (plain MLP: G's channels are the hidden units), a flipped gate inside the
                      # ResnetBlockFC all 32 channels at the 4 taps -- so a count N is N / c events of c entries, sd = sqrt(c N)


@pytest.mark.parametrize("name", list(REAL_SHAPES))
def test_gradients_fp64_arbiter_at_the_real_training_shapes(hip, name):
    """NO relu-kink mask: every ray keeps its upstream gradient.  Truth = the oracle's formulas in fp64 (torch ops on the GPU);
    fp32 oracle on the CPU = what the reference computes.  Two correct fp32 evaluations gate a few units differently where a
    pre-activation sits within rounding of zero, so neither matches the truth to 1e-4 everywhere -- but the HIP path must not be
    FURTHER from it than the fp32 oracle is."""
    c = _real_case(name)
    s = c["s"]
    truth = _oracle_grads(c["scene"], c["mlp"], c["cfg"], s["ids_render"], c["rays"], c["z"], s["n"], s["hard_cap"], c["full"],
                          dtype=torch.float64, device="cuda")
    ours = _hip_real(hip, c, c["full"])
    rec, fails = {}, []
    for a, b, t, nme in zip(ours, c["ref_full"], truth, c["names"]):
        a = a.detach().cpu().double()
        t, b = t.view_as(a), b.view_as(a).double()
        tmax, tnorm = t.abs().max().item(), t.norm().item()
        e_hip, e_ref = (a - t).abs(), (b - t).abs()
        r = dict(max_hip=e_hip.max().item() / tmax, max_ref=e_ref.max().item() / tmax, l2_hip=e_hip.norm().item() / tnorm,
                 l2_ref=e_ref.norm().item() / tnorm, n_off_hip=int((e_hip > 1e-4 * tmax).sum()), n_off_ref=int((e_ref > 1e-4 * tmax).sum()),
                 numel=a.numel())
        rec[nme] = r
        print(f"{name} {nme}: max/max-entry HIP {r['max_hip']:.2e} oracle {r['max_ref']:.2e}; L2 HIP {r['l2_hip']:.2e} oracle {r['l2_ref']:.2e}; "
              f"entries off by > 1e-4 max: HIP {r['n_off_hip']} oracle {r['n_off_ref']} of {r['numel']}")
        if r["max_hip"] > ARB_FACTOR * r["max_ref"] + ARB_EPS_MAX:
            fails.append((nme, "max", r["max_hip"], r["max_ref"]))
        if r["l2_hip"] > ARB_FACTOR * r["l2_ref"] + ARB_EPS_L2:
            fails.append((nme, "l2", r["l2_hip"], r["l2_ref"]))
        cluster = 4 * (1 + 31 * s["nb"])
        if r["n_off_hip"] > ARB_COUNT * r["n_off_ref"] + 10 + ARB_COUNT_SIGMA * (cluster * max(r["n_off_ref"], 1)) ** 0.5:
            fails.append((nme, "count", r["n_off_hip"], r["n_off_ref"]))
    _record(name, dict(fp64_arbiter=rec))
    assert not fails, fails


def test_multiscale_render_and_backward_vs_oracle(hip):
    """trainer.py:220-242 with prediction_mode "multiscale" (the default exp_re10k.yaml runs with): the encoder returns four feature
    maps, full to 1/8 resolution; BTSNet.encode resizes them to scale 0's size (nearest, models_bts.py:111-119) and every training
    step renders once per scale after net.set_scale(i).  Each scale's outputs and gradients (w.r.t. the MLP and w.r.t. that
    scale's OWN low-resolution feature map, through the resize) against the oracle on the resized map."""
    import torch.nn.functional as F
    import behindthescenes_amd as bts
    from tests._hip_helpers import make_conf, load_mlp
    cfg = O.FieldConfig(d_min=1.0, d_max=100.0, code_mode="distance")
    g = torch.Generator().manual_seed(91)
    n, v, H, W, C, Hd, K = 2, 3, 64, 96, 32, 32, 24
    scene = O.synthetic_scene(n, v, H, W, C, seed=91, intrinsics=O.K_RE10K, baseline=0.2, smooth=True)
    mlp = O.init_mlp(C + 39, Hd, 1, gen=g)
    feats = [F.avg_pool2d(torch.randn(n, C, H >> s, W >> s, generator=g), 3, 1, 1) * 2 for s in range(4)]
    conf = make_conf(cfg, C, Hd, 1, H, W)
    conf["encoder"].update(n_scales=4, pyramid=True, num_views=n)
    net = bts.BTSNet(conf)
    load_mlp(net, mlp)
    assert [tuple(f.shape[-2:]) for f in net.encoder.feats] == [(H >> s, W >> s) for s in range(4)] and list(net.encoder.scales) == [0, 1, 2, 3]
    with torch.no_grad():
        for f_dst, f_src in zip(net.encoder.feats, feats):
            f_dst.copy_(f_src)
    net = net.cuda().train()
    renderer = hip.NeRFRenderer(n_coarse=K, lindisp=True, hard_alpha_cap=False).cuda()
    net.encode(scene["images"].cuda(), scene["projs"].cuda(), scene["poses"].cuda(), ids_encoder=[0], ids_render=[1, 2])
    for s_i in range(4):
        up = F.interpolate(feats[s_i], (H, W))                     # what the reference renders from at this scale
        sc = dict(scene, feat=up)
        rays, z, mask = _patch_rays_with_kink_mask(sc, mlp, cfg, [0], [1, 2], 8, K, g, margin=2e-5)
        c_rgb = torch.randn(n * 512, 6, generator=g) * mask.unsqueeze(-1)

        def loss_fn(w, rgb, depth, a):
            return (rgb * c_rgb.to(rgb.device, rgb.dtype)).sum() + 0.05 * (depth * mask.to(depth.device, depth.dtype)).sum()

        # oracle: gradients w.r.t. the MLP and w.r.t. the LOW-RESOLUTION map through the resize
        params = [t.clone().requires_grad_(True) for t in mlp.tensors()]
        lo = feats[s_i].clone().requires_grad_(True)
        m = O.MlpParams(params[0], params[1], [tuple(params[2:6])], params[-2], params[-1])
        st = O.make_state(dict(scene, feat=F.interpolate(lo, (H, W))), [1, 2], cfg)
        ow, orgb, odepth, oa, *_ = O.composite(rays.reshape(-1, 8), z, n, st, m, cfg, hard_alpha_cap=False)
        ref = torch.autograd.grad(loss_fn(ow, orgb, odepth, oa), params + [lo])
        # HIP path at this scale
        net.set_scale(s_i)
        assert net.get_scale() == s_i
        net.zero_grad(set_to_none=True)
        w, rgb, depth, a, inv, _, _ = renderer.composite(net, rays.reshape(-1, 8).cuda(), z.cuda(), sb=n)
        torch.testing.assert_close(depth.detach().cpu(), odepth.detach(), rtol=DEPTH_RTOL_MS, atol=1e-5)
        torch.testing.assert_close(rgb.detach().cpu(), orgb.detach(), rtol=0, atol=1e-5)
        loss_fn(w, rgb, depth, a).backward()
        mc = net.mlp_coarse
        ours = [mc.lin_in.weight.grad, mc.lin_in.bias.grad, mc.blocks[0].fc_0.weight.grad, mc.blocks[0].fc_0.bias.grad,
                mc.blocks[0].fc_1.weight.grad, mc.blocks[0].fc_1.bias.grad, mc.lin_out.weight.grad, mc.lin_out.bias.grad,
                net.encoder.feats[s_i].grad]
        for a_, b_, nme in zip(ours, ref, ["w_in", "b_in", "fc_0.w", "fc_0.b", "fc_1.w", "fc_1.b", "w_out", "b_out", f"feat[{s_i}]"]):
            err = _rel_to_max(a_, b_.view_as(a_.cpu()))
            assert err <= GRAD_RTOL, (s_i, nme, err)
        for j in range(4):   # the other scales' maps received nothing from this render
            if j != s_i:
                assert net.encoder.feats[j].grad is None or float(net.encoder.feats[j].grad.abs().max()) == 0.0


def test_projection_kernels_vs_torch(hip):
    """bts_project_features / _bwd are plain GEMMs: check against torch matmul (fp64 accumulate) incl. ragged pixel counts."""
    from behindthescenes_amd import native
    g = torch.Generator().manual_seed(1)
    for (C, Hd, N, H, W) in ((64, 64, 2, 24, 80), (32, 32, 3, 7, 13), (64, 64, 1, 192, 640), (64, 64, 3, 5, 27), (32, 32, 2, 64, 96)):
        spec = native.FieldSpec(C=C, d_hidden=Hd, n_blocks=0)
        F_ = torch.randn(N, C, H, W, generator=g).cuda()
        mlp = torch.randn(spec.mlp_param_count(), generator=g).cuda() * 0.2
        w = mlp[:Hd * spec.d_in].view(Hd, spec.d_in)[:, :C]
        order = native.proj_storage_order(Hd).cuda()      # channel s of the stored map holds hidden unit order[s]
        G = native.project_features(spec, F_, mlp)
        G_ref = torch.einsum("nchw,jc->nhwj", F_.double(), w.double())[..., order]
        assert (G.double() - G_ref).abs().max().item() <= 2e-5 * G_ref.abs().max().item()
        dG = torch.randn(N, H, W, Hd, generator=g).cuda()   # gradient w.r.t. the stored map
        dF, dM = native.project_features_bwd(spec, F_, dG, mlp)
        dG_nat = torch.empty_like(dG)
        dG_nat[..., order] = dG
        dF_ref = torch.einsum("nhwj,jc->nchw", dG_nat.double(), w.double())
        dW_ref = torch.einsum("nhwj,nchw->jc", dG_nat.double(), F_.double())
        assert (dF.double() - dF_ref).abs().max().item() <= 2e-5 * dF_ref.abs().max().item()
        dW = dM[:Hd * spec.d_in].view(Hd, spec.d_in)
        assert (dW[:, :C].double() - dW_ref).abs().max().item() <= 1e-4 * dW_ref.abs().max().item()
        assert float(dW[:, C:].abs().max()) == 0.0 and float(dM[Hd * spec.d_in:].abs().max()) == 0.0
        # one gradient only: all four waves of a work-group take the one role (the fused kernel's other launch shapes)
        dF1, none = native.project_features_bwd(spec, F_, dG, mlp, need_feat=True, need_mlp=False)
        assert none is None and torch.equal(dF1, dF)
        none, dM1 = native.project_features_bwd(spec, F_, dG, mlp, need_feat=False, need_mlp=True)
        assert none is None and (dM1 - dM).abs().max().item() <= 1e-5 * dM.abs().max().item()       # atomics: summation order only


def test_direct_feature_path_matches_projected_path(hip):
    """The per-point lin_in variant (raw channels-last F, PROJ=false) and the default projected-G variant are the same function up
    to summation order: compare them on a golden case, and the direct one against the reference golden too."""
    from behindthescenes_amd import native
    from tests._hip_helpers import net_from_case
    for name in ("kitti_train", "re10k_train"):
        c = Case(name)
        net = net_from_case(c)
        ft = net.native_field()
        mlp = net.mlp_coarse.packed().detach()
        feat_nhwc = native.nchw_to_nhwc(net.grid_f_features[0][:, 0].detach().contiguous())
        ft_direct = native.FieldTensors(net.spec, None, ft.K_enc, ft.w2c_enc, ft.imgs_nhwc4, ft.K_r, ft.w2c_r, None, feat_nhwc=feat_nhwc)
        rays, z = c.rays.reshape(-1, 8).cuda(), c.z_samp.cuda()
        a = native.render_fwd(ft, mlp, rays, z, hard_alpha_cap=c.hard_cap, want_weights=True)
        b = native.render_fwd(ft_direct, mlp, rays, z, hard_alpha_cap=c.hard_cap, want_weights=True)
        assert torch.equal(a["invalid"], b["invalid"])
        torch.testing.assert_close(a["depth"], b["depth"], rtol=1e-5, atol=0)
        torch.testing.assert_close(a["rgb"], b["rgb"], rtol=0, atol=2e-6)
        torch.testing.assert_close(b["depth"].cpu(), c.t["out_depth"], rtol=1e-4, atol=0)
        torch.testing.assert_close(b["weights"].cpu(), c.t["out_weights"], rtol=0, atol=1e-5)
