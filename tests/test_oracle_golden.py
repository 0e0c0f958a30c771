"""CPU: pins the oracle (oracle/bts_oracle.py) against fixtures produced by the REAL reference
(tests/golden/gen_golden.py) and, when the reference tree is present, against the live reference."""
import numpy as np
import pytest
import torch

from oracle import bts_oracle as O
from tests._cases import Case, RENDER_CASES, GRAD_CASES, GOLDEN


@pytest.mark.parametrize("name", RENDER_CASES)
def test_composite_matches_reference_golden(name):
    c = Case(name)
    with torch.no_grad():
        w, rgb, depth, a, inv, _, rs = O.composite(c.rays.reshape(-1, 8), c.z_samp, c.rays.shape[0], c.state, c.mlp, c.cfg,
                                                  hard_alpha_cap=c.hard_cap)
    t = c.t
    # same torch ops as the reference -> essentially bit-level agreement
    assert torch.equal(inv, t["out_invalid"])
    torch.testing.assert_close(depth, t["out_depth"], rtol=2e-6, atol=0)
    torch.testing.assert_close(rgb, t["out_rgb"], rtol=0, atol=2e-6)
    torch.testing.assert_close(w, t["out_weights"], rtol=0, atol=2e-6)
    torch.testing.assert_close(a, t["out_alphas"], rtol=0, atol=2e-6)
    torch.testing.assert_close(rs, t["out_rgb_samps"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("name", RENDER_CASES)
def test_field_query_matches_reference_golden(name):
    c = Case(name)
    with torch.no_grad():
        rgb, inv, sig = O.field_forward(c.t["q_pts"], c.state, c.mlp, c.cfg)
        torch.testing.assert_close(rgb, c.t["q_rgb"], rtol=0, atol=1e-6)
        assert torch.equal(inv, c.t["q_invalid"])
        torch.testing.assert_close(sig, c.t["q_sigma"], rtol=2e-6, atol=1e-7)
        if "q_sigma_density" in c.t:
            _, inv_d, sig_d = O.field_forward(c.t["q_pts"], c.state, c.mlp, c.cfg, only_density=True)
            torch.testing.assert_close(sig_d, c.t["q_sigma_density"], rtol=2e-6, atol=1e-7)
            # the reference returns invalid_features un-reduced for only_density: (n, nv_enc=1, P, 1)  (models_bts.py:337)
            assert torch.equal(inv_d, c.t["q_invalid_density"][:, 0])


@pytest.mark.parametrize("name", GRAD_CASES)
def test_gradients_match_reference_golden(name):
    c = Case(name)
    params = [p.clone().requires_grad_(True) for p in c.mlp.tensors()]
    feat = c.state.feat.clone().requires_grad_(True)
    nb = c.meta["nb"]
    blocks = [tuple(params[2 + 4 * i: 6 + 4 * i]) for i in range(nb)]
    mlp = O.MlpParams(params[0], params[1], blocks, params[-2], params[-1])
    st = O.FieldState(feat, c.state.K_enc, c.state.w2c_enc, c.state.imgs, c.state.K_r, c.state.w2c_r)
    w, rgb, depth, *_ = O.composite(c.rays.reshape(-1, 8), c.z_samp, c.rays.shape[0], st, mlp, c.cfg, hard_alpha_cap=c.hard_cap)
    loss = (rgb * c.t["gin_rgb"]).sum() + (depth * c.t["gin_depth"]).sum()
    grads = torch.autograd.grad(loss, params + [feat])
    names = ["g_w_in", "g_b_in"] + sum([[f"g_blk{i}_w0", f"g_blk{i}_b0", f"g_blk{i}_w1", f"g_blk{i}_b1"] for i in range(nb)], []) \
        + ["g_w_out", "g_b_out", "g_feat"]
    for g, nme in zip(grads, names):
        ref = c.t[nme]
        scale = ref.abs().max().item() + 1e-12
        assert (g - ref).abs().max().item() <= 2e-5 * scale, nme


def test_sample_coarse_matches_reference_golden():
    for name in RENDER_CASES:
        c = Case(name)
        z = O.sample_coarse(c.rays.reshape(-1, 8), c.meta["K"], True, c.t["u"])
        assert torch.equal(z, c.z_samp), name


def test_gen_rays_and_distance_to_z_golden():
    z = np.load(f"{GOLDEN}/misc.npz")
    poses, projs = torch.from_numpy(z["poses"]), torch.from_numpy(z["projs"])
    focal, center = projs[:, [0, 1], [0, 1]], projs[:, [0, 1], [2, 2]]
    for key, nd in (("rays_norm", True), ("rays_unnorm", False)):
        r = O.gen_rays(poses, 20, 12, 3.0, 80.0, focal, center, norm_dir=nd)
        torch.testing.assert_close(r, torch.from_numpy(z[key]), rtol=0, atol=1e-7)
    dz = O.distance_to_z(torch.from_numpy(z["depths"]), torch.from_numpy(z["projs2"]))
    torch.testing.assert_close(dz, torch.from_numpy(z["dist_to_z"]), rtol=1e-6, atol=0)


@pytest.mark.needs_reference
def test_oracle_against_live_reference_fullres_slice():
    """Build-container only: a fresh (not committed) comparison against the live reference at the real
    192x640 resolution on a slice of rays, so that the oracle is not just fitted to the small fixtures."""
    from oracle.ref_shim import load_reference
    from tests.golden.gen_golden import ref_conf, load_mlp_into
    ref = load_reference()
    cfg = O.FieldConfig()
    scene = O.synthetic_scene(1, 2, 192, 640, 64, seed=9, intrinsics=O.K_KITTIRAW)
    g = torch.Generator().manual_seed(5)
    mlp = O.init_mlp(103, 64, 0, gen=g)
    net = ref.make_net(ref_conf(cfg, 0, 64), [scene["feat"]])
    load_mlp_into(net, mlp)
    renderer = ref.NeRFRenderer(n_coarse=64, lindisp=True, hard_alpha_cap=True).eval()
    net.eval()
    net.encode(scene["images"], scene["projs"], scene["poses"], ids_encoder=[0], ids_render=[0])
    rays = O.image_rays(scene["poses"], scene["projs"], 192, 640, 3.0, 80.0)[:, ::97].contiguous()
    u = torch.rand(rays.shape[1], 64, generator=g)
    z = O.sample_coarse(rays.reshape(-1, 8), 64, True, u)
    with torch.no_grad():
        ref_out = renderer.composite(net, rays.reshape(-1, 8), z, coarse=True, sb=1)
        st = O.make_state(scene, [0], cfg)
        our = O.composite(rays.reshape(-1, 8), z, 1, st, mlp, cfg, hard_alpha_cap=True)
    torch.testing.assert_close(our[2], ref_out[2], rtol=2e-6, atol=0)
    torch.testing.assert_close(our[1], ref_out[1], rtol=0, atol=2e-6)
    assert torch.equal(our[4], ref_out[4])


def _train_step_case():
    import ast
    z = np.load(f"{GOLDEN}/train_step.npz")
    meta = ast.literal_eval(str(z["meta"]))
    t = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}
    return meta, t


def test_loss_matches_reference_golden():
    """oracle/bts_loss.py on the render dict the reference's own ReconstructionLoss consumed -> the reference's loss value"""
    from oracle import bts_loss as OL
    meta, t = _train_step_case()
    coarse = dict(rgb=t["out_rgb"], depth=t["out_depth"], weights=t["out_weights"], alphas=t["out_alphas"], invalid=t["out_invalid"])
    n, pc, ps = meta["n"], meta["patches"], meta["patch"]
    loss, parts = OL.reconstruction_loss(coarse, t["rgb_gt"].view(n, pc, ps, ps, 3))
    assert abs(loss.item() - t["loss"].item()) <= 1e-6
    assert abs(parts["loss_eas"].item() - t["loss_eas"].item()) <= 1e-6
    assert abs(parts["loss_rgb_coarse"].item() - t["loss_rgb_coarse"].item()) <= 1e-6
    assert abs(parts["loss_invalid_ratio"].item() - t["loss_invalid_ratio"].item()) <= 1e-7


def test_train_step_oracle_end_to_end_golden():
    """oracle renderer + oracle loss + autograd == the reference's training-step loss and gradients"""
    from oracle import bts_loss as OL
    meta, t = _train_step_case()
    n, pc, ps, K = meta["n"], meta["patches"], meta["patch"], meta["K"]
    cfg = O.FieldConfig(d_min=meta["d_min"], d_max=meta["d_max"])
    params = [t[k].clone().requires_grad_(True) for k in ("w_in", "b_in", "w_out", "b_out")]
    feat = t["feat"].clone().requires_grad_(True)
    mlp = O.MlpParams(params[0], params[1], [], params[2], params[3])
    scene = dict(images=t["images"], feat=feat, projs=t["projs"], poses=t["poses"])
    st = O.make_state(scene, meta["ids_render"], cfg)
    st = O.FieldState(feat, st.K_enc, st.w2c_enc, st.imgs, st.K_r, st.w2c_r)
    w, rgb, depth, a, inv, _, _ = O.composite(t["rays"].reshape(-1, 8), t["z_samp"], n, st, mlp, cfg, hard_alpha_cap=True)
    nv = len(meta["ids_render"])
    coarse = dict(rgb=rgb.view(n, pc, ps, ps, nv, 3), depth=depth.view(n, pc, ps, ps), weights=w.view(n, pc, ps, ps, K),
                  alphas=a.view(n, pc, ps, ps, K), invalid=inv.view(n, pc, ps, ps, K, nv))
    loss, _ = OL.reconstruction_loss(coarse, t["rgb_gt"].view(n, pc, ps, ps, 3))
    assert abs(loss.item() - t["loss"].item()) <= 1e-6
    grads = torch.autograd.grad(loss, params + [feat])
    for g, nme in zip(grads, ("g_w_in", "g_b_in", "g_w_out", "g_b_out", "g_feat")):
        ref = t[nme]
        assert (g - ref).abs().max().item() <= 2e-5 * (ref.abs().max().item() + 1e-12), nme


def test_occupancy_profile_matches_the_reference_golden():
    """oracle.profile_points / occupancy_profile against the reference's own get_pts / render_profile
    (scripts/inference_setup.py:84-97, 201-229) -- same torch ops, so the grid is bit-identical and the profile equal wherever no
    running sum sits on the threshold."""
    from tests._cases import ProfileCase
    c = ProfileCase()
    m = c.meta
    q = O.profile_points(tuple(m["x_range"]), tuple(m["y_range"]), tuple(m["z_range"]), m["x_res"], m["y_res"], m["z_res"])
    assert torch.equal(q, c.t["q_pts"])
    with torch.no_grad():
        prof, sigma, invalid = O.occupancy_profile(q, c.state, c.mlp, c.cfg, threshold=m["threshold"], batch_size=50000)
    assert torch.equal(invalid, c.t["invalid"])
    torch.testing.assert_close(sigma, c.t["sigma"], rtol=2e-6, atol=1e-7)
    ok = c.decided_columns()
    assert ok.float().mean() > 0.95
    assert torch.equal(prof[ok], c.t["profile"][ok])
    assert (prof - c.t["profile"]).abs().max().item() <= 1.0 / m["y_res"] + 1e-6     # the others: one level at most


def test_density_noise_matches_the_reference_golden():
    """nerf.py:279-280: the reference renderer in train() mode with noise_std = 0.7 (tests/golden/gen_golden_noise.py: the generator is
    seeded, so the tensor composite() drew is known).  oracle.composite(sigma_noise=...) reproduces outputs and autograd gradients; 17 %
    of the samples of this case sit on relu's zero side because of the noise."""
    c = Case("noise")
    t = c.t
    params = [p.clone().requires_grad_(True) for p in c.mlp.tensors()]
    feat = c.state.feat.clone().requires_grad_(True)
    mlp = O.MlpParams(params[0], params[1], [], params[-2], params[-1])
    st = O.FieldState(feat, c.state.K_enc, c.state.w2c_enc, c.state.imgs, c.state.K_r, c.state.w2c_r)
    w, rgb, depth, a, inv, *_ = O.composite(c.rays.reshape(-1, 8), c.z_samp, c.rays.shape[0], st, mlp, c.cfg, hard_alpha_cap=c.hard_cap,
                                            sigma_noise=t["sigma_noise"])
    assert torch.equal(inv, t["out_invalid"])
    torch.testing.assert_close(depth, t["out_depth"], rtol=2e-6, atol=0)
    torch.testing.assert_close(rgb, t["out_rgb"], rtol=0, atol=2e-6)
    torch.testing.assert_close(w, t["out_weights"], rtol=0, atol=2e-6)
    torch.testing.assert_close(a, t["out_alphas"], rtol=0, atol=2e-6)
    # without the noise the same call is a different render: the fixture does exercise the branch
    with torch.no_grad():
        w0 = O.composite(c.rays.reshape(-1, 8), c.z_samp, c.rays.shape[0], c.state, c.mlp, c.cfg, hard_alpha_cap=c.hard_cap)[0]
    assert (w0 - t["out_weights"]).abs().max().item() > 1e-2
    grads = torch.autograd.grad((rgb * t["gin_rgb"]).sum() + (depth * t["gin_depth"]).sum(), params + [feat])
    for g, nme in zip(grads, ["g_w_in", "g_b_in", "g_w_out", "g_b_out", "g_feat"]):
        ref = t[nme]
        assert (g - ref.view_as(g)).abs().max().item() <= 2e-5 * (ref.abs().max().item() + 1e-12), nme
