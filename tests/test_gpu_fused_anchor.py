"""GPU: the two paths bench.py times -- ``FusedEvalFrame`` (bts_eval_frame) and ``FusedTrainStep`` (bts_train_step_fwd / _bwd) -- anchored
DIRECTLY on the oracle and on the real reference's golden step (round-5 verdict, missing 2): until this file their only anchor was
transitive (fused == entry-by-entry at toy shapes, entry-by-entry == reference golden).

* the evaluator's frame (models/bts/evaluator.py:60-79) at BASELINE.json configs[1]'s full size against O.composite + O.distance_to_z:
  depth within 1e-4 relative (max-norm), colours / weights at the bars of tests/test_gpu_parity.py, Abs-Rel difference <= 1e-4;
* the trainer's step (models/bts/trainer.py:208-259 + models/bts/model/loss.py:83-293) on tests/golden/train_step.npz (generated from
  the imported reference, tests/golden/gen_golden_loss.py): loss within 1e-5, every gradient within 1e-4 of its largest entry;
* the same step at exp_kitti_raw.yaml's per-sample shape (192x640, 2048 rays / sample, K = 64, nv = 2) against the oracle's composite
  + loss + autograd, and exp_re10k.yaml's (256x384, one ResnetBlockFC, distance code, K = 48) on one scale.

The steps' own random draws (patches, jitter) are replaced through the deterministic sub-seam (`patches=`, `jitter=`)."""
import ast

import numpy as np
import pytest
import torch

from oracle import bts_loss as OL
from oracle import bts_oracle as O
from tests._cases import GOLDEN, robust_ray_mask

pytestmark = pytest.mark.gpu

DEPTH_RTOL, ABS_TOL, NOISE_FLOOR = 1e-4, 1e-5, 5e-5


def _net(cfg, mlp, scene, H, W, C, train, learn_empty_feature=None):
    import behindthescenes_amd as bts
    from tests._hip_helpers import load_mlp, make_conf
    net = bts.BTSNet(make_conf(cfg, C, mlp.w_in.shape[0], len(mlp.blocks), H, W))
    load_mlp(net, mlp)
    with torch.no_grad():
        net.encoder.feats[0].data = scene["feat"].clone()
        if learn_empty_feature is not None:
            net.empty_feature.copy_(learn_empty_feature)
    net = net.cuda()
    return net.train(train)


@pytest.mark.parametrize("learn_empty", [False, True], ids=["plain", "learn_empty"])
def test_fused_eval_frame_vs_oracle_full_size(learn_empty):
    """BASELINE.json configs[1]: one 192x640 stereo pair, both frames' rays rendered from the encoder view, K = 64, hard alpha cap
    (245 760 rays, 15.7 M field queries) through ONE bts_eval_frame call -- exactly what bench.py's headline times.  learn_empty is
    eval_depth.yaml's effective setting (BTSNet's default, models_bts.py:24)."""
    import behindthescenes_amd as bts
    n, v, H, W, C, K = 1, 2, 192, 640, 64, 64
    cfg = O.FieldConfig(learn_empty=learn_empty)
    g = torch.Generator().manual_seed(61)
    scene = O.synthetic_scene(n, v, H, W, C, seed=61, intrinsics=O.K_KITTIRAW, smooth=True)
    mlp = O.init_mlp(C + 39, 64, 0, gen=g)
    empty = torch.randn(C, generator=g) if learn_empty else None
    u = torch.rand(n * v * H * W, K, generator=g)
    # ---- the oracle: ImageRaySampler.sample -> sample_coarse -> composite -> reconstruct -> distance_to_z
    rays = O.image_rays(scene["poses"], scene["projs"], H, W, cfg.d_min, cfg.d_max)
    z = O.sample_coarse(rays.reshape(-1, 8), K, True, u)
    st = O.make_state(scene, [0], cfg, empty)
    with torch.no_grad():
        ow, orgb, odepth, oa, oinv, _, _ = O.composite(rays.reshape(-1, 8), z, n, st, mlp, cfg, hard_alpha_cap=True)
    oz = O.distance_to_z(odepth.view(n, v, H, W), scene["projs"])
    # ---- the product: one library call
    net = _net(cfg, mlp, scene, H, W, C, train=False, learn_empty_feature=empty)
    wrapped = bts.NeRFRenderer.from_conf(dict(n_coarse=K, lindisp=True, hard_alpha_cap=True)).bind_parallel(net).eval().cuda()
    frame = bts.FusedEvalFrame(wrapped, bts.ImageRaySampler(cfg.d_min, cfg.d_max))
    data = frame(scene["images"].cuda(), scene["projs"].cuda(), scene["poses"].cuda(), ids_encoder=[0], ids_render=[0], jitter=u.cuda())
    assert frame.last_path == "fused", frame.last_path
    c = data["coarse"][0]
    assert c["depth"].shape == (n, v, H, W) and c["rgb"].shape == (n, v, H, W, 1, 3) and c["weights"].shape == (n, v, H, W, K)
    assert (data["rays"].cpu() - rays).abs().max().item() <= 2e-6
    torch.testing.assert_close(data["rgb_gt"].cpu(), (scene["images"] * .5 + .5).permute(0, 1, 3, 4, 2), rtol=0, atol=0)
    # ---- rays on which an `invalid` flag flipped (a pixel projecting exactly onto a frustum border: SURVEY 7 hazard iv) are set aside,
    # but every one of them must sit within 3e-6 of a border test
    inv = c["invalid"].cpu().reshape(-1, K, 1) > 0.5
    flips = (inv != oinv).any(-1).any(-1)
    border = ~robust_ray_mask(st, rays, z, margin=3e-6)
    assert not (flips & ~border).any(), "flag differs on a ray that is not within 3e-6 of any frustum border"
    if learn_empty:       # a hidden flip of the encoder flag swaps the feature vector: those rays sit near the border as well
        flips = flips | ~robust_ray_mask(st, rays, z, margin=1e-5)
    ok = ~flips
    print(f"rays set aside: {int(flips.sum())} of {flips.numel()}")
    assert flips.float().mean().item() < 0.05
    dz, odz = c["depth"].cpu().reshape(-1), oz.reshape(-1)
    rel = ((dz - odz).abs() / odz.abs())[ok]
    assert rel.max().item() <= DEPTH_RTOL, rel.max().item()            # north_star: depth maps within 1e-4 rel -- max-norm
    for key, got, ref in (("rgb", c["rgb"].cpu().reshape(-1, 3), orgb), ("weights", c["weights"].cpu().reshape(-1, K), ow),
                          ("alphas", c["alphas"].cpu().reshape(-1, K), oa)):
        e = (got - ref).abs()[ok].flatten()
        big = int((e > ABS_TOL).sum())
        print(f"  {key}: max |err| {e.max().item():.2e}, entries above 1e-5: {big} of {e.numel()}")
        assert big <= (2e-4 if key == "alphas" else 1e-4) * e.numel(), (key, big)
        assert e.max().item() <= (3 * NOISE_FLOOR if key == "alphas" else NOISE_FLOOR), (key, e.max().item())
    # ---- Abs-Rel (evaluator.py:96-151) against synthetic sparse ground truth: SURVEY 8d
    gt = torch.rand(1, 1, H, W, generator=g) * 77 + 3
    gt = gt * (torch.rand(1, 1, H, W, generator=g) < 0.05) * (torch.arange(H).view(1, 1, -1, 1) >= int(0.4 * H))
    ours, theirs = O.abs_rel(c["depth"][:, :1].cpu(), gt), O.abs_rel(oz[:, :1], gt)
    print(f"Abs-Rel: fused frame {ours:.6f}, oracle {theirs:.6f}")
    assert abs(ours - theirs) <= 1e-4


def _fused_step(net, K, rays_per_sample, cfg, hard_cap, policy="weight_guided"):
    import behindthescenes_amd as bts
    from behindthescenes_amd.train_step import FusedTrainStep
    renderer = bts.NeRFRenderer.from_conf(dict(n_coarse=K, lindisp=True, hard_alpha_cap=hard_cap, lean_training_outputs=True)).cuda().train()
    sampler = bts.PatchRaySampler(ray_batch_size=rays_per_sample, z_near=cfg.d_min, z_far=cfg.d_max, patch_size=8)
    crit = bts.ReconstructionLoss({"criterion": "l1+ssim", "invalid_policy": policy, "lambda_edge_aware_smoothness": 0.001})
    return FusedTrainStep(renderer.bind_parallel(net).train(), sampler, crit)


def test_fused_train_step_vs_reference_golden():
    """tests/golden/train_step.npz: the REAL reference's PatchRaySampler.sample (seed 701) -> composite (jitter `u`) -> reconstruct ->
    ReconstructionLoss -> backward.  FusedTrainStep draws the same patches from the same CPU generator state (the reference's order of
    draws) and takes `u` through the jitter seam: rays and patch colours must come out as the reference's, the loss within 1e-5, the
    gradients of lin_in / lin_out / the feature map within 1e-4 of their largest entry."""
    z = np.load(f"{GOLDEN}/train_step.npz")
    meta = ast.literal_eval(str(z["meta"]))
    t = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}
    n, pc, ps, K, H, W = meta["n"], meta["patches"], meta["patch"], meta["K"], meta["H"], meta["W"]
    cfg = O.FieldConfig(d_min=meta["d_min"], d_max=meta["d_max"])
    scene = dict(images=t["images"], feat=t["feat"], projs=t["projs"], poses=t["poses"])
    net = _net(cfg, O.MlpParams(t["w_in"], t["b_in"], [], t["w_out"], t["b_out"]), scene, H, W, meta["C"], train=True)
    step = _fused_step(net, K, pc * ps * ps, cfg, hard_cap=True)
    torch.manual_seed(meta["seed"] + 1)                       # gen_golden_loss.py:41 -- the state the reference's sampler drew from
    loss, parts, data = step(t["images"].cuda(), t["projs"].cuda(), t["poses"].cuda(), ids_encoder=[0], ids_render=meta["ids_render"],
                             ids_loss=meta["ids_loss"], jitter=t["u"].cuda())
    assert step.last_path == "fused", step.last_path
    # the step's own sampling reproduced the reference's: same patches, same rays, same ground-truth colours
    assert (data["rays"].cpu() - t["rays"]).abs().max().item() <= 2e-6
    torch.testing.assert_close(data["rgb_gt"].cpu().reshape(t["rgb_gt"].shape), t["rgb_gt"], rtol=0, atol=1e-6)
    c = data["coarse"][0]
    torch.testing.assert_close(c["depth"].detach().cpu(), t["out_depth"], rtol=1e-4, atol=0)
    torch.testing.assert_close(c["rgb"].detach().cpu(), t["out_rgb"], rtol=0, atol=1e-5)
    assert abs(loss.item() - t["loss"].item()) <= 1e-5, (loss.item(), t["loss"].item())
    assert abs(parts["loss"] - t["loss"].item()) <= 1e-5 and abs(parts["loss_invalid_ratio"] - t["loss_invalid_ratio"].item()) <= 1e-6
    assert abs(parts["loss_eas"] - t["loss_eas"].item()) <= 1e-6 and abs(parts["loss_rgb_coarse"] - t["loss_rgb_coarse"].item()) <= 1e-6
    loss.backward()
    got = dict(g_w_in=net.mlp_coarse.lin_in.weight.grad, g_b_in=net.mlp_coarse.lin_in.bias.grad,
               g_w_out=net.mlp_coarse.lin_out.weight.grad, g_b_out=net.mlp_coarse.lin_out.bias.grad, g_feat=net.encoder.feats[0].grad)
    for k, gr in got.items():
        ref = t[k].view_as(gr.cpu())
        err = (gr.cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-20)
        assert err <= 1e-4, (k, err)


STEP_SHAPES = {
    # exp_kitti_raw.yaml per sample: 192x640, 2048 rays (32 patches), K = 64, two loss + two render frames, hard alpha cap
    "kitti_raw": dict(n=2, v=4, H=192, W=640, C=64, Hd=64, nb=0, K=64, rays=2048, ids_loss=[0, 1], ids_render=[2, 3], hard_cap=True,
                      cfg=dict(d_min=3.0, d_max=80.0), intr="K_KITTIRAW"),
    # exp_re10k.yaml per sample (one scale): 256x384, 1024 rays, K = 48, one ResnetBlockFC of width 32, distance code, no alpha cap
    "re10k": dict(n=2, v=3, H=256, W=384, C=32, Hd=32, nb=1, K=48, rays=1024, ids_loss=[0], ids_render=[1, 2], hard_cap=False,
                  cfg=dict(d_min=1.0, d_max=100.0, code_mode="distance"), intr="K_RE10K"),
}


@pytest.mark.parametrize("shape", list(STEP_SHAPES))
def test_fused_train_step_vs_oracle_at_the_yaml_shapes(shape):
    """The two library calls at the configs' real per-sample shapes against the oracle's restatement of the same step -- composite
    (nerf.py:210-313) on the step's rays with the same jitter, the loss of loss.py:83-293 (oracle/bts_loss.py), torch autograd for the
    gradients.  Loss within 1e-5; gradients within 1e-4 of the largest entry (the feature-map gradient is compared where the oracle's
    own `invalid` flags agree with the kernel's: a flipped flag re-routes a sample's whole contribution)."""
    s = STEP_SHAPES[shape]
    n, v, H, W, C, K = s["n"], s["v"], s["H"], s["W"], s["C"], s["K"]
    cfg = O.FieldConfig(**s["cfg"])
    g = torch.Generator().manual_seed(77)
    scene = O.synthetic_scene(n, v, H, W, C, seed=77, intrinsics=getattr(O, s["intr"]), smooth=True, baseline=0.4)
    mlp = O.init_mlp(C + 39, s["Hd"], s["nb"], gen=g)
    u = torch.rand(n * s["rays"], K, generator=g)
    net = _net(cfg, mlp, scene, H, W, C, train=True)
    step = _fused_step(net, K, s["rays"], cfg, s["hard_cap"])
    torch.manual_seed(5)
    loss, parts, data = step(scene["images"].cuda(), scene["projs"].cuda(), scene["poses"].cuda(), ids_encoder=[0], ids_render=s["ids_render"],
                             ids_loss=s["ids_loss"], jitter=u.cuda())
    assert step.last_path == "fused", step.last_path
    loss.backward()
    # ---- the oracle on the step's own rays / patch colours (the sampler is pinned to the reference's by the golden test above and by
    # tests/test_gpu_protocol.py)
    rays, rgb_gt = data["rays"].cpu(), data["rgb_gt"].cpu()
    P = s["rays"] // 64
    feat = scene["feat"].clone().requires_grad_(True)
    leaves = [mlp.w_in, mlp.b_in, mlp.w_out, mlp.b_out] + [x for b in mlp.blocks for x in b]
    for x in leaves:
        x.requires_grad_(True)
    st = O.make_state(dict(scene, feat=feat), s["ids_render"], cfg)
    zs = O.sample_coarse(rays.reshape(-1, 8), K, True, u)
    ow, orgb, odepth, oa, oinv, _, _ = O.composite(rays.reshape(-1, 8), zs, n, st, mlp, cfg, hard_alpha_cap=s["hard_cap"])
    nv = len(s["ids_render"])
    coarse = dict(rgb=orgb.view(n, P, 8, 8, nv, 3), depth=odepth.view(n, P, 8, 8), weights=ow.view(n, P, 8, 8, K),
                  invalid=oinv.view(n, P, 8, 8, K, nv), alphas=oa.view(n, P, 8, 8, K))
    oloss, oparts = OL.reconstruction_loss(coarse, rgb_gt.view(n, P, 8, 8, 3))
    oloss.backward()
    c = data["coarse"][0]
    rel = ((c["depth"].detach().cpu().reshape(-1) - odepth.detach()).abs() / odepth.detach().abs())
    robust = robust_ray_mask(O.make_state(scene, s["ids_render"], cfg), rays, zs, margin=1e-5)
    assert rel[robust].max().item() <= DEPTH_RTOL, rel[robust].max().item()
    assert robust.float().mean().item() > 0.9
    assert abs(loss.item() - oloss.item()) <= 1e-5, (loss.item(), oloss.item())
    assert abs(parts["loss_invalid_ratio"] - float(oparts["loss_invalid_ratio"])) <= 1e-4
    m = net.mlp_coarse
    pairs = [("lin_in.weight", m.lin_in.weight.grad, mlp.w_in.grad), ("lin_in.bias", m.lin_in.bias.grad, mlp.b_in.grad),
             ("lin_out.weight", m.lin_out.weight.grad, mlp.w_out.grad), ("lin_out.bias", m.lin_out.bias.grad, mlp.b_out.grad),
             ("feat", net.encoder.feats[0].grad, feat.grad)]
    for i, (blk, ob) in enumerate(zip(m.blocks, mlp.blocks)):
        pairs += [(f"blk{i}.fc_0.weight", blk.fc_0.weight.grad, ob[0].grad), (f"blk{i}.fc_0.bias", blk.fc_0.bias.grad, ob[1].grad),
                  (f"blk{i}.fc_1.weight", blk.fc_1.weight.grad, ob[2].grad), (f"blk{i}.fc_1.bias", blk.fc_1.bias.grad, ob[3].grad)]
    worst = {}
    for k, got, ref in pairs:
        assert got is not None and ref is not None, k
        ref = ref.view_as(got.cpu())
        worst[k] = (got.cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-20)
    print({k: f"{e:.1e}" for k, e in worst.items()})
    for k, e in worst.items():
        assert e <= 1e-4, (k, e, worst)
