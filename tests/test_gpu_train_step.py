"""GPU: one training-step data flow after the CNN through the drop-in modules (BTSNet.encode -> NeRFRenderer.composite on the
reference's seeded patch rays -> _format_outputs -> PatchRaySampler.reconstruct -> photometric loss -> backward), against the
loss value and gradients the REAL reference produced for the same inputs (tests/golden/train_step.npz).
Tolerances (north_star): loss within 1e-5 absolute; gradients within 2e-5 of the largest entry outside the footprint of fp32-undecidable
relu gates (tests/test_gpu_fused_anchor.py::_check_grads_up_to_gate_events), 1e-3 inside it."""
import ast

import numpy as np
import pytest
import torch

from oracle import bts_loss as OL
from oracle import bts_oracle as O
from tests._cases import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fused_loss", [False, True], ids=["oracle_loss", "hip_loss"])
def test_train_step_loss_and_gradients_vs_reference_golden(fused_loss):
    """fused_loss=False: the oracle's torch restatement of the loss consumes the HIP renderer's outputs (isolates the renderer);
    fused_loss=True: the drop-in ReconstructionLoss (bts_photometric_loss, one HIP pass) -- the whole step after the CNN is HIP."""
    import behindthescenes_amd as bts
    from behindthescenes_amd import _lib
    from tests._hip_helpers import load_mlp, make_conf
    _lib.load()
    z = np.load(f"{GOLDEN}/train_step.npz")
    meta = ast.literal_eval(str(z["meta"]))
    t = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}
    n, pc, ps, K, H, W = meta["n"], meta["patches"], meta["patch"], meta["K"], meta["H"], meta["W"]
    cfg = O.FieldConfig(d_min=meta["d_min"], d_max=meta["d_max"])
    net = bts.BTSNet(make_conf(cfg, meta["C"], meta["Hd"], 0, H, W))
    load_mlp(net, O.MlpParams(t["w_in"], t["b_in"], [], t["w_out"], t["b_out"]))
    with torch.no_grad():
        net.encoder.feats[0].data = t["feat"].clone()
    net = net.cuda().train()
    images = t["images"].cuda()
    net.encode(images, t["projs"].cuda(), t["poses"].cuda(), ids_encoder=[0], ids_render=meta["ids_render"], images_alt=images * .5 + .5)
    renderer = bts.NeRFRenderer(n_coarse=K, lindisp=True, hard_alpha_cap=True).cuda().train()
    sampler = bts.PatchRaySampler(ray_batch_size=pc * ps * ps, z_near=cfg.d_min, z_far=cfg.d_max, patch_size=ps)
    rays = t["rays"].cuda()
    comp = renderer.composite(net, rays.reshape(-1, 8), t["z_samp"].cuda(), coarse=True, sb=n)
    out = renderer._format_outputs(comp, n, want_weights=True, want_alphas=True, want_z_samps=False, want_rgb_samps=True)
    rd = dict(coarse=out, rgb_gt=t["rgb_gt"].cuda())
    rd["fine"] = dict(rd["coarse"])
    rd = sampler.reconstruct(rd)
    c = rd["coarse"]
    # the renderer's outputs in the layout the loss consumes
    assert c["rgb"].shape == t["out_rgb"].shape and c["invalid"].shape == t["out_invalid"].shape
    torch.testing.assert_close(c["depth"].detach().cpu(), t["out_depth"], rtol=1e-4, atol=0)
    torch.testing.assert_close(c["rgb"].detach().cpu(), t["out_rgb"], rtol=0, atol=1e-5)
    if fused_loss:
        crit = bts.ReconstructionLoss({"criterion": "l1+ssim", "invalid_policy": "weight_guided", "lambda_edge_aware_smoothness": 0.001})
        loss, parts = crit(dict(coarse=[c], fine=[rd["fine"]], rgb_gt=rd["rgb_gt"]))
        assert abs(parts["loss"] - t["loss"].item()) <= 1e-5 and abs(parts["loss_invalid_ratio"] - t["loss_invalid_ratio"].item()) <= 1e-6
        ref_loss, ref_parts = OL.reconstruction_loss({k: v.detach() for k, v in c.items()}, rd["rgb_gt"])
        assert abs(parts["loss_eas"] - ref_parts["loss_eas"].item()) <= 1e-6 and abs(parts["loss_rgb_coarse"] - ref_parts["loss_rgb_coarse"].item()) <= 1e-6
    else:
        loss, parts = OL.reconstruction_loss(c, rd["rgb_gt"])
        assert abs(parts["loss_invalid_ratio"].item() - t["loss_invalid_ratio"].item()) <= 1e-6
    assert abs(loss.item() - t["loss"].item()) <= 1e-5, (loss.item(), t["loss"].item())
    loss.backward()
    # gradients: within 2e-5 of each tensor's largest entry, except the footprint of the relu gates the fp64 evaluation shows to be undecidable
    # in fp32 (on this fixture ONE: unit 30 at one sample -- its four tap texels sit at 1.0e-4, right ON the former flat 1e-4 bar)
    from tests.test_gpu_fused_anchor import _check_grads_up_to_gate_events, _gate_ambiguity, _hip_grads
    golden = {"lin_in.weight": t["g_w_in"], "lin_in.bias": t["g_b_in"], "lin_out.weight": t["g_w_out"], "lin_out.bias": t["g_b_out"], "feat": t["g_feat"]}
    scene = dict(images=t["images"], feat=t["feat"], projs=t["projs"], poses=t["poses"])
    amb, taps = _gate_ambiguity(scene, O.MlpParams(t["w_in"], t["b_in"], [], t["w_out"], t["b_out"]), cfg, t["rays"], t["z_samp"])
    _check_grads_up_to_gate_events("golden, entry by entry", _hip_grads(net), golden, amb, taps)


@pytest.mark.parametrize("policy", ["weight_guided", "strict"])
def test_lean_training_outputs_match_the_full_step(policy):
    """with ``lean_training_outputs`` the render kernel's epilogue hands the loss sum_k w * invalid / max_k invalid per
    ray and view and the per-sample tensors (weights, alphas, invalid, rgb_samps) never reach HBM.  Same jitter, same patches: loss,
    invalid ratio and every gradient must equal the full-output step (weight_guided is also the reference golden's policy)."""
    import behindthescenes_amd as bts
    from behindthescenes_amd import _lib
    from tests._hip_helpers import load_mlp, make_conf
    _lib.load()
    z = np.load(f"{GOLDEN}/train_step.npz")
    meta = ast.literal_eval(str(z["meta"]))
    t = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}
    n, pc, ps, K, H, W = meta["n"], meta["patches"], meta["patch"], meta["K"], meta["H"], meta["W"]
    cfg = O.FieldConfig(d_min=meta["d_min"], d_max=meta["d_max"])
    results = []
    for lean in (False, True):
        net = bts.BTSNet(make_conf(cfg, meta["C"], meta["Hd"], 0, H, W))
        load_mlp(net, O.MlpParams(t["w_in"], t["b_in"], [], t["w_out"], t["b_out"]))
        with torch.no_grad():
            net.encoder.feats[0].data = t["feat"].clone()
        net = net.cuda().train()
        images = t["images"].cuda()
        net.encode(images, t["projs"].cuda(), t["poses"].cuda(), ids_encoder=[0], ids_render=meta["ids_render"], images_alt=images * .5 + .5)
        renderer = bts.NeRFRenderer.from_conf(dict(n_coarse=K, lindisp=True, hard_alpha_cap=True, lean_training_outputs=lean)).cuda().train()
        wrapped = renderer.bind_parallel(net).train()
        sampler = bts.PatchRaySampler(ray_batch_size=pc * ps * ps, z_near=cfg.d_min, z_far=cfg.d_max, patch_size=ps)
        torch.manual_seed(11)                                  # the jitter of sample_coarse
        rd = wrapped(t["rays"].cuda(), want_weights=True, want_alphas=True, want_rgb_samps=True)    # trainer.py:245
        assert ("weights" in rd["coarse"]) != lean and ("invalid_wsum" in rd["coarse"]) == lean
        rd["fine"] = dict(rd["coarse"])
        rd["rgb_gt"] = t["rgb_gt"].cuda()
        rd = sampler.reconstruct(rd)
        crit = bts.ReconstructionLoss({"criterion": "l1+ssim", "invalid_policy": policy, "lambda_edge_aware_smoothness": 0.001})
        loss, parts = crit(dict(coarse=[rd["coarse"]], fine=[rd["fine"]], rgb_gt=rd["rgb_gt"]))
        loss.backward()
        results.append((loss.item(), parts["loss_invalid_ratio"], rd["coarse"]["depth"].detach().clone(),
                        [p.grad.clone() for p in (net.mlp_coarse.lin_in.weight, net.mlp_coarse.lin_out.weight, net.encoder.feats[0])]))
    (l0, r0, d0, g0), (l1, r1, d1, g1) = results
    assert torch.equal(d0, d1)                                 # same kernel arithmetic, with and without the epilogue
    assert abs(l0 - l1) <= 1e-6 and abs(r0 - r1) <= 1e-7, (l0, l1, r0, r1)
    for a, b in zip(g0, g1):
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()    # float atomics: run-to-run summation order
    # eval mode ignores the flag: every requested per-sample tensor is there
    wrapped.eval()
    with torch.no_grad():
        rd = wrapped(t["rays"].cuda(), want_weights=True, want_alphas=True)
    assert "weights" in rd["coarse"] and "invalid" in rd["coarse"]
