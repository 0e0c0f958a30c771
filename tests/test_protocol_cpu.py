"""CPU: the protocol pieces around the hot loop that are plain torch glue (no kernel involved) against fixtures produced by the REAL
reference (tests/golden/gen_golden_protocol.py -> protocol.npz): importance-sampling helpers with the reference's CPU random stream,
the sampling schedule, and the reconstruct() views of the three ray samplers."""
import os

import numpy as np
import pytest
import torch

import behindthescenes_amd as bts

G = {k: torch.from_numpy(v) if v.dtype.kind == "f" else v
     for k, v in np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "protocol.npz")).items()}


@pytest.mark.parametrize("lindisp,tag", [(True, "lin"), (False, "dep")])
def test_importance_sampling_helpers_follow_the_reference_stream(lindisp, tag):
    """sample_coarse_from_dist / sample_fine / sample_fine_depth (nerf.py:125-208) with torch's CPU generator seeded as in the
    fixture: same draws in the same order -> the same depths."""
    r = bts.NeRFRenderer(n_coarse=8, n_fine=6, n_fine_depth=2, lindisp=lindisp, depth_std=0.5)
    rays, wts, depth, zc = G["smp_rays"], G["smp_weights"], G["smp_depth"], G[f"smp_{tag}_zc"]
    torch.manual_seed(6)
    torch.testing.assert_close(r.sample_coarse_from_dist(rays, wts, zc), G[f"smp_{tag}_dist"], rtol=1e-6, atol=1e-6)
    torch.manual_seed(7)
    torch.testing.assert_close(r.sample_fine(rays, wts), G[f"smp_{tag}_fine"], rtol=1e-6, atol=1e-6)
    torch.manual_seed(8)
    torch.testing.assert_close(r.sample_fine_depth(rays, depth), G[f"smp_{tag}_fdepth"], rtol=1e-6, atol=1e-6)


def test_sched_step_trace_matches_the_reference():
    r = bts.NeRFRenderer(n_coarse=4, n_fine=0, sched=[[3, 7], [8, 16], [0, 4]])
    trace = []
    for step in range(10):
        r.sched_step(1 if step % 3 else 2)
        trace.append([int(r.iter_idx), int(r.last_sched), r.n_coarse, r.n_fine, int(r.using_fine)])
    assert np.array_equal(np.array(trace), G["sched_trace"]), (trace, G["sched_trace"].tolist())


@pytest.mark.parametrize("kind", ["img", "rnd"])
def test_reconstruct_views_match_the_reference(kind):
    """ImageRaySampler.reconstruct (ray_sampler.py:262-321) and RandomRaySampler.reconstruct (:52-106): same keys, shapes, values."""
    part = {k[len(kind) + 4:]: G[k].clone() for k in G if k.startswith(f"{kind}_in_")}
    if kind == "img":
        s = bts.ImageRaySampler(3.0, 80.0, 6, 10)
        gt = G["img_gt"]
    else:
        s = bts.RandomRaySampler(ray_batch_size=37, z_near=3.0, z_far=80.0)
        gt = G["rnd_gt"]
    rd = s.reconstruct(dict(coarse=dict(part), fine=dict(part), rgb_gt=gt.clone()))
    want = {k[len(kind) + 5:]: G[k] for k in G if k.startswith(f"{kind}_out_")}
    assert set(rd["coarse"]) == set(want) - {"rgb_gt"}
    for k, t in rd["coarse"].items():
        assert t.shape == want[k].shape, (k, t.shape, want[k].shape)
        assert torch.equal(t, want[k]), k
        assert torch.equal(rd["fine"][k], want[k]), k
    assert rd["rgb_gt"].shape == want["rgb_gt"].shape and torch.equal(rd["rgb_gt"], want["rgb_gt"])


def test_lean_training_outputs_reconstruct_and_config():
    """NeRFRenderer.lean_training_outputs: a config key, off by default; the samplers' reconstruct views the per-ray
    reductions of the lean dict (no per-sample tensors) in the patch layout."""
    import behindthescenes_amd as bts
    assert bts.NeRFRenderer.from_conf(dict(n_coarse=8)).lean_training_outputs is False
    assert bts.NeRFRenderer.from_conf(dict(n_coarse=8, lean_training_outputs=True)).lean_training_outputs is True
    n, pc, ps, nv = 2, 3, 8, 4
    B = pc * ps * ps
    sampler = bts.PatchRaySampler(ray_batch_size=B, z_near=3.0, z_far=80.0, patch_size=ps)
    sampler._patch_count = pc
    lean = dict(rgb=torch.randn(n, B, nv * 3), depth=torch.randn(n, B), invalid_wsum=torch.rand(n, B, nv), invalid_any=torch.rand(n, B, nv).round())
    rd = dict(coarse=dict(lean), fine=dict(lean), rgb_gt=torch.rand(n, B, 3))
    out = sampler.reconstruct(rd)
    for key in ("coarse", "fine"):
        c = out[key]
        assert c["rgb"].shape == (n, pc, ps, ps, nv, 3) and c["depth"].shape == (n, pc, ps, ps)
        assert c["invalid_wsum"].shape == (n, pc, ps, ps, nv) and c["invalid_any"].shape == (n, pc, ps, ps, nv)
        assert torch.equal(c["invalid_wsum"].reshape(n, B, nv), lean["invalid_wsum"])
        assert "weights" not in c and "invalid" not in c
    assert out["rgb_gt"].shape == (n, pc, ps, ps, 3)


def test_packed_parameter_vector_is_cached_until_a_parameter_changes():
    """ResnetFC.packed(): one flat vector (and one autograd split of its gradient) however many renders of a step read it; a new one
    after an optimizer step / load_state_dict (version counter), a change of grad mode, or invalidate_packed() (BTSNet.encode)."""
    import copy
    from behindthescenes_amd.mlp import ResnetFC
    m = ResnetFC(71, n_blocks=1, d_hidden=32)
    a = m.packed()
    assert m.packed() is a and a.requires_grad
    with torch.no_grad():
        b = m.packed()
    assert b is not a and not b.requires_grad
    (a * torch.arange(a.numel(), dtype=a.dtype)).sum().backward()          # the split lands on the individual parameters
    assert all(p.grad is not None for p in m.parameters())
    n_in = m.lin_in.weight.numel()
    torch.testing.assert_close(m.lin_in.weight.grad.reshape(-1), torch.arange(n_in, dtype=a.dtype))
    with torch.no_grad():
        m.lin_out.bias.add_(1.0)                                          # what an optimizer step does
    c = m.packed()
    assert c is not a and float(c[-1].detach()) == float(m.lin_out.bias[-1].detach())
    m.invalidate_packed()
    assert m.packed() is not c
    m2 = copy.deepcopy(m)                                                 # the cache (a non-leaf tensor) stays behind
    assert m2.__dict__["_packed_cache"] is None and torch.equal(m2.packed(), m.packed())


def test_loss_dict_is_a_lazy_mapping_that_behaves_like_the_reference_dict():
    import copy
    import json
    import pickle
    from behindthescenes_amd.loss import LazyScalars
    d = LazyScalars(["loss", "loss_rgb_coarse"], torch.tensor([1.5, 0.25]))
    assert len(d) == 2 and list(d) == ["loss", "loss_rgb_coarse"]                      # (no materialisation needed for these)
    assert d["loss"] == 1.5 and dict(d) == {"loss": 1.5, "loss_rgb_coarse": 0.25}
    assert "loss" in d and "nope" not in d and isinstance(d["loss_rgb_coarse"], float)
    d["lr"] = 1e-4                                                                     # engine handlers add entries (dict semantics)
    d.update(loss=2.0)
    assert d["loss"] == 2.0 and list(d.items())[-1] == ("lr", 1e-4) and len(d) == 3
    for clone in (pickle.loads(pickle.dumps(d)), copy.deepcopy(d)):                    # torch.save of an engine's output
        assert type(clone) is dict and clone == dict(d)
    assert json.loads(json.dumps(dict(d)))["loss"] == 2.0


def test_scale_maps_keep_their_size_only_for_power_of_two_ratios():
    """BTSNet._scale_shift: the decoder's scale s goes to the renderer at its own size (BtsFieldCfg.feat_shift = s) when scale 0 is
    exactly 2^s times larger; any other ratio is resized like the reference does (models_bts.py:115-117)."""
    conf = dict(z_near=3, z_far=80, code=dict(num_freqs=6, freq_factor=1.5, include_input=True),
                encoder=dict(type="feature_map", size=(16, 32), d_out=64), mlp_coarse=dict(type="resnet", n_blocks=0, d_hidden=64),
                mlp_fine=dict(type="empty"))
    net = bts.BTSNet(conf)
    assert net._scale_shift((16, 32), (16, 32)) == 0 and net._scale_shift((8, 16), (16, 32)) == 1 and net._scale_shift((2, 4), (16, 32)) == 3
    assert net._scale_shift((8, 15), (16, 32)) is None and net._scale_shift((5, 10), (16, 32)) is None
    net_r = bts.BTSNet(dict(conf, native_scale_maps=False))
    assert net_r._scale_shift((8, 16), (16, 32)) is None and net_r._scale_shift((16, 32), (16, 32)) == 0


@pytest.mark.parametrize("n_scales,with_fine,eas", [(1, True, 0.001), (4, True, 0.001), (4, False, 0.0), (2, True, 0.0)])
def test_loss_algebra_as_one_matrix_equals_the_spelled_out_path(monkeypatch, n_scales, with_fine, eas):
    """ReconstructionLoss._call_photometric_only (loss, logging dict and their gradient as ONE matrix applied to the per-scale sums of the
    HIP pass) against the general path that spells the reference's algebra out (loss.py:219-293) -- on the CPU, with the HIP pass replaced
    by a differentiable stand-in that returns (sum of rgb errors, sum of a smoothness term, invalid count)."""
    from behindthescenes_amd import loss as L

    class FakeSums:
        @staticmethod
        def apply(rgb, depth, weights, invalid, rgb_gt, ph, pw, policy, eas_on, invalid_wsum=None, invalid_any=None):
            e = (rgb.reshape(rgb.shape[0], -1, 3) - rgb_gt[:, None, :]).abs().sum()
            s = (depth ** 2).sum() * 0.01 if (eas_on and depth is not None) else rgb.sum() * 0.0
            return torch.stack([e, s, (invalid_wsum > 0.9).all(-1).float().sum().detach()])

    monkeypatch.setattr(L, "_PhotometricSums", FakeSums)
    g = torch.Generator().manual_seed(n_scales)
    n, pc, h, w, nv = 2, 3, 8, 8, 2

    def data(req):
        coarse = []
        for s in range(n_scales):
            gs = torch.Generator().manual_seed(100 + s)
            c = dict(rgb=torch.rand(n, pc, h, w, nv, 3, generator=gs).requires_grad_(req), depth=(torch.rand(n, pc, h, w, generator=gs) * 10).requires_grad_(req),
                     invalid_wsum=torch.rand(n, pc, h, w, nv, generator=gs) * 1.2, invalid_any=torch.zeros(n, pc, h, w, nv))
            coarse.append(c)
        fine = [dict(c) for c in coarse] if with_fine else [dict() for _ in coarse]   # trainer.py:247-248: fine = dict(coarse)
        return dict(coarse=coarse, fine=fine, rgb_gt=torch.rand(n, pc, h, w, 3, generator=torch.Generator().manual_seed(7)))

    crit = L.ReconstructionLoss({"criterion": "l1+ssim", "invalid_policy": "weight_guided", "lambda_edge_aware_smoothness": eas, "lambda_coarse": 0.7,
                                 "lambda_fine": 1.3})
    d1 = data(True)
    loss1, parts1 = crit(d1)
    loss1.backward()
    monkeypatch.setattr(L.ReconstructionLoss, "_call_photometric_only", lambda self, data: None)     # force the general path
    d2 = data(True)
    loss2, parts2 = crit(d2)
    loss2.backward()
    assert abs(loss1.item() - loss2.item()) <= 1e-6 * max(1.0, abs(loss2.item()))
    assert list(parts1) == list(parts2)
    for k in parts2:
        assert abs(parts1[k] - parts2[k]) <= 1e-6 * max(1.0, abs(parts2[k])), k
    for a, b in zip(d1["coarse"], d2["coarse"]):
        torch.testing.assert_close(a["rgb"].grad, b["rgb"].grad, rtol=1e-5, atol=1e-9)
        if eas > 0:
            torch.testing.assert_close(a["depth"].grad, b["depth"].grad, rtol=1e-5, atol=1e-9)


def test_fused_train_step_says_why_a_configuration_takes_the_entry_by_entry_path():
    """behindthescenes_amd.FusedTrainStep (ABI 7: the training step in two library calls) covers the shipped training configurations and
    names the condition that sends any other one through the reference's call sequence; host logic only (no kernel runs here).  The
    loss matrix it hands the library is the criterion's own (rows = the logging dict, row 8 = the loss) and follows the lambdas."""
    import behindthescenes_amd as bts
    from behindthescenes_amd import synthetic as S
    net = bts.BTSNet(S.field_conf(64, 64, 0, 48, 160)).train()
    renderer = bts.NeRFRenderer.from_conf(dict(n_coarse=64, lindisp=True, hard_alpha_cap=True, lean_training_outputs=True)).train()
    sampler = bts.PatchRaySampler(ray_batch_size=256, z_near=3.0, z_far=80.0, patch_size=8)
    crit = bts.ReconstructionLoss({"criterion": "l1+ssim", "invalid_policy": "weight_guided", "lambda_edge_aware_smoothness": 0.001})
    step = bts.FusedTrainStep(renderer.bind_parallel(net).train(), sampler, crit)
    ok = dict(ids_encoder=[0], ids_render=[2, 3], ids_loss=[0, 1])
    assert step.why_not(None, **ok) is None
    assert "float32" in step.why_not(torch.zeros(1, 4, 3, 48, 160), **ok)                       # CPU frames: the library has no CPU path
    assert "encoder view" in step.why_not(None, ids_encoder=[0, 1], ids_render=[2], ids_loss=[3])
    assert "render" in step.why_not(None, ids_encoder=[0], ids_render=list(range(9)), ids_loss=[0])
    renderer.eval()
    assert "training mode" in step.why_not(None, **ok)
    renderer.train()
    renderer.noise_std = 0.5
    assert "noise" in step.why_not(None, **ok)
    renderer.noise_std = 0.0
    crit.lambda_entropy = 0.01
    assert "regulariser" in step.why_not(None, **ok)
    crit.lambda_entropy = 0
    M = crit.loss_matrix(2, (512, 512), (True, True))
    assert M.shape == (9, 6) and abs(M[8, 0].item() - 2.0 / 512 / 2) < 1e-12 and abs(M[8, 4].item() - 0.001 / 2 / 512 / 2) < 1e-12
    crit.lambda_edge_aware_smoothness = 0.01          # a schedule that changes a lambda gets a new matrix (the reference reads it per call)
    assert abs(crit.loss_matrix(2, (512, 512), (True, True))[8, 1].item() - 0.01 / 512 / 2) < 1e-12


def test_patch_draws_into_preallocated_int32_rows_consume_the_generator_like_the_reference_calls():
    """FusedTrainStep draws the patch coordinates straight into a pinned int32 block (``PatchRaySampler.draw_patches(rows=...)``):
    ``t.random_(0, hi)`` per row must give the values -- and leave the generator in the state -- of the reference's
    ``torch.randint(0, hi, (P,))`` calls in the same order (ray_sampler.py:141-143), for int32 as for int64."""
    import behindthescenes_amd as bts
    for P_rays, n, v, h, w in ((2048, 8, 3, 192, 640), (4096, 2, 4, 64, 96), (64, 3, 1, 9, 9)):
        ps = bts.PatchRaySampler(ray_batch_size=P_rays, z_near=3.0, z_far=80.0, patch_size=8)
        torch.manual_seed(123)
        pv, py, px = ps.draw_patches(n, v, h, w)
        after = torch.rand(3)
        P = ps._patch_count
        block = torch.full((3, n, P), -1, dtype=torch.int32)
        rows = [block[j, i] for i in range(n) for j in range(3)]
        torch.manual_seed(123)
        assert ps.draw_patches(n, v, h, w, rows=rows) is None
        assert torch.equal(block[0].long(), pv) and torch.equal(block[1].long(), py) and torch.equal(block[2].long(), px)
        assert torch.equal(torch.rand(3), after)


def test_feature_map_layout_helpers_and_the_stand_in_encoder_keep_the_memory_format():
    """ABI 8's host side: which entry point a map takes is read off its strides (``native.is_channels_last`` / ``as_feature_map``); the
    stand-in encoder can hold its maps in either format and ``set_feature_map`` replaces the data without changing it."""
    import behindthescenes_amd as bts
    from behindthescenes_amd import native, synthetic as S
    x = torch.randn(2, 8, 5, 7)
    xc = x.contiguous(memory_format=torch.channels_last)
    assert not native.is_channels_last(x) and native.is_channels_last(xc)
    assert not native.is_channels_last(torch.randn(2, 1, 5, 7).contiguous(memory_format=torch.channels_last))   # also plain contiguous
    assert native.as_feature_map(x) is x and native.as_feature_map(xc) is xc                                     # no copies
    sl = x[:, :, ::2]                                                                                            # neither: one dense copy
    y = native.as_feature_map(sl)
    assert y.is_contiguous() and torch.equal(y, sl)
    assert native.as_feature_map(xc.half()).dtype == torch.float32 and native.is_channels_last(native.as_feature_map(xc.half()))
    for cl in (False, True):
        enc = bts.FeatureMapEncoder((6, 10), 8, num_views=2, n_scales=2, pyramid=True, channels_last=cl)
        assert [tuple(p.shape) for p in enc.feats] == [(2, 8, 6, 10), (2, 8, 3, 5)]
        assert all(native.is_channels_last(p) == cl for p in enc.feats)

        class Net:                                   # what set_feature_map touches
            encoder = enc
        S.set_feature_map(Net, torch.arange(3 * 8 * 6 * 10, dtype=torch.float32).reshape(3, 8, 6, 10))          # another batch size
        p = enc.feats[0]
        assert tuple(p.shape) == (3, 8, 6, 10) and native.is_channels_last(p) == cl and p[2, 7, 5, 9].item() == 3 * 8 * 6 * 10 - 1
