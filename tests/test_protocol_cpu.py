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
