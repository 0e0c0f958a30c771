"""GPU: the fused photometric loss (bts_photometric_loss through the drop-in ReconstructionLoss) against the CPU oracle restatement
of the reference's loss (oracle/bts_loss.py, itself pinned to the real reference by tests/golden/train_step.npz): value within 1e-5,
gradients with respect to rgb and depth within 1e-4 of the largest entry, for every invalid policy, odd patch shapes, several
render views, depths outside the clamp range and rays that are entirely invalid."""
import pytest
import torch
import torch.nn.functional as F

from oracle import bts_loss as OL

pytestmark = pytest.mark.gpu


def _inputs(n, pc, h, w, nv, K, seed):
    g = torch.Generator().manual_seed(seed)
    gt = F.avg_pool2d(torch.rand(n * pc, 3, h + 4, w + 4, generator=g), 3, 1, 1)[:, :, 2:-2, 2:-2]          # correlated like real frames
    gt = gt.reshape(n, pc, 3, h, w).permute(0, 1, 3, 4, 2).contiguous()
    rgb = (gt.unsqueeze(-2) + 0.15 * torch.randn(n, pc, h, w, nv, 3, generator=g)).clamp(0, 1)
    rgb[:, 0, :, :, 0] = gt[:, 0]                                  # an exact match: SSIM term 0, L1 gradient sign(0) = 0
    depth = torch.rand(n, pc, h, w, generator=g) * 100 + 0.5       # beyond the [1e-3, 80] clamp on purpose
    depth[:, 1, 0, 0] = 1e-4
    wts = torch.rand(n, pc, h, w, K, generator=g)
    wts = wts / wts.sum(-1, keepdim=True)
    inv = (torch.rand(n, pc, h, w, K, nv, generator=g) < 0.3).float()
    inv[:, :, 0, :, :, :] = 1.0                                     # a row of rays invalid in every view
    inv[:, :, 1, :, : K // 2, 0] = 0.0                              # ... and one valid in view 0 only
    alphas = torch.rand(n, pc, h, w, K, generator=g)
    return rgb, depth, wts, inv, alphas, gt


@pytest.mark.parametrize("policy", ["weight_guided", "strict", "none"])
@pytest.mark.parametrize("shape", [(2, 5, 8, 8, 3, 12), (1, 3, 4, 6, 1, 7), (1, 2, 2, 32, 2, 5)], ids=["8x8", "4x6", "2x32"])
def test_fused_loss_vs_oracle(policy, shape):
    import behindthescenes_amd as bts
    n, pc, h, w, nv, K = shape
    rgb, depth, wts, inv, alphas, gt = _inputs(n, pc, h, w, nv, K, seed=h * 100 + nv)
    # oracle (CPU autograd)
    r0, d0 = rgb.clone().requires_grad_(True), depth.clone().requires_grad_(True)
    ref, ref_parts = OL.reconstruction_loss(dict(rgb=r0, depth=d0, weights=wts, invalid=inv, alphas=alphas), gt, invalid_policy=policy,
                                            lambda_eas=0.01)
    ref.backward()
    # HIP
    r1, d1 = rgb.cuda().requires_grad_(True), depth.cuda().requires_grad_(True)
    level = dict(rgb=r1, depth=d1, weights=wts.cuda(), invalid=inv.cuda(), alphas=alphas.cuda())
    crit = bts.ReconstructionLoss({"criterion": "l1+ssim", "invalid_policy": policy, "lambda_edge_aware_smoothness": 0.01})
    loss, parts = crit(dict(coarse=[level], fine=[dict(level)], rgb_gt=gt.cuda()))
    loss.backward()
    assert abs(loss.item() - ref.item()) <= 1e-5, (loss.item(), ref.item())
    assert abs(parts["loss_invalid_ratio"] - ref_parts["loss_invalid_ratio"].item()) <= 1e-6
    assert abs(parts["loss_eas"] - ref_parts["loss_eas"].item()) <= 1e-5 * max(1.0, abs(ref_parts["loss_eas"].item()))
    for got, want, name in ((r1.grad.cpu(), r0.grad, "rgb"), (d1.grad.cpu(), d0.grad, "depth")):
        err = (got - want).abs().max().item() / want.abs().max().clamp_min(1e-20).item()
        assert err <= 1e-4, (name, err)


def test_regularisers_and_multiscale_vs_oracle():
    """alpha regulariser + ray entropy on top of the fused term, two scales with the scale-0 invalid mask (loss.py:95-118, 259-267)."""
    import behindthescenes_amd as bts
    n, pc, h, w, nv, K = 2, 3, 8, 8, 2, 9
    lv = []
    for s in range(2):
        rgb, depth, wts, inv, alphas, gt = _inputs(n, pc, h, w, nv, K, seed=40 + s)
        lv.append(dict(rgb=rgb, depth=depth, weights=wts, invalid=inv, alphas=alphas))
    gt = _inputs(n, pc, h, w, nv, K, seed=40)[5]
    want = 0.0
    for s in range(2):
        level = dict(lv[s], weights=lv[0]["weights"], invalid=lv[0]["invalid"])       # the reference masks every scale with scale 0's rays
        l, parts = OL.reconstruction_loss(level, gt, invalid_policy="weight_guided", lambda_eas=0.01 / 2 ** s, lambda_alpha_reg=0.05)
        want = want + l
    want = want / 2
    crit = bts.ReconstructionLoss({"criterion": "l1+ssim", "invalid_policy": "weight_guided", "lambda_edge_aware_smoothness": 0.01,
                                   "lambda_alpha_reg": 0.05})
    cu = [{k: v.cuda() for k, v in level.items()} for level in lv]
    loss, parts = crit(dict(coarse=cu, fine=[dict(c) for c in cu], rgb_gt=gt.cuda()))
    assert abs(loss.item() - want.item()) <= 2e-5, (loss.item(), want.item())
