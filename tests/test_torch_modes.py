"""the two field modes no shipped config uses -- merged encoder views (``combine_ids``) and MLP-predicted
colours (``sample_color: false``) -- are served by PyTorch compositions (behindthescenes_amd/torch_modes.py), as the survey prescribes.
Pinned to the REAL reference's outputs and autograd gradients (tests/golden/modes.npz, tests/golden/gen_golden_modes.py), on the CPU
here and on the GPU in the ``-m gpu`` run (the same ops on PyTorch-ROCm)."""
import ast
import os

import numpy as np
import pytest
import torch

import behindthescenes_amd as bts

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "modes.npz")


def _case(name, device):
    z = np.load(GOLDEN)
    t = {k[len(name) + 1:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(name + "_") and not k.endswith("_meta")}
    meta = ast.literal_eval(str(z[name + "_meta"]))
    conf = dict(z_near=3.0, z_far=80.0, inv_z=True, learn_empty=True, code_mode="z", sample_color=meta["sample_color"],
                code=dict(num_freqs=6, freq_factor=1.5, include_input=True),
                encoder=dict(type="feature_map", size=(meta["H"], meta["W"]), d_out=meta["C"], num_views=t["feats"].shape[0]),
                mlp_coarse=dict(type="resnet", n_blocks=0, d_hidden=meta["Hd"]), mlp_fine=dict(type="empty"))
    with pytest.warns(UserWarning, match="PyTorch composition") if not bts.torch_modes._warned else _nullcontext():
        net = bts.BTSNet(conf)
        with torch.no_grad():
            net.encoder.feats[0].copy_(t["feats"]), net.empty_feature.copy_(t["empty"])
            m = net.mlp_coarse
            m.lin_in.weight.copy_(t["w_in"]), m.lin_in.bias.copy_(t["b_in"]), m.lin_out.weight.copy_(t["w_out"]), m.lin_out.bias.copy_(t["b_out"])
        net = net.to(device).eval()
        net.encode(t["images"].to(device), t["projs"].to(device), t["poses"].to(device), ids_encoder=meta["ids_encoder"],
                   ids_render=meta["ids_render"], combine_ids=meta["combine_ids"])
    return net, t, meta


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def _check(name, device):
    net, t, meta = _case(name, device)
    assert net.torch_mode and net._d_out == (1 if meta["sample_color"] else 4)
    renderer = bts.NeRFRenderer(n_coarse=meta["K"], lindisp=True, hard_alpha_cap=True).to(device).eval()
    rays, z = t["rays"].to(device), t["z"].to(device)
    w, rgb, depth, alphas, invalid, _, rgbs = renderer.composite(net, rays.reshape(-1, 8), z, coarse=True, sb=meta["n"])
    # On the GPU the same torch ops round differently (rocBLAS instead of the CPU's GEMM), and these modes SELECT: a point within rounding
    # of a frustum border flips its flag and with it the view its features come from (combine) or its empty feature -- its ray then
    # differs by a finite amount.  CPU: every entry within 1e-5.  GPU: the rays whose flags agree with the reference's, which must be
    # nearly all of them, within 5e-5 (alpha = 1 - exp(-delta sigma) carries the rounding into the per-sample tensors).
    cpu = device.type == "cpu"
    flags_same = (invalid.cpu() == t["invalid"]).all(-1).all(-1)                  # per ray
    assert flags_same.float().mean().item() >= (1.0 if cpu else 0.97)
    ps = 1e-5 if cpu else 5e-5
    for got, key, tol in ((w, "weights", ps), (rgb, "rgb", 1e-5 if cpu else 2e-5), (alphas, "alphas", ps), (rgbs, "rgb_samps", 1e-5)):
        assert got.shape == t[key].shape, key
        err = (got.detach().cpu() - t[key]).abs().reshape(got.shape[0], -1).amax(-1)
        # (combine also selects among the ENCODER views by flags the outputs do not show: a few per cent more rays may differ there)
        assert (err[flags_same] <= tol).float().mean().item() >= (1.0 if cpu else 0.9), (name, key, err[flags_same].max().item())
    rel = (depth.detach().cpu() - t["depth"]).abs() / t["depth"].abs()
    assert (rel[flags_same] <= 1e-4).float().mean().item() >= (1.0 if cpu else 0.9)         # north_star's depth bar
    # autograd through the composition = the reference's gradients
    loss = (rgb * t["gin_rgb"].to(device)).sum() + (depth * t["gin_depth"].to(device)).sum()
    m = net.mlp_coarse
    params = {"g_lin_in_weight": m.lin_in.weight, "g_lin_in_bias": m.lin_in.bias, "g_lin_out_weight": m.lin_out.weight, "g_lin_out_bias": m.lin_out.bias,
              "g_feats": net.encoder.feats[0], "g_empty": net.empty_feature}
    grads = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
    for (k, _), g in zip(params.items(), grads):
        ref = t[k]
        if g is None:
            assert ref.numel() == 1 and float(ref) == 0.0, k
            continue
        assert (g.cpu() - ref).abs().max().item() <= (1e-4 if cpu else 2e-2) * (ref.abs().max().item() + 1e-20), (name, k)   # (GPU: a flipped ray moves a gradient)
    # the field protocol on raw points (BTSNet.forward)
    with torch.no_grad():
        q_rgb, q_inv, q_sig = net(t["q_pts"].to(device))
    same = (q_inv.cpu() == t["q_invalid"]).all(-1)
    assert same.float().mean().item() >= (1.0 if device.type == "cpu" else 0.999)
    assert (q_rgb.cpu() - t["q_rgb"])[same].abs().max().item() <= 1e-5 and (q_sig.cpu() - t["q_sigma"]).abs().max().item() <= 1e-4 * t["q_sigma"].abs().max().item()
    # through the wrapper: the reference's output dict
    out = renderer.bind_parallel(net).eval()(rays, want_weights=True, want_alphas=True, want_rgb_samps=True)
    assert out["coarse"]["rgb"].shape == (meta["n"], rays.shape[1], rgb.shape[-1]) and set(out["coarse"]) >= {"rgb", "depth", "invalid", "weights", "alphas", "rgb_samps"}
    with pytest.raises(bts.BtsNativeError):
        net.native_field()          # no state in the fused kernels' layouts: loud, not silent


@pytest.mark.parametrize("name", ["combine", "mlpcolor"])
def test_unshipped_field_modes_match_the_reference_cpu(name):
    _check(name, torch.device("cpu"))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["combine", "mlpcolor"])
def test_unshipped_field_modes_match_the_reference_gpu(name):
    _check(name, torch.device("cuda"))
