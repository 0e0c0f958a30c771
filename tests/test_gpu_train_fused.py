"""GPU: ``FusedTrainStep`` (bts_train_step_fwd / bts_train_step_bwd, ABI 7: a training step's share of the renderer in two calls) against
the entry-by-entry sequence of the reference's trainer (models/bts/trainer.py:208-259 + the criterion call) on the same seeds.

The two paths run the same kernels with the same arguments: every forward output -- rays, patch colours, rgb, depth, the invalid-ray
reductions -- must be bit-identical; the loss and the logging dict differ only in the order of the final sums (1e-6 relative); the
gradients only in the order of the backward's float atomics (2e-5 of the largest entry, the bound of tests/test_gpu_scales.py).  The
entry-by-entry path itself is pinned to the real reference by tests/test_gpu_train_step.py (golden loss and gradients)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = {
    # exp_kitti_raw.yaml / exp_kitti_360.yaml in small: plain MLP 64 -> 64, hard alpha cap, one render per step
    "kitti": dict(n=2, V=4, H=48, W=160, C=64, HD=64, NB=0, K=64, ids_loss=[0, 1], ids_render=[2, 3], rays=512, z=(3.0, 80.0), hard_cap=True,
                  code_mode="z", scales=1, policy="weight_guided"),
    # non-consecutive frame ids (kitti360-mono's pattern, trainer.py:147-157), the encoder frame among the render views, strict policy
    "kitti_mono": dict(n=2, V=8, H=48, W=160, C=64, HD=64, NB=0, K=64, ids_loss=[1, 2, 5, 6], ids_render=[0, 3, 4, 7], rays=256, z=(3.0, 80.0),
                       hard_cap=True, code_mode="z", scales=1, policy="strict"),
    # exp_re10k.yaml in small: one ResnetBlockFC of width 32, distance code, four renders per step on a feature pyramid, K = 48 (48-lane mode)
    "re10k": dict(n=3, V=3, H=64, W=96, C=32, HD=32, NB=1, K=48, ids_loss=[0], ids_render=[1, 2], rays=256, z=(1.0, 100.0), hard_cap=False,
                  code_mode="distance", scales=4, policy="weight_guided"),
    # K > 64 (BASELINE.json's 128 samples for RE10K): the row passes with several chunks per ray
    "re10k_k128": dict(n=2, V=3, H=64, W=96, C=32, HD=32, NB=1, K=128, ids_loss=[0], ids_render=[1, 2], rays=128, z=(1.0, 100.0), hard_cap=False,
                       code_mode="distance", scales=2, policy="weight_guided"),
}


def _setup(cfg, learn_empty=False, seed=5):
    import behindthescenes_amd as bts
    from behindthescenes_amd import synthetic as S
    from behindthescenes_amd.train_step import FusedTrainStep
    dev = torch.device("cuda")
    scene = S.synthetic_scene(cfg["n"], cfg["V"], cfg["H"], cfg["W"], cfg["C"], seed=seed, baseline=0.4, smooth=True)
    conf = S.field_conf(cfg["C"], cfg["HD"], cfg["NB"], cfg["H"], cfg["W"], z_near=cfg["z"][0], z_far=cfg["z"][1], code_mode=cfg["code_mode"],
                        learn_empty=learn_empty)
    torch.manual_seed(11)
    net = bts.BTSNet(conf)
    net.encoder = bts.FeatureMapEncoder((cfg["H"], cfg["W"]), cfg["C"], num_views=cfg["n"], n_scales=cfg["scales"], pyramid=cfg["scales"] > 1)
    S.init_mlp_(net.mlp_coarse, seed=7)
    net = net.to(dev).train()
    renderer = bts.NeRFRenderer.from_conf(dict(n_coarse=cfg["K"], lindisp=True, hard_alpha_cap=cfg["hard_cap"], lean_training_outputs=True)).to(dev).train()
    sampler = bts.PatchRaySampler(ray_batch_size=cfg["rays"], z_near=cfg["z"][0], z_far=cfg["z"][1], patch_size=8)
    crit = bts.ReconstructionLoss({"criterion": "l1+ssim", "invalid_policy": cfg["policy"], "lambda_edge_aware_smoothness": 0.001})
    step = FusedTrainStep(renderer.bind_parallel(net).train(), sampler, crit, multiscale=cfg["scales"] > 1)
    return step, net, [scene[k].to(dev) for k in ("images", "projs", "poses")]


def _run(step, net, inputs, cfg, fused, seed=3, scale=None):
    step.fused = fused
    net.zero_grad(set_to_none=True)
    torch.manual_seed(seed)          # the CPU generator (flip, patches) and the device generator (jitter) both start over
    loss, loss_dict, data = step(*inputs, ids_encoder=[0], ids_render=cfg["ids_render"], ids_loss=cfg["ids_loss"])
    assert step.last_path == ("fused" if fused else "entries: switched off (fused=False)"), step.last_path
    (loss if scale is None else loss * scale).backward()
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    return loss.detach().clone(), dict(loss_dict), data, grads


@pytest.mark.parametrize("shape", list(SHAPES))
def test_fused_step_equals_the_entry_by_entry_step(shape):
    cfg = SHAPES[shape]
    step, net, inputs = _setup(cfg)
    l_e, d_e, data_e, g_e = _run(step, net, inputs, cfg, fused=False)
    l_f, d_f, data_f, g_f = _run(step, net, inputs, cfg, fused=True)
    # forward: the same kernels on the same inputs
    assert torch.equal(data_e["rays"], data_f["rays"]) and torch.equal(data_e["rgb_gt"], data_f["rgb_gt"])
    assert len(data_e["coarse"]) == len(data_f["coarse"]) == cfg["scales"]
    for ce, cf in zip(data_e["coarse"], data_f["coarse"]):
        for k in ("rgb", "depth", "invalid_wsum", "invalid_any"):
            assert ce[k].shape == cf[k].shape, (k, ce[k].shape, cf[k].shape)
            assert torch.equal(ce[k].detach(), cf[k]), (shape, k, (ce[k].detach() - cf[k]).abs().max().item())
    assert abs(l_e.item() - l_f.item()) <= 1e-6 * max(1.0, abs(l_e.item())), (l_e.item(), l_f.item())
    assert set(d_e) == set(d_f)
    for k in d_e:
        assert abs(d_e[k] - d_f[k]) <= 1e-6 * max(1.0, abs(d_e[k])), (k, d_e[k], d_f[k])
    # backward: the same passes; only the order of the float atomics differs
    assert set(g_e) == set(g_f) and len(g_e) >= 4 + cfg["scales"]
    for k in g_e:
        top = g_e[k].abs().max().item()
        assert top > 0, k
        err = (g_e[k] - g_f[k]).abs().max().item() / top
        assert err <= 2e-5, (shape, k, err)


def test_concurrent_scales_equal_serial_scales():
    """BtsTrainStep.concurrent_scales: the four scales' chains of an exp_re10k.yaml step on side queues of the library against the same
    chains one after the other -- identical forward bits, gradients to the order of the float atomics."""
    cfg = SHAPES["re10k"]
    step, net, inputs = _setup(cfg)
    step.concurrent_scales = False
    l_s, d_s, data_s, g_s = _run(step, net, inputs, cfg, fused=True)
    step.concurrent_scales = True
    for _ in range(3):      # (several rounds: an ordering bug between the queues would show up as a changing result)
        l_c, d_c, data_c, g_c = _run(step, net, inputs, cfg, fused=True)
        assert torch.equal(l_s, l_c) and all(d_s[k] == d_c[k] for k in d_s)
        for a, b in zip(data_s["coarse"], data_c["coarse"]):
            assert all(torch.equal(a[k], b[k]) for k in ("rgb", "depth", "invalid_wsum", "invalid_any"))
        for k in g_s:
            assert (g_s[k] - g_c[k]).abs().max().item() <= 2e-5 * g_s[k].abs().max().item(), k


def test_fused_step_with_the_learned_empty_feature_and_an_upstream_factor():
    """learn_empty (models_bts.py:176-182): the empty feature's gradient and its share of lin_in's come out of the second call; an upstream
    gradient other than 1 (a scaled loss) reaches every gradient."""
    cfg = SHAPES["kitti"]
    step, net, inputs = _setup(cfg, learn_empty=True)
    l_e, _, _, g_e = _run(step, net, inputs, cfg, fused=False, scale=3.0)
    l_f, _, _, g_f = _run(step, net, inputs, cfg, fused=True, scale=3.0)
    assert "empty_feature" in g_e and "empty_feature" in g_f
    assert abs(l_e.item() - l_f.item()) <= 1e-6
    for k in g_e:
        top = g_e[k].abs().max().item()
        assert top > 0 and (g_e[k] - g_f[k]).abs().max().item() / top <= 2e-5, k
    _, _, _, g_1 = _run(step, net, inputs, cfg, fused=True)
    k = "mlp_coarse.lin_in.weight"
    assert (g_f[k] - 3.0 * g_1[k]).abs().max().item() <= 2e-5 * g_f[k].abs().max().item()


def test_fused_step_state_is_clean_between_steps_and_without_grad():
    """The kept (d_proj, tile flags) pairs are all zero again after a step; a step under no_grad leaves no saved state behind; two
    forwards whose backwards come later each keep their own state (gradient accumulation)."""
    from behindthescenes_amd import train_step as TS
    cfg = SHAPES["kitti"]
    TS.release_arenas()           # (arenas of the other tests' shapes)
    step, net, inputs = _setup(cfg)
    _, _, _, g1 = _run(step, net, inputs, cfg, fused=True)
    assert sum(len(p) for p in TS._ARENAS.values()) == 1
    for pool in TS._ARENAS.values():
        for a in pool:
            assert a.busy is None
            for sc in a.scales:
                assert not sc["d_proj"].any() and not sc["d_tiles"].any()
    with torch.no_grad():
        torch.manual_seed(3)
        loss, _, _ = step(*inputs, ids_encoder=[0], ids_render=cfg["ids_render"], ids_loss=cfg["ids_loss"])
    assert not loss.requires_grad and all(a.busy is None for pool in TS._ARENAS.values() for a in pool)
    # two steps in flight
    net.zero_grad(set_to_none=True)
    torch.manual_seed(3)
    la, _, _ = step(*inputs, ids_encoder=[0], ids_render=cfg["ids_render"], ids_loss=cfg["ids_loss"])
    torch.manual_seed(3)
    lb, _, _ = step(*inputs, ids_encoder=[0], ids_render=cfg["ids_render"], ids_loss=cfg["ids_loss"])
    assert sum(len(p) for p in TS._ARENAS.values()) == 2
    lb.backward(), la.backward()
    k = "mlp_coarse.lin_in.weight"
    got = dict(net.named_parameters())[k].grad
    assert (got - 2.0 * g1[k]).abs().max().item() <= 4e-5 * got.abs().max().item()
    TS.release_arenas()


def test_configurations_outside_the_two_call_path_run_entry_by_entry():
    cfg = SHAPES["kitti"]
    step, net, inputs = _setup(cfg)
    step.wrapped.renderer.lean_training_outputs = False
    torch.manual_seed(3)
    loss, _, data = step(*inputs, ids_encoder=[0], ids_render=cfg["ids_render"], ids_loss=cfg["ids_loss"])
    assert step.last_path.startswith("entries: the renderer is not in training mode with lean") and "weights" in data["coarse"][0]
    loss.backward()
    step.wrapped.renderer.lean_training_outputs = True
    step.criterion.lambda_depth_reg = 0.1
    assert "regulariser" in step.why_not(inputs[0], [0], cfg["ids_render"], cfg["ids_loss"])
    step.criterion.lambda_depth_reg = 0.0
    assert step.why_not(inputs[0], [0], cfg["ids_render"], cfg["ids_loss"]) is None
    # MLP-predicted colours (sample_color=False: a four-output MLP served by torch_modes.py) never reach the one-output kernels
    net.sample_color = False
    assert "sample_color=False" in step.why_not(inputs[0], [0], cfg["ids_render"], cfg["ids_loss"])
    net.sample_color = True


def test_fused_eval_frame_equals_the_entry_by_entry_frame():
    """bts_eval_frame (ABI 7: the evaluator's forward after the CNN in one call) against encode -> ImageRaySampler.sample -> renderer ->
    reconstruct -> distance_to_z: the same kernels, the same jitter stream -- every output bit for bit; nv = 1 with the encoder frame as
    the render view (eval_depth.yaml) and nv = 2 other frames."""
    import behindthescenes_amd as bts
    from behindthescenes_amd import synthetic as S
    dev = torch.device("cuda")
    for ids_render, learn_empty in (([0], True), ([1, 2], False)):
        scene = S.synthetic_scene(2, 3, 48, 160, 64, seed=9, intrinsics=S.K_KITTIRAW, smooth=True)
        torch.manual_seed(4)
        net = bts.BTSNet(S.field_conf(64, 64, 0, 48, 160, learn_empty=learn_empty))
        net.encoder = bts.FeatureMapEncoder((48, 160), 64, num_views=2)
        S.init_mlp_(net.mlp_coarse, seed=7)
        net = net.to(dev).eval()
        wrapped = bts.NeRFRenderer.from_conf(dict(n_coarse=64, lindisp=True, hard_alpha_cap=True)).bind_parallel(net).eval().to(dev)
        frame = bts.FusedEvalFrame(wrapped, bts.ImageRaySampler(3.0, 80.0))
        inputs = [scene[k].to(dev) for k in ("images", "projs", "poses")]
        outs = []
        for fused in (False, True):
            frame.fused = fused
            torch.manual_seed(21)
            outs.append(frame(*inputs, ids_encoder=[0], ids_render=ids_render))
            assert frame.last_path == ("fused" if fused else "entries: switched off (fused=False)")
        a, b = outs
        assert torch.equal(a["rays"], b["rays"]) and torch.equal(a["rgb_gt"].contiguous(), b["rgb_gt"].contiguous())
        for k in ("rgb", "depth", "invalid", "weights", "alphas"):
            assert a["coarse"][0][k].shape == b["coarse"][0][k].shape, k
            assert torch.equal(a["coarse"][0][k], b["coarse"][0][k]), (ids_render, k)


def test_fused_eval_frame_reads_a_channels_last_map_as_it_is_and_leaves_no_stale_state():
    """ABI 9 (BtsEvalFrame.feat_channels_last): an encoder whose scale-0 map is in torch's channels_last format -- what the shipped
    Monodepth2 decoder writes -- goes through bts_eval_frame without a layout copy; the projected map G differs from the NCHW route's by
    fp32 rounding (another summation order, tests/test_gpu_channels_last.py), the frame's outputs accordingly.  And a fused frame
    leaves no field state behind: a field query afterwards raises instead of running on the previous encode's maps (round-5 advice)."""
    import behindthescenes_amd as bts
    from behindthescenes_amd import native, synthetic as S
    dev = torch.device("cuda")
    scene = S.synthetic_scene(1, 2, 48, 160, 64, seed=9, intrinsics=S.K_KITTIRAW, smooth=True)
    torch.manual_seed(4)
    net = bts.BTSNet(S.field_conf(64, 64, 0, 48, 160))
    net.encoder = bts.FeatureMapEncoder((48, 160), 64, num_views=1)
    S.init_mlp_(net.mlp_coarse, seed=7)
    net = net.to(dev).eval()
    wrapped = bts.NeRFRenderer.from_conf(dict(n_coarse=64, lindisp=True, hard_alpha_cap=True)).bind_parallel(net).eval().to(dev)
    frame = bts.FusedEvalFrame(wrapped, bts.ImageRaySampler(3.0, 80.0))
    inputs = [scene[k].to(dev) for k in ("images", "projs", "poses")]
    jit = torch.rand(2 * 48 * 160, 64, device=dev)
    a = frame(*inputs, ids_encoder=[0], ids_render=[0], jitter=jit)

    class ChannelsLast(torch.nn.Module):          # the same maps, channels_last in memory
        def __init__(self, inner):
            super().__init__()
            self.inner, self.latent_size, self.scales = inner, inner.latent_size, inner.scales

        def forward(self, x):
            return [m.contiguous(memory_format=torch.channels_last) for m in self.inner(x)]
    net.encoder = ChannelsLast(net.encoder)
    seen = {}
    orig = native.eval_frame

    def spy(fr, stream):
        seen["cl"] = int(fr.feat_channels_last)
        return orig(fr, stream)
    native.eval_frame = spy
    try:
        b = frame(*inputs, ids_encoder=[0], ids_render=[0], jitter=jit)
    finally:
        native.eval_frame = orig
    assert frame.last_path == "fused" and seen["cl"] == 1
    ca, cb = a["coarse"][0], b["coarse"][0]
    assert torch.equal(ca["invalid"], cb["invalid"])
    assert ((ca["depth"] - cb["depth"]).abs() / cb["depth"].abs()).max().item() <= 1e-5
    for k in ("rgb", "weights", "alphas"):
        assert (ca[k] - cb[k]).abs().max().item() <= 1e-5, (k, (ca[k] - cb[k]).abs().max().item())
    with pytest.raises(native.BtsNativeError, match="encode"):
        net(torch.zeros(1, 8, 3, device=dev))
    net.encode(*inputs, ids_encoder=[0], ids_render=[0])
    net(torch.zeros(1, 8, 3, device=dev))
