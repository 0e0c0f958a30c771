"""``FusedTrainStep`` -- the renderer's share of a training iteration in TWO calls into the library.

The reference's training forward (``BTSWrapper.forward``, models/bts/trainer.py:208-259) is, after the CNN: ``net.encode`` ->
``sampler.sample`` -> one ``renderer(...)`` per scale -> ``sampler.reconstruct`` -> (utils/base_trainer.py:287-297) the criterion;
autograd then walks the chain back.  Entry by entry that is ~20 calls into libbts_render.so per step, each with its allocations,
autograd node and ctypes marshalling, plus ~25 small torch kernels of glue: at exp_kitti_raw.yaml's shapes the host needs 1.1 ms to
issue what the GPU runs in 0.5 ms.  This module makes the same sequence ONE autograd node around ``bts_train_step_fwd`` /
``bts_train_step_bwd`` (include/bts_render.h, ABI 7): the same kernels with the same arguments in the same order -- the forward is
bit-identical to the entry-by-entry path, the backward equal up to the order of its float atomics (tests/test_gpu_train_fused.py).

Everything random stays where the reference draws it: the flip decision and the patch coordinates on the CPU generator in the
reference's order (models_bts.py:99, ray_sampler.py:141-143), the stratified jitter of every render with ``torch.rand`` on the device
generator (nerf.py:112).  A configuration the two-call path does not cover (a fine pass, a sampling schedule, density noise, one of the
optional regularisers, per-sample outputs requested by the caller) runs the entry-by-entry sequence instead -- the same kernels through
``BTSNet.encode`` / ``NeRFRenderer.forward`` / ``ReconstructionLoss``; ``FusedTrainStep.why_not`` says which condition failed."""
import ctypes as C
import weakref

import torch
import torch.nn.functional as F

from . import _lib, native
from .loss import LazyScalars, ReconstructionLoss
from .ray_sampler import PatchRaySampler
from .renderer import NeRFRenderer


def _r64(x):
    return (x + 63) // 64 * 64


class _Arena:
    """The step's device scratch for one shape: state the forward leaves for the backward (projected maps, saved activations, loss
    gradients), the kept all-zero (d_proj, tile flags) pairs, cameras, packed frames, the backward workspace.  Allocated once per
    (shape, device, stream) and reused by every step -- pointers that do not change from step to step."""

    def __init__(self, key, dev, n, nv, H, W, Hd, Bp, K, P, shifts, spec, ws_bytes):
        f32 = dict(device=dev, dtype=torch.float32)
        B = n * Bp
        self.key = key
        self.busy = None              # weakref to the step that owns the saved state between its forward and its backward
        self.cams = torch.empty(n * (25 + nv * 25), **f32)
        self.imgs = torch.empty((n, nv, H, W, 4), **f32)
        self.ws = torch.empty(ws_bytes // 4 + 4, **f32)
        self.d_empty_proj = torch.empty(Hd, **f32)
        # the step's parameter block: everything that depends on the shape and the configuration only is written once (st_key says for
        # which configuration), a step then sets the handful of pointers that move
        self.st, self.st_key, self.st_keep = _lib.BtsTrainStep(), None, None
        # patch coordinates: a small ring of (pinned host, device) int32 blocks (3, n, P).  The draws go straight into the pinned block,
        # one asynchronous copy follows; an event per slot keeps the host from overwriting a block whose copy has not run yet
        self.idx_ring = [dict(pin=torch.empty((3, n, P), dtype=torch.int32).pin_memory(), dev=torch.empty((3, n, P), device=dev, dtype=torch.int32),
                              ev=None) for _ in range(4)]
        for sl in self.idx_ring:
            sl["rows"] = [sl["pin"][j, i] for i in range(n) for j in range(3)]
        self.idx_next = 0
        self.scales = []
        for sh in shifts:
            h, w = H >> sh, W >> sh
            tiles = native.proj_tile_count(spec, h, w)
            s = dict(proj=torch.empty((n, h, w, Hd), **f32), tiles=torch.empty((n, tiles), device=dev, dtype=torch.uint8),
                     z=torch.empty((B, K), **f32), sigma_raw=torch.empty((B, K), **f32), trans=torch.empty((B, K), **f32),
                     rgb_samps=torch.empty((B, K, nv * 3), **f32), parts=torch.empty((n * P, 4), **f32),
                     g_rgb=torch.empty((B, nv * 3), **f32), g_depth=torch.empty(B, **f32), gs_rgb=torch.empty((B, nv * 3), **f32),
                     gs_depth=torch.empty(B, **f32),
                     # the kept pair of ABI 6: all zero between steps (the projection backward returns it to zero)
                     d_proj=torch.zeros((n, h, w, Hd), **f32), d_tiles=torch.zeros((n, tiles), device=dev, dtype=torch.uint8))
            self.scales.append(s)


    def upload_patches(self, draw):
        """``draw(rows)`` fills the next pinned block (PatchRaySampler.draw_patches); -> the device block (3, n, P) int32."""
        sl = self.idx_ring[self.idx_next]
        self.idx_next = (self.idx_next + 1) % len(self.idx_ring)
        if sl["ev"] is not None and not sl["ev"].query():
            sl["ev"].synchronize()
        draw(sl["rows"])
        sl["dev"].copy_(sl["pin"], non_blocking=True)
        if sl["ev"] is None:
            sl["ev"] = torch.cuda.Event()
        sl["ev"].record()
        return sl["dev"]


_ARENAS = {}


def release_arenas():
    """Drops the kept scratch of every shape (e.g. when a process switches from training to evaluation)."""
    _ARENAS.clear()


def _arena(key, make):
    """A free arena of this shape: normally the one and only; a second one is made when a step's forward runs while an earlier step of
    the same shape still waits for its backward (gradient accumulation over micro-batches whose backwards come later)."""
    pool = _ARENAS.setdefault(key, [])
    for a in pool:
        if a.busy is None or a.busy() is None:
            return a
    a = make()
    pool.append(a)
    return a


class _Token:
    __slots__ = ("__weakref__",)


class _TrainStepFn(torch.autograd.Function):
    """(packed MLP parameters, empty feature | None, one feature map per scale) -> (loss, the logging vector).  Differentiable output:
    the loss.  ``job`` carries everything else (the filled BtsTrainStep, the arena, the tensors its pointers name)."""

    @staticmethod
    def forward(ctx, mlp_params, empty_feature, job, *feats):
        ctx.set_materialize_grads(False)
        st, arena = job.st, job.arena
        st.mlp_params = mlp_params.data_ptr()
        st.empty_feature = None if empty_feature is None else empty_feature.data_ptr()
        for s, f in enumerate(feats):
            st.scale[s].feat_nchw = f.data_ptr()
            st.scale[s].feat_channels_last = int(native.is_channels_last(f))
        native.train_step_fwd(st, native._stream(mlp_params))
        vals = job.vals
        ctx.n_feats = len(feats)
        # (needs_input_grad reflects requires_grad even under torch.no_grad(), and inside forward() grad mode is always off: the caller's
        # mode travels in the job)
        ctx.job = job if (job.grad_mode and any(ctx.needs_input_grad)) else None
        if ctx.job is not None:
            ctx.save_for_backward(mlp_params, *([empty_feature] if empty_feature is not None else []), *feats)
            ctx.has_empty = empty_feature is not None
            job.token = _Token()
            arena.busy = weakref.ref(job.token)
        else:
            arena.busy = None
        ctx.mark_non_differentiable(vals)
        return vals[8], vals

    @staticmethod
    def backward(ctx, g_loss, _g_vals):
        job = ctx.job
        if g_loss is None or job is None:
            return (None,) * (3 + ctx.n_feats)
        saved = ctx.saved_tensors
        mlp_params = saved[0]
        feats = saved[2:] if ctx.has_empty else saved[1:]
        st, arena = job.st, job.arena
        if arena.busy is None or arena.busy() is not job.token:
            raise native.BtsNativeError("FusedTrainStep: the saved state of this step was released (a second backward through the same step? "
                                        "call backward once per forward, or use retain_graph with the entry-by-entry path)")
        need_mlp, need_empty = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_f = ctx.needs_input_grad[3:]
        dev = mlp_params.device
        d_mlp = torch.empty_like(mlp_params) if (need_mlp or (need_empty and ctx.has_empty)) else None
        d_empty = torch.empty(job.C, device=dev, dtype=torch.float32) if (need_empty and ctx.has_empty) else None
        d_feats = [torch.empty_like(f) if nf else None for f, nf in zip(feats, need_f)]
        st.mlp_params = mlp_params.data_ptr()
        st.d_mlp_params = None if d_mlp is None else d_mlp.data_ptr()
        st.d_empty_feature = None if d_empty is None else d_empty.data_ptr()
        st.d_empty_proj = arena.d_empty_proj.data_ptr() if ctx.has_empty else None
        for s, (f, d) in enumerate(zip(feats, d_feats)):
            st.scale[s].feat_nchw = f.data_ptr()
            st.scale[s].feat_channels_last = int(native.is_channels_last(f))          # (empty_like keeps the format: d has f's layout)
            st.scale[s].d_feat_nchw = None if d is None else d.data_ptr()
        g = g_loss.detach().float().contiguous()
        try:
            native.train_step_bwd(st, g, native._stream(mlp_params))
        except Exception:
            for sc in arena.scales:       # a failed launch may have left the kept pairs dirty: they start over
                sc["d_proj"].zero_(), sc["d_tiles"].zero_()
            raise
        finally:
            arena.busy = None
            job.token = None
        return (d_mlp if need_mlp else None, d_empty, None) + tuple(d_feats)


class _Job:
    __slots__ = ("st", "arena", "vals", "keep", "grad_mode", "token", "C")


class FusedTrainStep(torch.nn.Module):
    """``loss, loss_dict, data = step(images, projs, poses, ids_encoder, ids_render, ids_loss)``

    wrapped    ``NeRFRenderer.bind_parallel(net)`` (the reference's ``self.renderer``: ``.net`` + ``.renderer``)
    sampler    the training ``PatchRaySampler``
    criterion  ``behindthescenes_amd.ReconstructionLoss``
    multiscale ``prediction_mode == "multiscale"`` (trainer.py:220-242: one render per encoder scale), else one render of the current scale

    ``data`` is the dict the reference's ``BTSWrapper.forward`` returns for a training step, in ``lean_training_outputs`` form:
    ``coarse`` / ``fine`` (aliases, trainer.py:247-248) per scale with ``rgb (n, P, ph, pw, nv, 3)``, ``depth (n, P, ph, pw)`` and the
    invalid-ray reductions, ``rgb_gt``, ``rays`` -- detached: the loss is already formed (and is what carries the graph)."""

    def __init__(self, wrapped, sampler, criterion, multiscale=False, fused=True, concurrent_scales=False):
        super().__init__()
        self.wrapped = wrapped
        self.sampler, self.criterion = sampler, criterion
        self.multiscale, self.fused = bool(multiscale), bool(fused)
        # a multiscale step's per-scale kernel chains on side queues of the library, forked from / joined into the current stream inside
        # the two calls (BtsTrainStep.concurrent_scales).  Measured and NOT the default: every kernel of a chain is a persistent grid
        # that fills the chip, so the queues take turns instead of overlapping -- exp_re10k.yaml 2.635 ms serial vs 2.655 ms side by
        # side, K = 128 5.32 vs 5.37 (profiles/r05i)
        self.concurrent_scales = bool(concurrent_scales)
        self.last_path = None        # "fused" | "entries: <reason>" of the most recent call

    # ---- which path ------------------------------------------------------------------------------------------------
    def why_not(self, images=None, ids_encoder=(0,), ids_render=(), ids_loss=()):
        """None when the two-call path covers this configuration, else the reason it does not."""
        net, r, crit, smp = self.wrapped.net, self.wrapped.renderer, self.criterion, self.sampler
        if not self.fused:
            return "switched off (fused=False)"
        if not isinstance(r, NeRFRenderer) or not isinstance(smp, PatchRaySampler) or not isinstance(crit, ReconstructionLoss):
            return "needs behindthescenes_amd's NeRFRenderer, PatchRaySampler and ReconstructionLoss"
        if getattr(self.wrapped, "simple_output", False):
            return "simple_output"
        if not (r.training and r.lean_training_outputs):
            return "the renderer is not in training mode with lean_training_outputs"
        if r.using_fine or r.n_fine or r.sched is not None or r.noise_std > 0.0 or r.white_bkgd:
            return "fine pass / sampling schedule / density noise / white background"
        if getattr(r.sample_coarse, "__func__", None) is not NeRFRenderer.sample_coarse:
            return "sample_coarse is overridden"
        if (crit.lambda_depth_reg > 0 or crit.lambda_alpha_reg > 0 or crit.lambda_surfaceness_reg > 0 or crit.lambda_depth_smoothness > 0
                or crit.lambda_entropy > 0):
            return "an optional regulariser of the loss is on"
        if crit.invalid_policy not in native.INVALID_POLICIES:
            return f"invalid_policy {crit.invalid_policy}"
        if smp.channels != 3 or smp.patch_size_x * smp.patch_size_y > 64:
            return "patches of more than 64 pixels / other than 3 colour channels"
        if net.mlp_fine is not None or not net.native_scale_maps:
            return "a separate fine MLP / resized scale maps"
        if (not net.sample_color) or net._d_out != 1:
            # the field's MLP is four outputs wide and served by torch_modes.py: the fused kernels know the one-output
            # density layout only (net._combined is encode's state; the two-call path has exactly one encoder view, checked next)
            return "sample_color=False (MLP-predicted colours run as a PyTorch composition, torch_modes.py)"
        if len(ids_encoder) != 1:
            return "more than one encoder view"
        if not (1 <= len(ids_render) <= _lib.BTS_MAX_VIEWS) or not (1 <= len(ids_loss) <= _lib.BTS_MAX_LOSS_VIEWS):
            return f"{len(ids_render)} render / {len(ids_loss)} loss views"
        n_scales = len(net.encoder.scales) if self.multiscale else 1
        if n_scales > _lib.BTS_MAX_SCALES:
            return f"{n_scales} scales"
        if images is not None and (not images.is_cuda or images.dtype != torch.float32 or images.shape[2] != 3):
            return "frames must be float32 (n, v, 3, H, W) on the GPU"
        return None

    # ---- the entry-by-entry sequence (trainer.py:208-259 + the criterion): the same kernels, one library call each ------------------
    def _entries(self, images, projs, poses, ids_encoder, ids_render, ids_loss):
        net, smp = self.wrapped.net, self.sampler
        images_ip = images * .5 + .5
        net.encode(images, projs, poses, ids_encoder=ids_encoder, ids_render=ids_render, images_alt=images_ip)
        all_rays, all_rgb_gt = smp.sample(native_take(images_ip, ids_loss), native_take(poses, ids_loss), native_take(projs, ids_loss))
        data = dict(coarse=[], fine=[])
        scales = list(net.encoder.scales) if self.multiscale else [net.get_scale()]
        back = net.get_scale()
        for scale in scales:
            net.set_scale(scale)
            rd = self.wrapped(all_rays, want_weights=True, want_alphas=True, want_rgb_samps=True)
            if "fine" not in rd:
                rd["fine"] = dict(rd["coarse"])
            rd["rgb_gt"], rd["rays"] = all_rgb_gt, all_rays
            rd = smp.reconstruct(rd)
            data["coarse"].append(rd["coarse"]), data["fine"].append(rd["fine"])
            data["rgb_gt"], data["rays"] = rd["rgb_gt"], rd["rays"]
        net.set_scale(back)
        loss, loss_dict = self.criterion(data)
        return loss, loss_dict, data

    # ---- the two-call path -------------------------------------------------------------------------------------------------------
    def forward(self, images, projs, poses, ids_encoder=(0,), ids_render=None, ids_loss=None, patches=None, jitter=None):
        """``patches`` = (frame, y, x) index tensors and ``jitter`` = one (n * rays, K) tensor of U[0, 1) per rendered scale replace the step's
        own draws (the deterministic sub-seam of SURVEY 8b: the reference's seeded draws are injected, tests/test_gpu_fused_anchor.py)."""
        def ids(x, default):
            if x is None:
                return default
            return [int(i) for i in (x.tolist() if torch.is_tensor(x) else x)]
        v = images.shape[1]
        ids_encoder, ids_render, ids_loss = ids(ids_encoder, [0]), ids(ids_render, list(range(v))), ids(ids_loss, list(range(v)))
        reason = self.why_not(images, ids_encoder, ids_render, ids_loss)
        if reason is not None:
            self.last_path = "entries: " + reason
            return self._entries(images, projs, poses, ids_encoder, ids_render, ids_loss)
        self.last_path = "fused"
        net, r, crit, smp = self.wrapped.net, self.wrapped.renderer, self.criterion, self.sampler
        n, v, _, H, W = images.shape
        dev = images.device
        images = images.contiguous()
        projs, poses = projs.detach().float().contiguous(), poses.detach().float().contiguous()
        id_enc = ids_encoder[0]
        # ---- the CNN (PyTorch): models_bts.py:99-110, flip augmentation included (its draw comes first, as in the reference)
        net.mlp_coarse.invalidate_packed()
        net.invalidate_field_state()                          # (the two calls bypass encode: nothing it cached describes this batch)
        enc_in = images[:, id_enc]
        do_flip = bool(net.flip_augmentation and net.training and (torch.rand(1) > .5).item())
        if do_flip:
            enc_in = torch.flip(enc_in, dims=(-1,))
        latents = net.encoder(enc_in)
        if do_flip:
            latents = [torch.flip(il, dims=(-1,)) for il in latents]
        size0 = tuple(latents[0].shape[-2:])
        if size0 != (H, W):
            raise native.BtsNativeError(f"the encoder's scale-0 map is {size0}, the frames are {(H, W)}: the renderer samples both on one grid")
        scales = list(net.encoder.scales) if self.multiscale else [net.get_scale()]
        feats, shifts = [], []
        for s in scales:
            il = latents[s]
            sh = net._scale_shift(tuple(il.shape[-2:]), size0)
            if sh is None:                                    # a size that is not H / 2^s: the reference's nearest resize (models_bts.py:115-117)
                il, sh = F.interpolate(il, size0), 0
            feats.append(native.as_feature_map(il)), shifts.append(sh)                 # (NCHW or channels_last, as the encoder wrote it)
        # ---- buffers: the arena (kept per shape) and one fresh block for what the caller gets to keep
        P, ph, pw = smp._patch_count, smp.patch_size_y, smp.patch_size_x
        Bp, K = P * ph * pw, int(r.n_coarse)
        B = n * Bp
        spec, nv, S = net.spec, len(ids_render), len(scales)
        stream = torch.cuda.current_stream(dev).cuda_stream
        key = (dev, stream, n, v, nv, H, W, Bp, K, P, ph, pw, tuple(shifts), spec)

        def make():
            cfg0 = native._spec_cfg(spec, n, H, W, nv, 0, -1)
            args = _lib.BtsRenderArgs(rays_per_sample=Bp, K=K)
            ws = int(_lib.load().bts_render_bwd_workspace(C.byref(cfg0), C.byref(args)))
            # one slice per scale: with `concurrent_scales` the scales' backward chains run side by side (BtsTrainStep.concurrent_scales)
            return _Arena(key, dev, n, nv, H, W, spec.d_hidden, Bp, K, P, shifts, spec, ((ws + 255) // 256 * 256) * len(shifts))
        arena = _arena(key, make)
        # ---- the step's draws: patches on the CPU generator in the reference's order (straight into pinned memory, one asynchronous
        # copy), one jitter tensor per render on the device generator
        if patches is None:
            idx = arena.upload_patches(lambda rows: smp.draw_patches(n, len(ids_loss), H, W, rows=rows))
        else:
            pv, py, px = patches
            idx = torch.stack((pv, py, px)).to(torch.int32).pin_memory().to(dev, non_blocking=True)
        if jitter is None:
            jit = [torch.rand((B, K), device=dev, dtype=torch.float32) for _ in scales]
        else:
            jit = [j.to(dev, torch.float32).contiguous() for j in ([jitter] if torch.is_tensor(jitter) else list(jitter))]
            if len(jit) != len(scales) or any(tuple(j.shape) != (B, K) for j in jit):
                raise native.BtsNativeError(f"jitter: {len(scales)} tensor(s) of shape {(B, K)} expected")
        per_scale = _r64(B * nv * 3) + _r64(B) + 2 * _r64(B * nv)
        out = torch.empty(_r64(B * 8) + _r64(B * 3) + 64 + S * per_scale, device=dev, dtype=torch.float32)
        o = [0]

        def take(count, *shape):
            t = out[o[0]:o[0] + count].view(*shape)
            o[0] += _r64(count)
            return t
        rays, rgb_gt, vals = take(B * 8, n, Bp, 8), take(B * 3, n, Bp, 3), take(9, 9)
        levels = [dict(rgb=take(B * nv * 3, n, Bp, nv * 3), depth=take(B, n, Bp), invalid_wsum=take(B * nv, n, Bp, nv),
                       invalid_any=take(B * nv, n, Bp, nv)) for _ in scales]
        # ---- the struct: what depends on the configuration only is (re)written when that changes ...
        st = arena.st
        M = crit.loss_matrix(S, (B,) * S, (True,) * S)            # trainer.py:247-248: fine = dict(coarse) on every scale
        ck = (tuple(ids_render), tuple(ids_loss), id_enc, bool(r.lindisp), bool(r.hard_alpha_cap), crit.invalid_policy,
              crit.lambda_edge_aware_smoothness > 0, self.concurrent_scales, float(smp.z_near), float(smp.z_far), id(M))
        if arena.st_key != ck:
            st.cfg = native._spec_cfg(spec, n, H, W, nv, 0, ids_render.index(id_enc) if id_enc in ids_render else -1)
            st.v, st.id_encoder, st.n_loss = v, id_enc, len(ids_loss)
            for j, i in enumerate(ids_render):
                st.ids_render[j] = i
            for j, i in enumerate(ids_loss):
                st.ids_loss[j] = i
            st.P, st.ph, st.pw, st.K = P, ph, pw, K
            st.lindisp, st.hard_alpha_cap = int(bool(r.lindisp)), int(bool(r.hard_alpha_cap))
            st.invalid_policy = native.INVALID_POLICIES[crit.invalid_policy]
            st.edge_aware_smoothness = int(crit.lambda_edge_aware_smoothness > 0)
            st.n_scales = S
            st.concurrent_scales = int(self.concurrent_scales and S > 1)
            st.z_near, st.z_far, st.img_scale, st.img_shift = float(smp.z_near), float(smp.z_far), 0.5, 0.5
            for i, x in enumerate(M.reshape(-1).tolist()):
                st.loss_matrix[i] = x
            st.cams, st.imgs_nhwc4 = arena.cams.data_ptr(), arena.imgs.data_ptr()
            st.bwd_workspace, st.bwd_workspace_bytes = arena.ws.data_ptr(), arena.ws.numel() * 4
            for s_ in range(S):
                q, a = st.scale[s_], arena.scales[s_]
                q.feat_shift = shifts[s_]
                q.proj_nhwc, q.sampled_tiles = a["proj"].data_ptr(), a["tiles"].data_ptr()
                q.z_samp, q.sigma_raw, q.trans, q.rgb_samps = a["z"].data_ptr(), a["sigma_raw"].data_ptr(), a["trans"].data_ptr(), a["rgb_samps"].data_ptr()
                q.loss_parts, q.g_rgb, q.g_depth = a["parts"].data_ptr(), a["g_rgb"].data_ptr(), a["g_depth"].data_ptr()
                q.gs_rgb, q.gs_depth = a["gs_rgb"].data_ptr(), a["gs_depth"].data_ptr()
                q.d_proj_nhwc, q.d_proj_tiles = a["d_proj"].data_ptr(), a["d_tiles"].data_ptr()
            arena.st_key, arena.st_keep = ck, M                   # (M is kept so that id(M) cannot be recycled while the key names it)
        # ... and what moves from step to step
        st.images, st.Ks, st.poses_c2w = images.data_ptr(), projs.data_ptr(), poses.data_ptr()
        st.patch_v, st.patch_y, st.patch_x = idx[0].data_ptr(), idx[1].data_ptr(), idx[2].data_ptr()
        st.rays, st.rgb_gt, st.loss_vals = rays.data_ptr(), rgb_gt.data_ptr(), vals.data_ptr()
        for s_ in range(S):
            q, lv = st.scale[s_], levels[s_]
            q.jitter = jit[s_].data_ptr()
            q.rgb, q.depth = lv["rgb"].data_ptr(), lv["depth"].data_ptr()
            q.invalid_wsum, q.invalid_any = lv["invalid_wsum"].data_ptr(), lv["invalid_any"].data_ptr()
        job = _Job()
        job.st, job.arena, job.vals, job.grad_mode, job.token, job.C = st, arena, vals, torch.is_grad_enabled(), None, spec.C
        job.keep = (images, projs, poses, idx, jit, out)         # what the struct's pointers name, for as long as the graph lives
        empty = net.empty_feature if net.learn_empty else None
        packed = net.mlp_coarse.packed()
        if packed.numel() != spec.mlp_param_count():            # the library sees a pointer only: a mis-sized vector would be read past its end
            raise native.BtsNativeError(f"the packed MLP holds {packed.numel()} values, the field's layout {spec.mlp_param_count()}")
        loss, vals_out = _TrainStepFn.apply(packed, empty, job, *feats)
        loss_dict = LazyScalars(crit._KEYS, vals_out)
        # ---- the reference's data dict (reconstruct's views)
        data = dict(coarse=[], fine=[])
        for lv in levels:
            part = dict(rgb=lv["rgb"].view(n, P, ph, pw, nv, 3), depth=lv["depth"].view(n, P, ph, pw),
                        invalid_wsum=lv["invalid_wsum"].view(n, P, ph, pw, nv), invalid_any=lv["invalid_any"].view(n, P, ph, pw, nv))
            data["coarse"].append(part), data["fine"].append(dict(part))
        data["rgb_gt"], data["rays"] = rgb_gt.view(n, P, ph, pw, 3), rays
        return loss, loss_dict, data


def native_take(x, ids):
    from .field import _take
    return _take(x, ids)


class FusedEvalFrame(torch.nn.Module):
    """``data = frame(images, projs, poses, ids_encoder=[0], ids_render=[0])`` -- the evaluator's forward after the data loader
    (``BTSWrapper.forward``, models/bts/evaluator.py:60-79) in ONE library call (``bts_eval_frame``, ABI 7): encode's hand-over, the rays
    of every pixel of every frame (``ImageRaySampler.sample``), the render with ``sample_coarse`` inside, ``distance_to_z``.  The same
    kernels with the same arguments as the entry-by-entry sequence (bit-identical outputs, tests/test_gpu_train_fused.py); what goes away
    is the host work between them -- 8 % of a 1.1 ms frame.  The jitter is the caller's ``torch.rand`` draw, as in the reference.

    ``data``: ``coarse`` / ``fine`` (aliases, evaluator.py:70-71) with ``rgb (n, v, H, W, nv, 3)``, ``depth (n, v, H, W)`` (z-depth when
    ``to_z``, evaluator.py:78-79), ``invalid (n, v, H, W, K, nv)``, ``weights`` / ``alphas (n, v, H, W, K)``; ``rgb_gt (n, v, H, W, 3)``;
    ``rays (n, v*H*W, 8)``.  A configuration outside the call's envelope runs the entry-by-entry sequence (``why_not``)."""

    def __init__(self, wrapped, sampler, fused=True):
        super().__init__()
        self.wrapped, self.sampler, self.fused = wrapped, sampler, bool(fused)
        self.last_path = None
        self._scratch = {}

    def why_not(self, images=None, ids_encoder=(0,), ids_render=(0,)):
        from .ray_sampler import ImageRaySampler
        net, r = self.wrapped.net, self.wrapped.renderer
        if not self.fused:
            return "switched off (fused=False)"
        if not isinstance(r, NeRFRenderer) or not isinstance(self.sampler, ImageRaySampler) or getattr(self.wrapped, "simple_output", False):
            return "needs behindthescenes_amd's NeRFRenderer and ImageRaySampler"
        if r.using_fine or r.n_fine or r.sched is not None or r.white_bkgd or (r.training and r.noise_std > 0.0):
            return "fine pass / sampling schedule / white background / density noise"
        if getattr(r.sample_coarse, "__func__", None) is not NeRFRenderer.sample_coarse:
            return "sample_coarse is overridden"
        if net.torch_mode or net.mlp_fine is not None or net.get_scale() != 0 or (net.flip_augmentation and net.training):
            return "a PyTorch-composed field mode / a separate fine MLP / a scale other than 0 / flip augmentation"
        if len(ids_encoder) != 1 or len(ids_render) > _lib.BTS_MAX_VIEWS or self.sampler.channels != 3:
            return "more than one encoder view / more than 8 render views"
        if images is not None and (not images.is_cuda or images.dtype != torch.float32 or images.shape[2] != 3 or
                                   (self.sampler.height is not None and (self.sampler.height, self.sampler.width) != tuple(images.shape[-2:]))):
            return "frames must be float32 (n, v, 3, H, W) on the GPU at the sampler's size"
        return None

    def _entries(self, images, projs, poses, ids_encoder, ids_render, want_weights, want_alphas, to_z):
        from .projection import distance_to_z
        net, smp = self.wrapped.net, self.sampler
        net.encode(images, projs, poses, ids_encoder=ids_encoder, ids_render=ids_render)
        all_rays, all_rgb_gt = smp.sample(images * .5 + .5, poses, projs)
        rd = self.wrapped(all_rays, want_weights=want_weights, want_alphas=want_alphas)
        if "fine" not in rd:
            rd["fine"] = dict(rd["coarse"])
        rd["rgb_gt"], rd["rays"] = all_rgb_gt, all_rays
        rd = smp.reconstruct(rd)
        if to_z:
            rd["coarse"]["depth"] = distance_to_z(rd["coarse"]["depth"], projs)
            rd["fine"]["depth"] = rd["coarse"]["depth"]
        return dict(coarse=[rd["coarse"]], fine=[rd["fine"]], rgb_gt=rd["rgb_gt"], rays=rd["rays"])

    @torch.no_grad()
    def forward(self, images, projs, poses, ids_encoder=(0,), ids_render=(0,), want_weights=True, want_alphas=True, to_z=True, jitter=None):
        """``jitter`` (n * v * H * W, K) of U[0, 1) replaces the frame's own ``torch.rand`` draw (fused path only: the deterministic sub-seam
        of SURVEY 8b, tests/test_gpu_fused_anchor.py)."""
        ids_encoder, ids_render = [int(i) for i in ids_encoder], [int(i) for i in ids_render]
        reason = self.why_not(images, ids_encoder, ids_render)
        if reason is None and jitter is not None and tuple(jitter.shape) != (images.shape[0] * images.shape[1] * images.shape[3] * images.shape[4],
                                                                             int(self.wrapped.renderer.n_coarse)):
            raise native.BtsNativeError("jitter: one (n * v * H * W, K) tensor expected")
        if reason is not None:
            if jitter is not None:
                raise native.BtsNativeError("jitter can only be injected into the fused frame (" + reason + ")")
            self.last_path = "entries: " + reason
            return self._entries(images, projs, poses, ids_encoder, ids_render, want_weights, want_alphas, to_z)
        self.last_path = "fused"
        net, r, smp = self.wrapped.net, self.wrapped.renderer, self.sampler
        n, v, _, H, W = images.shape
        dev = images.device
        images = images.contiguous()
        projs, poses = projs.detach().float().contiguous(), poses.detach().float().contiguous()
        id_enc, nv, K, spec = ids_encoder[0], len(ids_render), int(r.n_coarse), net.spec
        net.mlp_coarse.invalidate_packed()
        feat = net.encoder(images[:, id_enc])[0]
        if tuple(feat.shape[-2:]) != (H, W):
            raise native.BtsNativeError(f"the encoder's scale-0 map is {tuple(feat.shape[-2:])}, the frames are {(H, W)}")
        feat = native.as_feature_map(feat.detach())           # NCHW or channels-last (the shipped decoder's format), read as it is: ABI 9
        net.invalidate_field_state()                          # (this call bypasses encode: nothing it cached describes this frame)
        # the reference's order of draws: ImageRaySampler draws nothing, the renderer's jitter is one torch.rand (nerf.py:112)
        rgb_gt = (images * .5 + .5).permute(0, 1, 3, 4, 2)                     # (n, v, H, W, 3): ray_sampler.py:253-258 (a view, as there)
        B = n * v * H * W
        jitter = torch.rand((B, K), device=dev, dtype=torch.float32) if jitter is None else jitter.to(dev, torch.float32).contiguous()
        f32 = dict(device=dev, dtype=torch.float32)
        key = (dev, n, v, nv, H, W, spec.d_hidden)
        sc = self._scratch.get(key)
        if sc is None:
            sc = self._scratch[key] = dict(cams=torch.empty(n * (25 + nv * 25), **f32), imgs=torch.empty((n, max(nv, 1), H, W, 4), **f32),
                                           proj=torch.empty((n, H, W, spec.d_hidden), **f32), inv_K=torch.empty((n, v, 3, 3), **f32))
        out = dict(rays=torch.empty((n, v * H * W, 8), **f32), rgb=torch.empty((n, v, H, W, nv, 3), **f32), depth=torch.empty((n, v, H, W), **f32),
                   depth_z=torch.empty((n, v, H, W), **f32) if to_z else None, weights=torch.empty((n, v, H, W, K), **f32) if want_weights else None,
                   alphas=torch.empty((n, v, H, W, K), **f32) if want_alphas else None, invalid=torch.empty((n, v, H, W, K, nv), **f32))
        fr = _lib.BtsEvalFrame()
        fr.cfg = native._spec_cfg(spec, n, H, W, nv, 0, ids_render.index(id_enc) if id_enc in ids_render else -1)
        fr.v, fr.id_encoder = v, id_enc
        for j, i in enumerate(ids_render):
            fr.ids_render[j] = i
        fr.K, fr.lindisp, fr.hard_alpha_cap, fr.norm_dir = K, int(bool(r.lindisp)), int(bool(r.hard_alpha_cap)), int(bool(smp.norm_dir))
        fr.z_near, fr.z_far, fr.img_scale, fr.img_shift = float(smp.z_near), float(smp.z_far), 0.5, 0.5
        fr.feat_channels_last = int(native.is_channels_last(feat))
        params = net.mlp_coarse.packed().detach()
        empty = net.empty_feature.detach() if net.learn_empty else None

        def dp(t):
            return None if t is None else t.data_ptr()
        fr.images, fr.Ks, fr.poses_c2w, fr.feat_nchw, fr.mlp_params, fr.empty_feature, fr.jitter = (dp(images), dp(projs), dp(poses), dp(feat), dp(params),
                                                                                                   dp(empty), dp(jitter))
        fr.cams, fr.imgs_nhwc4, fr.proj_nhwc, fr.inv_K = dp(sc["cams"]), dp(sc["imgs"]), dp(sc["proj"]), dp(sc["inv_K"])
        for k_ in ("rays", "rgb", "depth", "depth_z", "weights", "alphas", "invalid"):
            setattr(fr, k_, dp(out[k_]))
        native.eval_frame(fr, native._stream(images))
        part = dict(rgb=out["rgb"], depth=out["depth_z"] if to_z else out["depth"], invalid=out["invalid"])
        if want_weights:
            part["weights"] = out["weights"]
        if want_alphas:
            part["alphas"] = out["alphas"]
        return dict(coarse=[part], fine=[dict(part)], rgb_gt=rgb_gt, rays=out["rays"])
