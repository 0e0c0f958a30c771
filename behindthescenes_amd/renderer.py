"""``NeRFRenderer`` / ``_RenderWrapper`` with the reference's constructor, buffers, call signature and output dict
(models/common/render/nerf.py:12-457).  ``composite`` -- the hot loop of the reference (chunked field queries + ~80 small
kernels per chunk) -- is ONE fused HIP kernel here (plus its backward); there is no chunking and no torch fallback: a model
that is not a ``behindthescenes_amd.BTSNet`` is rejected."""
import torch
from torch.autograd import profiler

from . import native
from .field import BTSNet


class _RenderWrapper(torch.nn.Module):
    def __init__(self, net, renderer, simple_output):
        super().__init__()
        self.net = net
        self.renderer = renderer
        self.simple_output = simple_output

    def forward(self, rays, want_weights=False, want_alphas=False, want_z_samps=False, want_rgb_samps=False,
                sample_from_dist=None):
        if rays.shape[0] == 0:
            return torch.zeros(0, 3, device=rays.device), torch.zeros(0, device=rays.device)
        outputs = self.renderer(self.net, rays, want_weights=want_weights and not self.simple_output,
                                want_alphas=want_alphas and not self.simple_output,
                                want_z_samps=want_z_samps and not self.simple_output,
                                want_rgb_samps=want_rgb_samps and not self.simple_output, sample_from_dist=sample_from_dist)
        if self.simple_output:
            part = outputs["fine"] if self.renderer.using_fine else outputs["coarse"]
            return part["rgb"], part["depth"]
        return outputs


class NeRFRenderer(torch.nn.Module):
    def __init__(self, n_coarse=128, n_fine=0, n_fine_depth=0, noise_std=0.0, depth_std=0.01, eval_batch_size=100000,
                 white_bkgd=False, lindisp=False, sched=None, hard_alpha_cap=False, lean_training_outputs=False):
        super().__init__()
        self.n_coarse, self.n_fine, self.n_fine_depth = n_coarse, n_fine, n_fine_depth
        self.noise_std, self.depth_std = noise_std, depth_std
        self.eval_batch_size = eval_batch_size      # kept for config compatibility; the fused kernel never chunks
        self.white_bkgd, self.lindisp = white_bkgd, lindisp
        self.using_fine = n_fine > 0
        self.sched = sched if sched is not None and len(sched) > 0 else None
        self.register_buffer("iter_idx", torch.tensor(0, dtype=torch.long), persistent=True)
        self.register_buffer("last_sched", torch.tensor(0, dtype=torch.long), persistent=True)
        self.hard_alpha_cap = hard_alpha_cap
        # SURVEY 8f.1 (not a reference key): in training mode return only what a training step consumes -- rgb, depth and the
        # per-ray reductions behindthescenes_amd.ReconstructionLoss builds its invalid-ray mask from -- see forward()
        self.lean_training_outputs = bool(lean_training_outputs)
        # the lean training render also projects only the tiles of the feature map its samples read (BTSNet.native_field(sampled=...));
        # an attribute, not a config key: switch it off for an A/B
        self.sparse_projection = True

    # ---- sampling (nerf.py:103-208) ---------------------------------------------------------------------------------
    def sample_coarse(self, rays, u=None):
        """rays (B, 8) -> z (B, Kc).  The jitter ``u`` ~ U[0,1) (B, Kc) is drawn with torch's generator on the rays' device
        when not given (the reference jitters even in eval, nerf.py:116)."""
        B = rays.shape[0]
        if u is None:
            u = torch.rand((B, self.n_coarse), device=rays.device, dtype=torch.float32)
        return native.sample_coarse(rays.float().contiguous(), u.contiguous(), self.lindisp)

    def sample_coarse_from_dist(self, rays, weights, z_samp):
        B = rays.shape[0]
        num_samples = self.n_coarse
        weights = weights.detach() + 1e-5
        pdf = weights / torch.sum(weights, -1, keepdim=True)
        cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
        u = torch.rand(B, num_samples, dtype=torch.float32, device=rays.device)
        ids = torch.clamp(torch.searchsorted(cdf, u, right=True) - 1, 0, num_samples - 1)
        interp = torch.rand_like(ids, dtype=torch.float32)
        if self.lindisp:
            z_samp = 1 / z_samp
        centers = .5 * (z_samp[:, 1:] + z_samp[:, :-1])
        borders = torch.cat((z_samp[:, :1], centers, z_samp[:, -1:]), dim=-1)
        left, right = torch.gather(borders, -1, ids), torch.gather(borders, -1, ids + 1)
        z_new = left * (1 - interp) + right * interp
        if self.lindisp:
            z_new = 1 / z_new
        assert not torch.any(torch.isnan(z_new))
        return z_new

    def sample_fine(self, rays, weights):
        B = rays.shape[0]
        weights = weights.detach() + 1e-5
        pdf = weights / torch.sum(weights, -1, keepdim=True)
        cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
        u = torch.rand(B, self.n_fine - self.n_fine_depth, dtype=torch.float32, device=rays.device)
        inds = torch.clamp_min(torch.searchsorted(cdf, u, right=True).float() - 1.0, 0.0)
        z_steps = (inds + torch.rand_like(inds)) / self.n_coarse
        near, far = rays[:, -2:-1], rays[:, -1:]
        if not self.lindisp:
            z = near * (1 - z_steps) + far * z_steps
        else:
            z = 1 / (1 / near * (1 - z_steps) + 1 / far * z_steps)
        assert not torch.any(torch.isnan(z))
        return z

    def sample_fine_depth(self, rays, depth):
        z = depth.unsqueeze(1).repeat((1, self.n_fine_depth))
        z = z + torch.randn_like(z) * self.depth_std
        z = torch.max(torch.min(z, rays[:, -1:]), rays[:, -2:-1])
        assert not torch.any(torch.isnan(z))
        return z

    # ---- the hot path (nerf.py:210-313) -------------------------------------------------------------------------------
    def composite(self, model, rays, z_samp, coarse=True, sb=0, want_weights=True, want_alphas=True, want_rgb_samps=True,
                  want_invalid=True, want_invalid_sums=False, sigma_noise=None):
        """rays (B, 8), z_samp (B, K) -> (weights, rgb, depth, alphas, invalid, z_samp, rgb_samps) like the reference.
        Entries that were not requested come back as ``None`` (the reference always materialises all of them).  With
        ``want_invalid_sums`` two more entries follow: (invalid_wsum, invalid_any), (B, nv) each."""
        with profiler.record_function("renderer_composite"):     # the reference's trace range (nerf.py:222); one fused kernel here
            return self._composite(model, rays, z_samp, coarse, sb, want_weights, want_alphas, want_rgb_samps, want_invalid, want_invalid_sums,
                                   sigma_noise)

    def _composite(self, model, rays, z_samp, coarse, sb, want_weights, want_alphas, want_rgb_samps, want_invalid, want_invalid_sums,
                   sigma_noise=None, jitter=None, want_z=False, sparse_proj=False):
        """z_samp None: ``sample_coarse`` runs INSIDE the render kernel from the jitter ``jitter`` (B, K) ~ U[0, 1) (BtsRenderArgs.jitter:
        the same routine, bit-identical depths, no z_samp round trip through HBM); entry 5 of the result is then the depth tensor only
        when ``want_z`` or when autograd needs it (else None)."""
        if not isinstance(model, BTSNet):
            raise native.BtsNativeError("composite() needs a behindthescenes_amd.BTSNet (the fused HIP kernel IS the field query)")
        rays = rays.float().contiguous()
        if z_samp is not None:
            z_samp, jitter = z_samp.float().contiguous(), None
        else:
            jitter = jitter.float().contiguous()
        if model.torch_mode:
            # MLP-predicted colours / merged encoder views run as a PyTorch composition (torch_modes.py) -- the field
            # query AND the compositing around it, differentiable through autograd; no shipped configuration comes here
            from . import torch_modes
            if z_samp is None:
                z_samp = native.sample_coarse(rays, jitter, self.lindisp) if rays.is_cuda else torch_modes.sample_coarse(rays, jitter, self.lindisp)
            comp = torch_modes.composite(self, model, rays, z_samp, coarse=coarse, sb=sb)
            if want_invalid_sums:
                raise native.BtsNativeError("lean_training_outputs is a feature of the fused kernels; switch it off for the PyTorch-composed modes")
            return comp
        # sparse_proj (the lean training path): the projected map is built for THIS render's samples only (BTSNet.native_field)
        ft = model.native_field(coarse, sampled=(rays, z_samp, jitter, bool(self.lindisp)) if sparse_proj else None)
        n = ft.n
        if sb > 0 and sb != n:
            raise native.BtsNativeError(f"super-batch {sb} does not match the encoded batch {n}")
        if sb <= 0 and n != 1:
            raise native.BtsNativeError("sb=0 (no super-batch) is only meaningful for an encoded batch of 1")
        shape = (z_samp if z_samp is not None else jitter).shape
        mlp_params = model.mlp(coarse).packed()      # models_bts.py:293-307: mlp_coarse, or mlp_fine when the fine pass has its own
        empty = model.empty_feature if model.learn_empty else None
        if sigma_noise is None and self.training and self.noise_std > 0.0:
            # nerf.py:279-280: sigmas + randn_like(sigmas) * noise_std in training mode (no shipped config turns it on); drawn here with
            # torch's generator, added inside the kernels (BtsRenderArgs.sigma_noise).  `sigma_noise` lets a caller inject the draw.
            sigma_noise = torch.randn(shape, device=rays.device, dtype=torch.float32) * self.noise_std
        rgb, depth, weights, alphas, invalid, rgb_samps, inv_wsum, inv_any, z_out = native.RenderFunction.apply(
            ft.proj_nhwc, mlp_params, empty, ft, rays, z_samp, bool(self.hard_alpha_cap), bool(self.white_bkgd),
            bool(want_weights), bool(want_alphas), bool(want_rgb_samps), torch.is_grad_enabled(), bool(want_invalid), bool(want_invalid_sums),
            None if sigma_noise is None else sigma_noise.float().contiguous(), jitter, bool(self.lindisp), bool(want_z))
        if z_samp is None:
            z_samp = z_out if z_out.numel() else None
        ret = (weights if want_weights else None, rgb, depth, alphas if want_alphas else None, invalid if want_invalid else None, z_samp,
               rgb_samps if want_rgb_samps else None)
        return ret + (inv_wsum, inv_any) if want_invalid_sums else ret

    def forward(self, model, rays, want_weights=False, want_alphas=False, want_z_samps=False, want_rgb_samps=False,
                sample_from_dist=None):
        """rays (SB, B', 8) -> {"coarse": {...}[, "fine": {...}]} (nerf.py:315-401)."""
        with profiler.record_function("renderer_forward"):       # nerf.py:328
            return self._forward(model, rays, want_weights, want_alphas, want_z_samps, want_rgb_samps, sample_from_dist)

    def _forward(self, model, rays, want_weights, want_alphas, want_z_samps, want_rgb_samps, sample_from_dist):
        if self.sched is not None and self.last_sched.item() > 0:
            self.n_coarse = self.sched[1][self.last_sched.item() - 1]
            self.n_fine = self.sched[2][self.last_sched.item() - 1]
        assert len(rays.shape) == 3
        sb = rays.shape[0]
        rays = rays.reshape(-1, 8)
        jitter = None
        if sample_from_dist is None:
            if isinstance(model, BTSNet) and getattr(self.sample_coarse, "__func__", None) is NeRFRenderer.sample_coarse:
                # nerf.py:103-123 inside the render kernel: only the jitter is drawn here (the reference's torch.rand_like, :112).
                # (a subclass or an instance that overrides sample_coarse -- the reference's extension point -- is honoured: its depths
                # are used as they come)
                z_coarse, jitter = None, torch.rand((rays.shape[0], self.n_coarse), device=rays.device, dtype=torch.float32)
            else:
                z_coarse = self.sample_coarse(rays)
        else:
            prop_weights, prop_z = sample_from_dist
            ns = prop_weights.shape[-1]
            z_coarse = self.sample_coarse_from_dist(rays, prop_weights.reshape(-1, ns), prop_z.reshape(-1, ns))
            z_coarse, _ = torch.sort(z_coarse, dim=-1)
        if self.lean_training_outputs and self.training and not self.using_fine and torch.is_grad_enabled() and not getattr(model, "torch_mode", False):
            # in a training step nothing downstream of the renderer reads the per-sample tensors except the loss'
            # invalid-ray mask, and that only through sum_k weights * invalid / any_k invalid per view -- the render kernel's
            # epilogue emits exactly those (8 B per ray and view), and weights / alphas / invalid are neither written nor returned
            # even when the (reference) trainer asks for them.  Opt-in (`lean_training_outputs`), training mode only.
            # (rgb_samps is still written, as the backward's saved state only: the stores are free in the latency-bound forward and
            # spare the backward one projection + four taps per view and sample; it is not returned)
            comp = self._composite(model, rays, z_coarse, True, sb, False, False, True, False, True, jitter=jitter,
                                   sparse_proj=self.sparse_projection)
            nv = comp[7].shape[-1]
            return dict(coarse=dict(rgb=comp[1].reshape(sb, -1, comp[1].shape[-1]), depth=comp[2].reshape(sb, -1),
                                    invalid_wsum=comp[7].reshape(sb, -1, nv), invalid_any=comp[8].reshape(sb, -1, nv)))
        need_w = want_weights or self.using_fine
        with profiler.record_function("renderer_composite"):
            comp = self._composite(model, rays, z_coarse, True, sb, need_w, want_alphas, want_rgb_samps, True, False, jitter=jitter,
                                   want_z=want_z_samps or self.using_fine)
        outputs = dict(coarse=self._format_outputs(comp, sb, want_weights, want_alphas, want_z_samps, want_rgb_samps))
        if self.using_fine:
            all_samps = [comp[5]]
            if self.n_fine - self.n_fine_depth > 0:
                all_samps.append(self.sample_fine(rays, comp[0].detach()))
            if self.n_fine_depth > 0:
                all_samps.append(self.sample_fine_depth(rays, comp[2]))
            z_comb, _ = torch.sort(torch.cat(all_samps, dim=-1), dim=-1)
            fine = self.composite(model, rays, z_comb, coarse=False, sb=sb, want_weights=want_weights, want_alphas=want_alphas,
                                  want_rgb_samps=want_rgb_samps)
            outputs["fine"] = self._format_outputs(fine, sb, want_weights, want_alphas, want_z_samps, want_rgb_samps)
        return outputs

    def _format_outputs(self, rendered, sb, want_weights=False, want_alphas=False, want_z_samps=False, want_rgb_samps=False):
        weights, rgb, depth, alphas, invalid, z_samps, rgb_samps = rendered
        K = invalid.shape[-2]
        if sb > 0:
            rgb = rgb.reshape(sb, -1, rgb.shape[-1])
            depth = depth.reshape(sb, -1)
            invalid = invalid.reshape(sb, -1, K, invalid.shape[-1])
        ret = dict(rgb=rgb, depth=depth, invalid=invalid)
        if want_weights:
            ret["weights"] = weights.reshape(sb, -1, K) if sb > 0 else weights
        if want_alphas:
            ret["alphas"] = alphas.reshape(sb, -1, K) if sb > 0 else alphas
        if want_z_samps:
            ret["z_samps"] = z_samps.reshape(sb, -1, K) if sb > 0 else z_samps
        if want_rgb_samps:
            ret["rgb_samps"] = rgb_samps.reshape(sb, -1, K, rgb_samps.shape[-1]) if sb > 0 else rgb_samps
        return ret

    def sched_step(self, steps=1):
        if self.sched is None:
            return
        self.iter_idx += steps
        while self.last_sched.item() < len(self.sched[0]) and self.iter_idx.item() >= self.sched[0][self.last_sched.item()]:
            self.n_coarse = self.sched[1][self.last_sched.item()]
            self.n_fine = self.sched[2][self.last_sched.item()]
            self.last_sched += 1   # (the reference leaves using_fine as constructed, nerf.py:403-423: reproduced)

    @classmethod
    def from_conf(cls, conf, white_bkgd=False, eval_batch_size=100000):
        return cls(conf.get("n_coarse", 128), conf.get("n_fine", 0), n_fine_depth=conf.get("n_fine_depth", 0),
                   noise_std=conf.get("noise_std", 0.0), depth_std=conf.get("depth_std", 0.01),
                   white_bkgd=conf.get("white_bkgd", white_bkgd), lindisp=conf.get("lindisp", True),
                   eval_batch_size=conf.get("eval_batch_size", eval_batch_size), sched=conf.get("sched", None),
                   hard_alpha_cap=conf.get("hard_alpha_cap", False), lean_training_outputs=conf.get("lean_training_outputs", False))

    def bind_parallel(self, net, gpus=None, simple_output=False):
        """Same contract as nerf.py:440-457.  Multi-GPU is one process per GPU (DDP over RCCL); the reference's dead
        ``torch.nn.DataParallel`` hook is deliberately not reproduced."""
        if gpus is not None and len(gpus) > 1:
            raise NotImplementedError("use one process per GPU (torch.distributed / DDP over RCCL) instead of DataParallel")
        return _RenderWrapper(net, self, simple_output=simple_output)
