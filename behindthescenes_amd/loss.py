"""Drop-in for the reference's ``ReconstructionLoss`` (models/bts/model/loss.py:43-293) on the renderer's patch outputs.

For the criterion every shipped config trains with ("l1+ssim") the whole photometric term -- SSIM + L1 per patch, minimum over the
render views, invalid-ray masking, edge-aware smoothness -- and its gradient with respect to ``rgb`` / ``depth`` come from ONE HIP
pass per scale (``bts_photometric_loss``, csrc/bts_loss.hip); the ~60 small kernels and nine ``.item()`` synchronisations of the
reference become one launch, one reduction and one device-to-host copy for the logging dict.  The optional regularisers (depth /
alpha / surfaceness / depth-smoothness / ray-entropy, all off in the shipped configs) are a few elementwise torch expressions on
top.  Same constructor keys, same call signature, same ``(loss, loss_dict)`` result as the reference."""
import math
from collections.abc import MutableMapping

import torch
from torch.autograd import profiler

from . import native


class _PhotometricSums(torch.autograd.Function):
    """(sum of the rgb term, sum of the smoothness term, number of invalid rays) over all patches; differentiable w.r.t. rgb, depth."""

    @staticmethod
    def forward(ctx, rgb, depth, weights, invalid, rgb_gt, ph, pw, policy, eas, invalid_wsum=None, invalid_any=None):
        ctx.set_materialize_grads(False)
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        parts, g_rgb, g_depth = native.photometric_loss(rgb, depth if eas else None, weights, invalid, rgb_gt, ph, pw, policy, eas, 1.0, 1.0,
                                                        need_grad=need, invalid_wsum=invalid_wsum, invalid_any=invalid_any)
        ctx.save_for_backward(g_rgb, g_depth)
        ctx.had_depth = depth is not None
        return parts.sum(0)[:3]

    @staticmethod
    def backward(ctx, g):
        g_rgb, g_depth = ctx.saved_tensors
        if g is None:
            return (None,) * 11
        d_rgb = g_rgb * g[0] if (g_rgb is not None and ctx.needs_input_grad[0]) else None
        d_depth = g_depth * g[1] if (g_depth is not None and ctx.needs_input_grad[1]) else None
        return d_rgb, d_depth, None, None, None, None, None, None, None, None, None


def _flat(t, tail):
    """(n, pc, h, w, *tail) -> contiguous float32 (B, prod(tail))"""
    return t.reshape((-1,) + tuple(tail)).float().contiguous()


def _same_view(a, b):
    """True when a and b are the same values in the same memory (same storage offset, shape, strides, grad history)."""
    return a is b or (a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.stride() == b.stride() and a.dtype == b.dtype
                      and a._base is not None and a._base is b._base)


class LazyScalars(MutableMapping):
    """The loss dict of loss.py:219-229 ({name: python float}) without its synchronisation: the values are copied to pinned host memory
    asynchronously when the loss is computed and the host waits for that copy only when an entry is read (logging handlers read them
    every N iterations; a step whose dict nobody reads never stalls, and the backward is launched while the forward still runs).  A
    mutable Mapping: ``d["loss"]``, ``d.items()``, ``dict(d)``, ``d["x"] = 1.0``, ``d.update(...)`` behave like the reference's dict (the
    first access of any kind waits for the copy); pickling / ``copy.deepcopy`` (torch.save of an engine's output) yield a plain dict.
    It is NOT a ``dict`` subclass -- ``json.dumps`` walks a dict subclass' own storage behind its methods' back and would print the
    unmaterialised (empty) one: ``json.dumps(dict(d))`` is the spelling."""

    def __init__(self, keys, values: torch.Tensor):
        self._keys, self._vals = list(keys), None
        if values.is_cuda:
            self._host = torch.empty(values.shape, dtype=values.dtype, pin_memory=True)
            self._host.copy_(values, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record()
        else:
            self._host, self._event = values, None

    def _dict(self):
        if self._vals is None:
            if self._event is not None:
                self._event.synchronize()
            self._vals = dict(zip(self._keys, self._host.tolist()))
        return self._vals

    def __getitem__(self, k):
        return self._dict()[k]

    def __setitem__(self, k, v):
        self._dict()[k] = v

    def __delitem__(self, k):
        del self._dict()[k]

    def __iter__(self):
        return iter(self._keys if self._vals is None else self._vals)

    def __len__(self):
        return len(self._keys if self._vals is None else self._vals)

    def __reduce__(self):
        return (dict, (self._dict(),))

    def __repr__(self):
        return repr(self._dict())


class ReconstructionLoss:
    def __init__(self, config, use_automasking=False) -> None:
        self.criterion_str = config.get("criterion", "l2")
        if self.criterion_str not in ("l2", "l1", "l1+ssim"):
            raise ValueError(f"unknown criterion {self.criterion_str}")
        self.invalid_policy = config.get("invalid_policy", "strict")
        assert self.invalid_policy in ["strict", "weight_guided", "weight_guided_diverse", None, "none"]
        self.ignore_invalid = self.invalid_policy is not None and self.invalid_policy != "none"
        self.lambda_coarse = config.get("lambda_coarse", 1)
        self.lambda_fine = config.get("lambda_fine", 1)
        self.use_automasking = use_automasking
        self.lambda_entropy = config.get("lambda_entropy", 0)
        self.lambda_depth_reg = config.get("lambda_depth_reg", 0)
        self.lambda_alpha_reg = config.get("lambda_alpha_reg", 0)
        self.lambda_surfaceness_reg = config.get("lambda_surfaceness_reg", 0)
        self.lambda_edge_aware_smoothness = config.get("lambda_edge_aware_smoothness", 0)
        self.lambda_depth_smoothness = config.get("lambda_depth_smoothness", 0)
        self.median_thresholding = config.get("median_thresholding", False)
        self.alpha_reg_reduction = config.get("alpha_reg_reduction", "ray")
        self.alpha_reg_fraction = config.get("alpha_reg_fraction", 1 / 8)
        if self.alpha_reg_reduction not in ("ray", "slice"):
            raise ValueError(f"Unknown reduction for alpha regularization: {self.alpha_reg_reduction}")
        self._lin = {}    # (scales, rays per scale, fine present, lambdas[, device]) -> the (9 x 3 S) matrix of loss_matrix()
        if self.criterion_str != "l1+ssim" or use_automasking or self.median_thresholding or self.invalid_policy == "weight_guided_diverse":
            raise NotImplementedError("the fused HIP loss covers criterion 'l1+ssim' with invalid_policy strict / weight_guided / none, "
                                      "without automasking / median thresholding (every shipped config); got "
                                      f"criterion={self.criterion_str}, invalid_policy={self.invalid_policy}, "
                                      f"automasking={use_automasking}, median_thresholding={self.median_thresholding}")

    @staticmethod
    def get_loss_metric_names():
        return ["loss", "loss_rgb_coarse", "loss_rgb_fine", "loss_ray_entropy", "loss_depth_reg"]

    def _sums(self, level, level0, rgb_gt, eas):
        """-> ((sum of the rgb term, sum of the smoothness term, number of invalid rays) as one (3,) tensor, number of rays) of one set
        of renderer outputs in patch layout."""
        rgb = level["rgb"]                                   # (n, pc, h, w, nv, 3)
        n, pc, h, w, nv, c = rgb.shape
        if c != 3:
            raise NotImplementedError("the fused HIP loss takes 3 colour channels")
        B = n * pc * h * w
        if "weights" not in level0:     # lean training outputs: the renderer's epilogue reduced weights / invalid per ray and view
            sums = _PhotometricSums.apply(_flat(rgb, (nv * 3,)), _flat(level["depth"], ()) if eas else None, None, None,
                                          _flat(rgb_gt, (3,)).detach(), h, w, self.invalid_policy, eas,
                                          _flat(level0["invalid_wsum"], (nv,)).detach() if self.invalid_policy == "weight_guided" else None,
                                          _flat(level0["invalid_any"], (nv,)).detach() if self.invalid_policy == "strict" else None)
            return sums, B
        K = level0["weights"].shape[-1]
        sums = _PhotometricSums.apply(_flat(rgb, (nv * 3,)), _flat(level["depth"], ()) if eas else None,
                                      _flat(level0["weights"], (K,)).detach() if self.invalid_policy == "weight_guided" else None,
                                      _flat(level0["invalid"], (K, nv)).detach() if self.ignore_invalid else None,
                                      _flat(rgb_gt, (3,)).detach(), h, w, self.invalid_policy, eas)
        return sums, B

    def _photometric(self, level, level0, rgb_gt, eas):
        """-> (mean rgb term, mean smoothness term, invalid-ray ratio)"""
        sums, B = self._sums(level, level0, rgb_gt, eas)
        return sums[0] / B, sums[1] / B, sums[2] / B

    _KEYS = ["loss_rgb_coarse", "loss_rgb_fine", "loss_ray_entropy", "loss_depth_reg", "loss_alpha_reg", "loss_eas",
             "loss_depth_smoothness", "loss_invalid_ratio", "loss"]

    def _call_photometric_only(self, data):
        """The shipped configs' loss -- photometric term (coarse + the fine dict trainer.py:247-248 aliases to it) and edge-aware
        smoothness, no optional regulariser -- is LINEAR in the per-scale sums the HIP pass returns: loss and the whole logging dict are
        one (9 x 3 S) matrix applied to them.  The general path below spells the same algebra with ~25 zero-dimensional torch ops per
        scale (each a kernel launch, plus their autograd nodes and the zero fills of SelectBackward): ~190 launches and 0.5 ms of a
        3.7 ms exp_re10k.yaml step.  Returns None when the data needs the general path."""
        coarse_0, fine_0 = data["coarse"][0], data["fine"][0]
        S = len(data["coarse"])
        mask_keys = ("weights", "invalid") if "weights" in coarse_0 else ("invalid_wsum", "invalid_any")
        has_fine = []
        for coarse, fine in zip(data["coarse"], data["fine"]):
            if len(fine) > 0 and not (_same_view(fine["rgb"], coarse["rgb"]) and all(_same_view(fine_0[k], coarse_0[k]) for k in mask_keys)):
                return None
            has_fine.append(len(fine) > 0)
        eas_on = self.lambda_edge_aware_smoothness > 0
        sums, Bs = zip(*(self._sums(c, coarse_0, data["rgb_gt"], eas_on) for c in data["coarse"]))
        st = torch.stack(sums).reshape(-1)                      # (3 S): rgb, eas, invalid count per scale
        M = self.loss_matrix(S, Bs, tuple(has_fine), st.device)
        loss = torch.dot(M[8], st)
        return loss, LazyScalars(self._KEYS, torch.mv(M, st.detach()))

    def loss_matrix(self, S, Bs, has_fine, device=None):
        """The (9 x 3 S) matrix that takes the per-scale sums [rgb term, smoothness term, invalid rays] of the HIP loss pass to the logging
        dict (rows in ``_KEYS`` order; row 8 = the loss).  Cached per (layout, lambdas): the reference reads ``self.lambda_*`` on every
        call, so a schedule that changes one of them gets a new matrix."""
        eas_on = self.lambda_edge_aware_smoothness > 0
        key = (S, tuple(Bs), tuple(has_fine), float(self.lambda_coarse), float(self.lambda_fine), float(self.lambda_edge_aware_smoothness))
        M = self._lin.get(key)
        if M is None:
            M = torch.zeros(9, 3 * S, dtype=torch.float64)
            for s in range(S):
                lam = self.lambda_coarse + self.lambda_fine if has_fine[s] else 1.0    # (without a fine dict the rgb term enters unscaled)
                M[0, 3 * s] = self.lambda_coarse / Bs[s]
                M[1, 3 * s] = (self.lambda_fine if has_fine[s] else 0) / Bs[s]
                M[5, 3 * s + 1] = 1.0 / Bs[s]
                M[8, 3 * s] = lam / Bs[s] / S
                M[8, 3 * s + 1] = (self.lambda_edge_aware_smoothness / 2 ** s if eas_on else 0) / Bs[s] / S
            M[7, 2] = 1.0 / Bs[0]
            M = self._lin[key] = M.float()
        if device is not None and device.type != "cpu":
            dkey = key + (device,)
            Md = self._lin.get(dkey)
            if Md is None:
                Md = self._lin[dkey] = M.to(device)
            return Md
        return M

    def __call__(self, data):
        with profiler.record_function("loss_computation"):       # loss.py:84
            return self._call(data)

    def _call(self, data):
        if not (self.lambda_depth_reg > 0 or self.lambda_alpha_reg > 0 or self.lambda_surfaceness_reg > 0 or self.lambda_depth_smoothness > 0
                or self.lambda_entropy > 0):
            res = self._call_photometric_only(data)
            if res is not None:
                return res
        n_scales = len(data["coarse"])
        coarse_0, fine_0 = data["coarse"][0], data["fine"][0]
        dev = coarse_0["rgb"].device
        if "alphas" not in coarse_0 and (self.lambda_alpha_reg > 0 or self.lambda_surfaceness_reg > 0 or self.lambda_entropy > 0):
            raise KeyError("the alpha / surfaceness / entropy regularisers read the per-sample `alphas`, which this render dict does not "
                           "carry: call the renderer with want_alphas=True and without lean_training_outputs (NeRFRenderer drops the "
                           "per-sample tensors in that mode)")
        zero = torch.zeros((), device=dev)
        loss = zero
        m = dict(coarse=zero, fine=zero, depth_reg=zero, alpha_reg=zero, surf=zero, eas=zero, dsmooth=zero, inv=zero)
        keep_cache = None

        def keep():   # 1 - invalid ray mask (n, pc, h, w) for the optional regularisers
            nonlocal keep_cache
            if keep_cache is None and "weights" not in coarse_0:    # lean training outputs
                if self.invalid_policy == "strict":
                    bad = torch.all(coarse_0["invalid_any"] > .5, dim=-1)
                elif self.invalid_policy == "weight_guided":
                    bad = torch.all(coarse_0["invalid_wsum"] > .9, dim=-1)
                else:
                    bad = torch.zeros(coarse_0["depth"].shape, dtype=torch.bool, device=dev)
                keep_cache = 1 - bad.to(torch.float32)
            if keep_cache is None:
                inv, wts = coarse_0["invalid"], coarse_0["weights"]
                if self.invalid_policy == "strict":
                    bad = torch.all(torch.any(inv > .5, dim=-2), dim=-1)
                elif self.invalid_policy == "weight_guided":
                    bad = torch.all((inv.to(torch.float32) * wts.unsqueeze(-1)).sum(-2) > .9, dim=-1)
                else:
                    bad = torch.zeros(inv.shape[:-2], dtype=torch.bool, device=dev)
                keep_cache = 1 - bad.to(torch.float32)
            return keep_cache

        for scale in range(n_scales):
            coarse, fine = data["coarse"][scale], data["fine"][scale]
            eas_on = self.lambda_edge_aware_smoothness > 0
            rgb_loss, eas, inv_ratio = self._photometric(coarse, coarse_0, data["rgb_gt"], eas_on)
            if scale == 0:
                m["inv"] = inv_ratio.detach()
            m["coarse"] = m["coarse"] + rgb_loss.detach() * self.lambda_coarse
            if len(fine) > 0:
                mask_keys = ("weights", "invalid") if "weights" in coarse_0 else ("invalid_wsum", "invalid_any")
                if _same_view(fine["rgb"], coarse["rgb"]) and all(_same_view(fine_0[k], coarse_0[k]) for k in mask_keys):
                    # trainer.py:247-248 aliases fine = dict(coarse): reconstruct() then views the SAME storage once per dict, so the
                    # tensors are different Python objects over identical memory -- one launch serves both terms
                    fine_loss = rgb_loss
                else:
                    fine_loss, _, _ = self._photometric(fine, fine_0, data["rgb_gt"], False)
                m["fine"] = m["fine"] + fine_loss.detach() * self.lambda_fine
                rgb_loss = rgb_loss * self.lambda_coarse + fine_loss * self.lambda_fine
            loss = loss + rgb_loss

            if self.lambda_depth_reg > 0:
                d = coarse["depth"]
                s = ((d[:, :, 1:, :] - d[:, :, :-1, :]) ** 2).mean() + ((d[:, :, :, 1:] - d[:, :, :, :-1]) ** 2).mean()
                m["depth_reg"] = m["depth_reg"] + s.detach()
                loss = loss + s * self.lambda_depth_reg
            if self.lambda_alpha_reg > 0:
                a = coarse["alphas"]
                a_sum = a[..., :-1].sum(-1)
                cap = torch.ones_like(a_sum) * (a.shape[-1] * self.alpha_reg_fraction)
                if self.ignore_invalid:
                    a_sum, cap = a_sum * keep(), cap * keep()
                if self.alpha_reg_reduction == "ray":
                    s = (a_sum - cap).clamp_min(0)
                else:
                    s = (a_sum.sum(dim=-1) - cap.sum(dim=-1)).clamp_min(0) / a_sum.shape[-1]
                s = s.mean()
                m["alpha_reg"] = m["alpha_reg"] + s.detach()
                loss = loss + s * self.lambda_alpha_reg
            if self.lambda_surfaceness_reg > 0:
                a = coarse["alphas"]
                p = (-torch.log(torch.exp(-a.abs()) + torch.exp(-(1 - a).abs()))).mean(-1)
                if self.ignore_invalid:
                    p = p * keep()
                s = p.mean()
                m["surf"] = m["surf"] + s.detach()
                loss = loss + s * self.lambda_surfaceness_reg
            if eas_on:
                m["eas"] = m["eas"] + eas.detach()
                loss = loss + eas * self.lambda_edge_aware_smoothness / (2 ** scale)
            if self.lambda_depth_smoothness > 0:
                d = coarse["depth"]
                s = ((d[..., :-1, :] - d[..., 1:, :]) ** 2).mean() + ((d[..., :, :-1] - d[..., :, 1:]) ** 2).mean()
                m["dsmooth"] = m["dsmooth"] + s.detach()
                loss = loss + s * self.lambda_depth_smoothness

        loss = loss / n_scales
        ent = zero
        if self.lambda_entropy > 0:
            a = coarse_0["alphas"] + 1e-5
            dens = a / a.sum(dim=-1, keepdim=True)
            ent = (-(dens * torch.log(dens)).sum(-1) / math.log2(a.shape[-1]) * keep()).mean()
            loss = loss + ent * self.lambda_entropy

        # one device-to-host copy for the whole logging dict (the reference pays one synchronisation per entry), and not even that one
        # stalls the step: the copy is asynchronous, the host waits for it when somebody READS an entry (LazyScalars)
        vals = torch.stack([m["coarse"], m["fine"], ent.detach(), m["depth_reg"], m["alpha_reg"], m["eas"], m["dsmooth"], m["inv"],
                            loss.detach()])
        return loss, LazyScalars(self._KEYS, vals)
