"""``BTSNet`` -- the density field, with the reference's constructor keys, attributes, ``encode`` / ``forward`` protocol and
state-dict layout (models/bts/model/models_bts.py:17-338), but whose per-point work (projection, bilinear feature fetch,
positional encoding, MLP, softplus, colour fetch) is done by the fused HIP kernels in libbts_render.so.

What stays PyTorch-ROCm: the CNN encoder call inside ``encode`` and the 4x4 pose inverse.  ``encode`` additionally hands
the feature map / colour frames over to the renderer's HBM layouts (projected channels-last G = F . w_in[:, :C]^T, rgb0-packed
frames) with HIP kernels; under autograd the hand-over is differentiable (per-pixel GEMMs), so gradients of the renderer reach
the CNN and lin_in exactly as they do in the reference."""
import dataclasses

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import profiler

from . import native
from .backbone import make_backbone
from .code import PositionalEncoding
from .mlp import make_mlp

EPS = 1e-3


def _take(x, ids):
    """``x[:, ids]`` for a Python list of frame indices without the host-to-device copy of an index tensor: list indexing makes
    torch upload the indices from pageable memory, which blocks the host behind whatever the stream is still running (the previous
    frame's render kernel) and so serialises host and device.  Consecutive ids are a view; anything else a stack of slices."""
    if torch.is_tensor(ids):
        return x[:, ids]
    ids = [int(i) for i in ids]
    if len(ids) > 0 and ids == list(range(ids[0], ids[0] + len(ids))):
        return x[:, ids[0]:ids[0] + len(ids)]
    return torch.stack([x[:, i] for i in ids], dim=1)


class BTSNet(nn.Module):
    def __init__(self, conf):
        super().__init__()
        self.d_min = conf.get("z_near")
        self.d_max = conf.get("z_far")
        self.learn_empty = conf.get("learn_empty", True)
        self.empty_empty = conf.get("empty_empty", False)
        self.inv_z = conf.get("inv_z", True)
        self.color_interpolation = conf.get("color_interpolation", "bilinear")
        self.code_mode = conf.get("code_mode", "z")
        if self.code_mode not in ["z", "distance"]:
            raise NotImplementedError(f"Unknown mode for positional encoding: {self.code_mode}")
        if self.color_interpolation != "bilinear":
            raise NotImplementedError("only bilinear colour interpolation (the default of every config) is implemented")

        self.encoder = make_backbone(conf["encoder"])
        self.code_xyz = PositionalEncoding.from_conf(conf["code"], d_in=3)
        self.flip_augmentation = conf.get("flip_augmentation", False)
        self.return_sample_depth = conf.get("return_sample_depth", False)
        self.sample_color = conf.get("sample_color", True)
        if self.return_sample_depth:
            raise NotImplementedError("return_sample_depth is not used by any shipped config")

        d_in = self.encoder.latent_size + self.code_xyz.d_out
        # models_bts.py:41-42: colours predicted by the MLP (sample_color=False) make it four outputs wide; no shipped config does that,
        # and the fused kernels know the one-output density MLP only: that mode runs as a PyTorch composition (torch_modes.py)
        self._d_in, self._d_out = d_in, (1 if self.sample_color else 4)
        self.mlp_coarse = make_mlp(conf["mlp_coarse"], d_in, d_out=self._d_out)
        # models_bts.py:45, 293-307: a separate MLP for the fine pass (`coarse=False`); every shipped config says `type: empty` and the
        # coarse MLP serves both.  Here it is one more packed parameter vector (and its own projected map G) through the same kernels.
        self.mlp_fine = make_mlp(conf["mlp_fine"], d_in, d_out=self._d_out, allow_empty=True)
        self._combined = False        # set by encode(): several encoder views merged per point (combine_ids, the waymo mode)
        if self.learn_empty:
            self.empty_feature = nn.Parameter(torch.randn((self.encoder.latent_size,), requires_grad=True))
        self._scale = 0
        self._native = {}   # scale -> native.FieldTensors
        if conf.get("fused_handover", False):
            # rounds 2 - 3 let the decoder's last convolution write G; measured slower than convolution + projection
            # passes and removed in round 4
            raise NotImplementedError("fused_handover was removed: the projection passes (bts_project_features / _bwd) are the hand-over")
        # models_bts.py:115-117 resizes the decoder's coarser scales to scale 0's size (F.interpolate, nearest) before the renderer
        # samples them.  A nearest resize by 2^s only repeats texels: the kernels index the scale's own map (BtsFieldCfg.feat_shift)
        # -- bit-identical results, no resized map (and none of its gradient) in HBM, the projection G = F . W^T runs on 1 / 4^s of
        # the texels.  `native_scale_maps: false` materialises the resized maps as the reference does.
        self.native_scale_maps = bool(conf.get("native_scale_maps", True))
        self.spec = native.FieldSpec(C=self.encoder.latent_size, d_hidden=self.mlp_coarse.d_hidden,
                                     n_blocks=self.mlp_coarse.n_blocks, num_freqs=self.code_xyz.num_freqs,
                                     freq_factor=self.code_xyz.freq_factor, d_min=float(self.d_min), d_max=float(self.d_max),
                                     inv_z=bool(self.inv_z), code_mode=self.code_mode, learn_empty=bool(self.learn_empty),
                                     empty_empty=bool(self.empty_empty))
        self.spec_fine = self.spec if self.mlp_fine is None else \
            dataclasses.replace(self.spec, d_hidden=self.mlp_fine.d_hidden, n_blocks=self.mlp_fine.n_blocks)
        if not self.code_xyz.include_input:
            raise NotImplementedError("include_input=False is not used by any shipped config")

    @property
    def torch_mode(self):
        """True when the field is served by the PyTorch compositions of torch_modes.py instead of the fused HIP kernels: the two modes
        no shipped config uses -- MLP-predicted colours (`sample_color: false`) and merged encoder views
        (`encode(..., combine_ids=...)` or more than one encoder view)."""
        return (not self.sample_color) or self._combined

    def mlp(self, coarse=True):
        """models_bts.py:293-307: the MLP of the coarse pass, or of the fine pass when a separate one was configured."""
        return self.mlp_coarse if coarse or self.mlp_fine is None else self.mlp_fine

    def _scale_shift(self, size, size0):
        """s if a map of `size` is scale 0's size divided by 2^s (and may be handed over as it is), else None (resize it)."""
        (h, w), (h0, w0) = size, size0
        if (h, w) == (h0, w0):
            return 0
        if self.native_scale_maps:
            for sh in range(1, 7):
                if (h << sh, w << sh) == (h0, w0):
                    return sh
        return None

    @property
    def grid_f_features(self):
        """models_bts.py:117, 128: the encoder's maps of every scale at scale 0's size, (n, nv_enc, C, H, W) each.  The renderer does
        not need them (see native_scale_maps); built on first access for callers that do."""
        if not getattr(self, "_has_latents", False):
            return None
        if self._grid_f_features is None:
            h_, w_ = self._grid_size
            self._grid_f_features = [il if il.shape[-2:] == (h_, w_) else
                                     F.interpolate(il[:, 0], (h_, w_)).unsqueeze(1) for il in self._latents_ms]
        return self._grid_f_features

    @property
    def grid_c_imgs(self):
        """models_bts.py:82, 129: the render views' frames in [0, 1], (n, nv, 3, H, W).  The renderer reads its own rgb0-packed copy
        (x * .5 + .5 happens inside bts_pack_rgb); this tensor is built when somebody asks for it."""
        src = getattr(self, "_grid_c_src", None)
        if src is None:
            return None
        if self._grid_c_raw:
            self._grid_c_src, self._grid_c_raw = src * .5 + .5, False
        return self._grid_c_src

    # ---- reference protocol -------------------------------------------------------------------------------------
    def set_scale(self, scale):
        self._scale = scale

    def get_scale(self):
        return self._scale

    def compute_grid_transforms(self, *args, **kwargs):
        pass

    def encode(self, images, Ks, poses_c2w, ids_encoder=None, ids_render=None, images_alt=None, combine_ids=None):
        """images (n,v,3,H,W) in [-1,1]; Ks (n,v,3,3) normalised intrinsics; poses_c2w (n,v,4,4)  (models_bts.py:65-136)."""
        # a new step: a new autograd graph for the packed parameter vector -- and the one point where edits packed() cannot see
        # (`p.data.copy_()`, an EMA swap: no version bump) are picked up, so also without autograd
        self.mlp_coarse.invalidate_packed()
        if self.mlp_fine is not None:
            self.mlp_fine.invalidate_packed()
        if ids_encoder is None:
            ids_encoder = list(range(images.shape[1]))
        self._combined = combine_ids is not None or len(ids_encoder) != 1
        # (the PyTorch-composed modes also run where torch runs; everything else is HIP and says so loudly on a CPU tensor)
        poses_w2c = torch.inverse(poses_c2w) if (self.torch_mode and not poses_c2w.is_cuda) else native.invert_small(poses_c2w)
        images_encoder, Ks_encoder, poses_w2c_encoder = _take(images, ids_encoder), _take(Ks, ids_encoder), _take(poses_w2c, ids_encoder)
        colours = images_alt if images_alt is not None else None
        if ids_render is None:
            ids_render = list(range(images.shape[1]))
        n, nv_enc, c, h, w = images_encoder.shape
        if self.torch_mode:
            return self._encode_torch_mode(images, Ks, poses_w2c, images_encoder, Ks_encoder, poses_w2c_encoder, ids_encoder, ids_render, colours,
                                           combine_ids)

        do_flip = bool(self.flip_augmentation and self.training and (torch.rand(1) > .5).item())
        if do_flip:
            images_encoder = torch.flip(images_encoder, dims=(-1,))
        enc_in = images_encoder.reshape(n * nv_enc, c, h, w)
        image_latents_ms = self.encoder(enc_in)
        if do_flip:
            image_latents_ms = [torch.flip(il, dims=(-1,)) for il in image_latents_ms]
        _, _, h_, w_ = image_latents_ms[0].shape
        self._shift_ms = [self._scale_shift(il.shape[-2:], (h_, w_)) for il in image_latents_ms]
        self._latents_ms = [(il if sh is not None else F.interpolate(il, (h_, w_))).unsqueeze(1)
                            for il, sh in zip(image_latents_ms, self._shift_ms)]
        self._shift_ms = [sh or 0 for sh in self._shift_ms]
        self._grid_size = (h_, w_)
        self._has_latents = True
        self._grid_f_features = None
        self.grid_f_Ks = Ks_encoder
        self.grid_f_poses_w2c = poses_w2c_encoder
        self.grid_f_combine = None
        # colours: (x*.5+.5) fused into the rgb0 packing kernel unless the caller supplies processed frames
        src = _take(colours if colours is not None else images, ids_render)
        self._grid_c_src, self._grid_c_raw = src, colours is None     # grid_c_imgs (models_bts.py:82) is materialised on first access
        self.grid_c_Ks = _take(Ks, ids_render)
        self.grid_c_poses_w2c = _take(poses_w2c, ids_render)
        self.grid_c_combine = None

        # ---- hand-over into the renderer's HBM layouts
        self._native = {}
        nv = src.shape[1]
        native.check_supported(self.spec, nv)
        if nv:
            scale, shift = (0.5, 0.5) if self._grid_c_raw else (1.0, 0.0)   # x * .5 + .5 inside the packing kernel (mul, then add)
            self._imgs_nhwc4 = native.pack_rgb(src.detach().float().contiguous(), scale, shift)
            self._K_r = self.grid_c_Ks.detach().float().contiguous()
            self._w2c_r = self.grid_c_poses_w2c.detach().float().contiguous()
        else:
            self._imgs_nhwc4 = self._K_r = self._w2c_r = None
        self._K_enc = Ks_encoder[:, 0].detach().float().contiguous()
        self._w2c_enc = poses_w2c_encoder[:, 0].detach().float().contiguous()
        # a render view that IS the encoder frame (eval_depth: ids_render = [0]) shares its camera bit for bit: the kernels skip its
        # second projection (BtsFieldCfg.enc_render_view)
        ids_r, id_e = [int(i) for i in ids_render], int(ids_encoder[0])
        self._enc_view = ids_r.index(id_e) if id_e in ids_r else -1

    def _encode_torch_mode(self, images, Ks, poses_w2c, images_encoder, Ks_encoder, poses_w2c_encoder, ids_encoder, ids_render, colours, combine_ids):
        """encode() for the PyTorch-composed modes (models_bts.py:93-136): every encoder view's maps at scale 0's size, the view groups."""
        from . import torch_modes
        torch_modes.warn_once("combine_ids / several encoder views" if self._combined else "sample_color=False")
        n, nv_enc, c, h, w = images_encoder.shape
        enc_groups = ren_groups = None
        if combine_ids is not None:
            groups = [list(g) for g in combine_ids]
            grouped = set(sum(groups, []))
            groups += [[i] for i in range(images.shape[1]) if i not in grouped]
            pos_e = {int(f): i for i, f in enumerate(ids_encoder)}
            pos_r = {int(f): i for i, f in enumerate(ids_render)}
            enc_groups = [g for g in ([pos_e[i] for i in grp if i in pos_e] for grp in groups) if g]
            ren_groups = [g for g in ([pos_r[i] for i in grp if i in pos_r] for grp in groups) if g]
        do_flip = bool(self.flip_augmentation and self.training and (torch.rand(1) > .5).item())
        enc_in = torch.flip(images_encoder, dims=(-1,)) if do_flip else images_encoder
        latents = self.encoder(enc_in.reshape(n * nv_enc, c, h, w))
        if do_flip:
            latents = [torch.flip(il, dims=(-1,)) for il in latents]
        h_, w_ = latents[0].shape[-2:]
        self._grid_f_features = [F.interpolate(il, (h_, w_)).view(n, nv_enc, -1, h_, w_) for il in latents]
        self._has_latents, self._grid_size = True, (h_, w_)
        self._latents_ms, self._shift_ms = None, None
        self.grid_f_Ks, self.grid_f_poses_w2c, self.grid_f_combine = Ks_encoder, poses_w2c_encoder, enc_groups
        src = _take(colours if colours is not None else images, ids_render)
        self._grid_c_src, self._grid_c_raw = src, colours is None
        self.grid_c_Ks, self.grid_c_poses_w2c, self.grid_c_combine = _take(Ks, ids_render), _take(poses_w2c, ids_render), ren_groups
        self._native = {}

    def invalidate_field_state(self):
        """Forget what the last ``encode`` left behind (maps, cameras, the projected-map cache).  The fused step / frame
        (train_step.py) hand the encoder's output to the library themselves: a later ``forward`` / ``occupancy_profile`` / render must
        not silently run on the PREVIOUS encode's maps -- it raises until ``encode`` is called again."""
        self._has_latents = False
        self._native = {}
        self._latents_ms = self._shift_ms = self._grid_f_features = None
        self._grid_c_src = None

    def native_field(self, coarse=True, sampled=None) -> "native.FieldTensors":
        """Field state of the current scale in the C-ABI layouts.  The projected feature map G = F . w_in[:, :C]^T is built lazily
        per scale (one HIP pass that also does the NCHW -> channels-last hand-over) and cached until the next ``encode`` or until
        lin_in.weight changes.  Under autograd the projection is differentiable w.r.t. F and the MLP parameters.  ``coarse=False`` with
        a separate fine MLP: the map projected with THAT MLP's lin_in.
        ``sampled`` = (rays (n*Bp, 8), z_samp | None, jitter | None, lindisp): a ONE-SHOT field for the render of exactly these samples
        -- only the 64-texel tiles of G their taps land in are projected (a training step's rays read 10-40 % of them), the rest of the
        map is uninitialised memory; never cached, never handed to field queries."""
        if self.torch_mode:
            raise native.BtsNativeError("this field runs as a PyTorch composition (sample_color=False / merged encoder views, torch_modes.py): "
                                        "it has no state in the fused kernels' layouts")
        if not getattr(self, "_has_latents", False) or self._latents_ms is None:
            raise native.BtsNativeError("no field state: call encode() first (a fused training step / evaluation frame leaves none behind)")
        s = self._scale
        fine = not coarse and self.mlp_fine is not None
        mlp, spec = (self.mlp_fine, self.spec_fine) if fine else (self.mlp_coarse, self.spec)
        if sampled is not None:
            rays, z_samp, jitter, lindisp = sampled
            f = self._latents_ms[s]
            f = f.reshape(f.shape[0], *f.shape[2:]).float()
            # the tile flags' geometry follows the map's layout (BtsFieldCfg.tile_blocks: 16 x 4 blocks are faster with a channels-last
            # map, runs of 64 texels with an NCHW one); everything that touches this render's flag arrays shares this spec
            spec = dataclasses.replace(spec, tile_blocks=native.is_channels_last(f))
            sh = self._shift_ms[s]
            tiles = native.mark_sampled_tiles(spec, f.shape[0], f.shape[-2] << sh, f.shape[-1] << sh, sh, self._K_enc, self._w2c_enc, rays, z_samp,
                                              jitter, lindisp)
            link = native.ProjLink()
            proj = native.ProjectFunction.apply(f, mlp.packed(), spec, link, tiles)
            ft = native.FieldTensors(spec, proj, self._K_enc, self._w2c_enc, self._imgs_nhwc4, self._K_r, self._w2c_r,
                                     self.empty_feature if self.learn_empty else None, feat_shift=self._shift_ms[s], enc_view=self._enc_view)
            ft.proj_link = link
            ft.partial = (rays.data_ptr(), rays.shape[0])     # valid for THIS sample set only (native.FieldTensors.partial)
            return ft
        version = (mlp.lin_in.weight._version, torch.is_grad_enabled())
        hit = self._native.get((s, fine))
        if hit is None or hit[1] != version:
            f = self._latents_ms[s]                            # (n, 1, C, h, w) -> (n, C, h, w): a pure view (selecting [:, 0] would
            f = f.reshape(f.shape[0], *f.shape[2:]).float()    # cost a zero fill + a copy of the whole map in its backward)
            spec = dataclasses.replace(spec, tile_blocks=native.is_channels_last(f))   # (the geometry of the gradient's tile flags, see above)
            link = native.ProjLink()   # lets a single render of this map hand its (sparse) gradient to the projection's backward as tiles
            proj = native.ProjectFunction.apply(f, mlp.packed(), spec, link)
            ft = native.FieldTensors(spec, proj, self._K_enc, self._w2c_enc, self._imgs_nhwc4, self._K_r, self._w2c_r,
                                     self.empty_feature if self.learn_empty else None, feat_shift=self._shift_ms[s], enc_view=self._enc_view)
            ft.proj_link = link
            self._native[(s, fine)] = (ft, version)
        return self._native[(s, fine)][0]

    def forward(self, xyz, coarse=True, viewdirs=None, far=False, only_density=False):
        """xyz (n, P, 3) world points -> rgb (n,P,nv*3), invalid (n,P,nv) float, sigma (n,P,1)  (models_bts.py:266-338).
        Forward-only (the reference's callers of this entry point -- occupancy profiles, LiDAR / 3D-bbox evaluators -- run it
        under no_grad); training goes through the renderer's composite, which is differentiable."""
        if self.torch_mode:
            from . import torch_modes
            with profiler.record_function("model_inference"):
                return torch_modes.field_forward(self, xyz, coarse=coarse, only_density=only_density)
        ft = self.native_field(coarse)
        with torch.no_grad(), profiler.record_function("model_inference"):   # models_bts.py:275
            rgb, invalid, sigma = native.field_query(ft, self.mlp(coarse).packed().detach(), xyz.detach().float().contiguous(),
                                                     only_density=only_density)
        nv = self._grid_c_src.shape[1]
        if only_density:
            rgb = torch.zeros((xyz.shape[0], xyz.shape[1], nv * 3), device=sigma.device)
            invalid = invalid.unsqueeze(1)   # the reference returns (n, nv_enc=1, P, 1) here (models_bts.py:337)
        return rgb, invalid, sigma

    def occupancy_profile(self, q_pts, levels, threshold=8.0, only_density=False, want_sigma=False):
        """scripts/inference_setup.py:201-229 (render_profile) as one fused pass: q_pts (n, levels * columns, 3), a dense grid with the
        vertical level slowest (get_pts' order) -> (n, columns) fraction of levels whose running density sum stays <= threshold, points
        flagged invalid by any view counting as density 1 (only_density: by the encoder view only).  The reference chunks the 4.19 M
        points of its grid into 50 000-point field queries and post-processes in torch; here no per-point tensor reaches HBM."""
        ft = self.native_field()
        with torch.no_grad(), profiler.record_function("model_inference"):
            return native.occupancy_profile(ft, self.mlp_coarse.packed().detach(), q_pts.detach().float().contiguous(), int(levels),
                                            float(threshold), only_density, want_sigma)
