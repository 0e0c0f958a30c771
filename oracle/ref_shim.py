"""TEST INFRASTRUCTURE ONLY -- import shim for the *real* reference.

Makes the hot-path modules of the upstream reference (a read-only checkout at
``/root/reference``) importable in a container that lacks its third-party
dependencies, so that golden vectors can be generated from the reference's own
code and so that our restatement (``oracle/bts_oracle.py``) can be validated
against it.  Nothing here is shipped, and nothing here may be used on the GPU
box (the reference tree does not exist there): only
``tests/golden/gen_golden.py`` and the ``needs_reference`` CPU tests import it.

What is stubbed (module scope imports of the reference that do not exist here):
``dotmap`` (nerf.py:9), ``cv2`` + ``torchvision`` (util.py:1-4, monodepth2.py:15),
``lpips`` (image_processor.py:5), ``omegaconf`` (ray_sampler.py:2).  The real
encoder cannot be built offline (monodepth2.py:258 downloads weights), so
``make_backbone`` is replaced by a learnable feature map in the spirit of the
reference's own ``EncoderDummy`` (trainer_overfit.py:24-33).
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("BTS_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "models", "bts"))


class _AttrDict(dict):
    """Minimal stand-in for dotmap.DotMap: attribute access + toDict()."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def toDict(self):
        return {k: (v.toDict() if isinstance(v, _AttrDict) else v) for k, v in self.items()}


def _install_stubs():
    if "dotmap" not in sys.modules:
        m = types.ModuleType("dotmap")
        m.DotMap = _AttrDict
        sys.modules["dotmap"] = m
    if "cv2" not in sys.modules:
        m = types.ModuleType("cv2")
        m.COLORMAP_HOT = 11
        sys.modules["cv2"] = m
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tr = types.ModuleType("torchvision.transforms")
        mo = types.ModuleType("torchvision.models")
        ut = types.ModuleType("torchvision.utils")
        fn = types.ModuleType("torchvision.transforms.functional")

        class _ResNet:  # base class referenced by monodepth2.py
            pass

        mo.ResNet = _ResNet
        for name in ("resnet18", "resnet34", "resnet50", "resnet101", "resnet152"):
            setattr(mo, name, None)
        mo.resnet = types.ModuleType("torchvision.models.resnet")
        mo.resnet.BasicBlock = object
        mo.resnet.Bottleneck = object
        tr.functional = fn
        tv.transforms, tv.models, tv.utils = tr, mo, ut
        sys.modules.update({
            "torchvision": tv, "torchvision.transforms": tr, "torchvision.models": mo,
            "torchvision.utils": ut, "torchvision.transforms.functional": fn,
            "torchvision.models.resnet": mo.resnet,
        })
    if "lpips" not in sys.modules:
        m = types.ModuleType("lpips")
        m.LPIPS = object
        m.normalize_tensor = lambda x, eps=1e-10: x
        sys.modules["lpips"] = m
    if "omegaconf" not in sys.modules:
        m = types.ModuleType("omegaconf")
        m.ListConfig = list
        sys.modules["omegaconf"] = m


_LOADED = None


def load_reference():
    """Returns a namespace with the reference's hot-path classes (imported unmodified)."""
    global _LOADED
    if _LOADED is not None:
        return _LOADED
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True  # never drop __pycache__ into the read-only tree
    _install_stubs()
    # our own repo has no top-level 'models'/'utils' packages, so this cannot shadow anything of ours
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import torch
    from torch import nn

    import models.bts.model.models_bts as ref_models_bts
    from models.bts.model.loss import ReconstructionLoss
    from models.bts.model.ray_sampler import ImageRaySampler, PatchRaySampler, RandomRaySampler
    from models.common.render.nerf import NeRFRenderer
    from models.common.util import util as ref_util
    from utils.projection_operations import distance_to_z

    class FeatureMapEncoder(nn.Module):
        """Learnable feature maps instead of the CNN (pattern: trainer_overfit.py:24-33).
        ``feats``: list (scales) of (num_images, C, H, W) tensors; forward expands nothing, it
        returns the stored maps for the first n images."""

        def __init__(self, feats):
            super().__init__()
            self.feats = nn.ParameterList([nn.Parameter(f.clone()) for f in feats])
            self.latent_size = feats[0].shape[1]
            self.scales = list(range(len(feats)))

        def forward(self, x):
            n = x.shape[0]
            return [f[:n] for f in self.feats]

    def make_net(conf, feats):
        """BTSNet(conf) with make_backbone monkey-patched to the feature-map encoder."""
        orig = ref_models_bts.make_backbone
        ref_models_bts.make_backbone = lambda c, **kw: FeatureMapEncoder(feats)
        try:
            net = ref_models_bts.BTSNet(conf)
        finally:
            ref_models_bts.make_backbone = orig
        return net

    ns = types.SimpleNamespace(
        torch=torch, BTSNet=ref_models_bts.BTSNet, make_net=make_net, NeRFRenderer=NeRFRenderer,
        ImageRaySampler=ImageRaySampler, PatchRaySampler=PatchRaySampler, RandomRaySampler=RandomRaySampler,
        ReconstructionLoss=ReconstructionLoss, gen_rays=ref_util.gen_rays, unproj_map=ref_util.unproj_map,
        distance_to_z=distance_to_z, FeatureMapEncoder=FeatureMapEncoder,
    )
    _LOADED = ns
    return ns
