"""Encoder hook.  The CNN is OUT of the hot path and stays PyTorch-ROCm: any module with
``latent_size``, ``scales`` and ``forward(images (N,3,H,W)) -> [feature maps (N,C,h,w)]`` can be registered here.
``feature_map`` is the learnable-feature-map stand-in the reference itself uses for overfitting
(models/bts/trainer_overfit.py:24-33); it is what the synthetic benchmarks and tests use."""
import torch
from torch import nn

_REGISTRY = {}


def register_backbone(name, factory):
    _REGISTRY[name] = factory


class FeatureMapEncoder(nn.Module):
    def __init__(self, size, feat_dim, num_views=1, n_scales=1, pyramid=False, channels_last=False):
        """pyramid=True halves the resolution per scale, like the decoder outputs of the reference's Monodepth2 (BTSNet.encode
        resizes every scale back to scale 0's size, models_bts.py:111-119).  channels_last=True keeps the maps in torch's channels_last
        format -- (num_views, h, w, feat_dim) in memory, what a CNN running MIOpen's NHWC kernels (the shipped Monodepth2 does) hands
        over; the default is the NCHW-contiguous tensor a plain ``nn.Conv2d`` stack returns."""
        super().__init__()
        sizes = [tuple(max(1, d >> s) for d in size) if pyramid else tuple(size) for s in range(n_scales)]
        fmt = torch.channels_last if channels_last else torch.contiguous_format
        self.feats = nn.ParameterList([nn.Parameter(torch.randn(num_views, feat_dim, *sz).contiguous(memory_format=fmt)) for sz in sizes])
        self.latent_size = feat_dim
        self.scales = list(range(n_scales))

    def forward(self, x):
        n = x.shape[0]
        out = []
        for f in self.feats:
            # the parameter itself when the batch matches: a slice would cost a zero fill + a copy of the whole map in its backward
            out.append(f if f.shape[0] == n else (f[:n] if f.shape[0] > n else f.expand(n, -1, -1, -1)))
        return out

    @classmethod
    def from_conf(cls, conf, **kw):
        return cls(tuple(conf["size"]), conf.get("d_out", 64), conf.get("num_views", 1), conf.get("n_scales", 1), conf.get("pyramid", False),
                   conf.get("channels_last", False))


register_backbone("feature_map", FeatureMapEncoder.from_conf)


def make_backbone(conf, **kwargs):
    enc_type = conf.get("type", "monodepth2")
    if enc_type in _REGISTRY:
        return _REGISTRY[enc_type](conf, **kwargs)
    if enc_type == "monodepth2":
        try:
            from .monodepth2 import Monodepth2
        except ImportError as e:  # pragma: no cover
            raise NotImplementedError("encoder type 'monodepth2' needs behindthescenes_amd.monodepth2; alternatively register the "
                                      "reference's own module with behindthescenes_amd.register_backbone (INTEGRATION.md)") from e
        return Monodepth2.from_conf(conf, **kwargs)
    raise NotImplementedError(f"Unsupported encoder type: {enc_type}")
