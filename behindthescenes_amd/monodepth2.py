"""``Monodepth2`` -- the reference's image encoder (ResNet + skip decoder emitting four scales of ``d_out``-channel features,
models/common/backbones/monodepth2.py:71-302) with the reference's / torchvision's state-dict keys, so that reference checkpoints
load unchanged:  ``encoder.encoder.{conv1,bn1,layer1..4.*,fc}``, ``decoder.decoder.{0..9}.conv.conv.{weight,bias}`` (the ten
ConvBlocks, scale 4 down to 0) and ``decoder.decoder.{10..13}.conv.{weight,bias}`` (the per-scale output convolutions).

The CNN is NOT part of the render path and stays PyTorch-ROCm (MIOpen) -- written here without torchvision, which the image lacks.
Its output F goes to the renderer through ``bts_project_features`` (G = F . w_in[:, :C]^T, channels-last, one streaming HIP pass and
its fused backward).  SURVEY.md section 8 row f4 asked whether the decoder's last convolution should write G itself: rounds 2 - 3
built that (the composed weights W' = w_f . W_conv handed to MIOpen) and measured it 1.4 ms per step SLOWER than convolution +
projection passes (49.9 vs 48.6 ms, profiles/r03g); with the round-4 projection kernels (0.21 + 0.45 ms per step at bs 16) the
projection is 1.3 % of the step, and the route was removed.
"""
import warnings

import torch
import torch.nn.functional as F
from torch import nn


# ---------------------------------------------------------------------------------------------------------------
# ResNet (parameter names of torchvision.models.resnet, which monodepth2.py:86-101 instantiates)
# ---------------------------------------------------------------------------------------------------------------
class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + identity)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)      # stride on the 3x3 (torchvision's "v1.5")
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + identity)


_RESNETS = {18: (BasicBlock, [2, 2, 2, 2]), 34: (BasicBlock, [3, 4, 6, 3]), 50: (Bottleneck, [3, 4, 6, 3]),
            101: (Bottleneck, [3, 4, 23, 3]), 152: (Bottleneck, [3, 8, 36, 3])}


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], 2)
        self.layer3 = self._make_layer(block, 256, layers[2], 2)
        self.layer4 = self._make_layer(block, 512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)     # unused by the encoder; present so that checkpoints load strictly
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1), nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                       nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)


class ResnetEncoder(nn.Module):
    """monodepth2.py:71-107: five feature maps, strides 2..32."""

    def __init__(self, num_layers, pretrained=False, num_input_images=1):
        super().__init__()
        if num_layers not in _RESNETS:
            raise ValueError(f"{num_layers} is not a valid number of resnet layers")
        if num_input_images != 1:
            raise NotImplementedError("multi-image input is not used by BehindTheScenes")
        block, layers = _RESNETS[num_layers]
        self.encoder = ResNet(block, layers)
        # the classifier head is never evaluated (it exists so that torchvision / reference checkpoints load strictly): frozen, or
        # DistributedDataParallel would wait for its gradient in the second iteration ("Expected to have finished reduction ...")
        self.encoder.fc.requires_grad_(False)
        self.num_ch_enc = [64, 64, 128, 256, 512] if num_layers <= 34 else [64, 256, 512, 1024, 2048]
        # `pretrained`: the reference downloads ImageNet weights here (monodepth2.py:258 passes True unconditionally); there is no
        # network in this environment -- Monodepth2(pretrained_path=...) loads a torchvision ResNet state dict, cp_location a BTS
        # checkpoint (same keys); without either the constructor warns that the run starts from a random initialisation.

    def forward(self, input_image):
        e = self.encoder
        x = (input_image - 0.45) / 0.225
        f0 = e.relu(e.bn1(e.conv1(x)))
        f1 = e.layer1(e.maxpool(f0))
        f2 = e.layer2(f1)
        f3 = e.layer3(f2)
        f4 = e.layer4(f3)
        return [f0, f1, f2, f3, f4]


# ---------------------------------------------------------------------------------------------------------------
# decoder (models/common/model/layers.py:11-41, monodepth2.py:172-239)
# ---------------------------------------------------------------------------------------------------------------
class Conv3x3(nn.Module):
    def __init__(self, in_channels, out_channels, use_refl=True):
        super().__init__()
        self.pad = nn.ReflectionPad2d(1) if use_refl else nn.ZeroPad2d(1)
        self.conv = nn.Conv2d(int(in_channels), int(out_channels), 3)

    def forward(self, x):
        return self.conv(self.pad(x))


class ConvBlock(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = Conv3x3(in_channels, out_channels)
        self.nonlin = nn.ELU(inplace=True)

    def forward(self, x):
        return self.nonlin(self.conv(x))


class Decoder(nn.Module):
    def __init__(self, num_ch_enc, num_ch_dec=None, d_out=1, scales=range(4), use_skips=True):
        super().__init__()
        self.use_skips, self.num_ch_enc, self.d_out, self.scales = use_skips, list(num_ch_enc), d_out, list(scales)
        num_ch_dec = [128, 128, 256, 256, 512] if num_ch_dec is None else list(num_ch_dec)
        self.num_ch_dec = [max(d_out, c) for c in num_ch_dec]
        convs, keys = [], {}
        for i in range(4, -1, -1):
            cin = self.num_ch_enc[-1] if i == 4 else self.num_ch_dec[i + 1]
            keys[("upconv", i, 0)] = len(convs)
            convs.append(ConvBlock(cin, self.num_ch_dec[i]))
            cin = self.num_ch_dec[i] + (self.num_ch_enc[i - 1] if use_skips and i > 0 else 0)
            keys[("upconv", i, 1)] = len(convs)
            convs.append(ConvBlock(cin, self.num_ch_dec[i]))
        for s in self.scales:
            keys[("dispconv", s)] = len(convs)
            convs.append(Conv3x3(self.num_ch_dec[s], d_out))
        self.decoder_keys = keys
        self.decoder = nn.ModuleList(convs)

    # the layers that PRODUCE the renderer's scale-0 feature map -- upconv(0,0), the x2 upsampling,
    # upconv(0,1), dispconv(0) -- run as bts_conv3x3_fwd / _bwd (csrc/bts_conv.hip: reflection, upsampling and ELU are index arithmetic
    # and an epilogue of one kernel per layer instead of padded copies, layout changes and element-wise passes around a library
    # convolution: 23 ms of a 47.5 ms exp_kitti_360.yaml step through MIOpen, profiles/r05c/md2_tail_probe.txt) whenever the three are
    # 64 -> 64 (d_out = 64: every KITTI configuration) on a GPU; `fused_tail = False` keeps the module path for an A/B.
    fused_tail = True

    def tail_is_fused(self, x):
        """x = the (N, 64, H/2, W/2) input of upconv(0, 0).  The envelope of bts_conv3x3_* (conv_ok in csrc/bts_conv.hip): fp32 on the
        GPU, 64 -> 64 channels, every layer's input at least 4 x 4 (reflection padding reads two pixels inwards) and a single image's
        full-resolution map below 2^31 elements; fp32 weights, no autocast (Conv3x3Function computes in fp32 only).  Anything else
        keeps the nn.Module layers."""
        if not (self.fused_tail and x.is_cuda and x.dtype == torch.float32 and 0 in self.scales and self.num_ch_dec[0] == 64
                and self.num_ch_dec[1] == 64 and self.d_out == 64):
            return False
        h, w = x.shape[-2:]
        if h < 4 or w < 4 or (2 * h) * (2 * w) * 64 >= 2 ** 31 or torch.is_autocast_enabled():
            return False
        m0 = self.decoder[self.decoder_keys[("upconv", 0, 0)]].conv.conv
        return m0.weight.dtype == torch.float32

    def _conv(self, key, x_nhwc, up2=False, elu=False, out_nchw=False):
        from . import native
        m = self.decoder[self.decoder_keys[key]]
        conv = m.conv.conv if isinstance(m, ConvBlock) else m.conv
        return native.Conv3x3Function.apply(x_nhwc, conv.weight, conv.bias, up2, elu, out_nchw)

    def trunk(self, input_features):
        """The activations feeding the per-scale output convolutions: {scale: (N, num_ch_dec[scale], h, w)}."""
        feats, x = {}, input_features[-1]
        for i in range(4, -1, -1):
            if i == 0 and self.tail_is_fused(x):
                # x (N, 64, H/2, W/2) in channels_last memory IS (N, H/2, W/2, 64): upconv(0,0), then x2 + upconv(0,1) in one kernel
                y = self._conv(("upconv", 0, 0), x.permute(0, 2, 3, 1), elu=True)
                y = self._conv(("upconv", 0, 1), y, up2=True, elu=True)
                feats[0] = y.permute(0, 3, 1, 2)        # (N, 64, H, W) view over channels-last memory, like MIOpen's output
                break
            x = self.decoder[self.decoder_keys[("upconv", i, 0)]](x)
            x = [F.interpolate(x, scale_factor=(2, 2), mode="nearest")]
            if self.use_skips and i > 0:
                skip = input_features[i - 1]
                x[0] = x[0][:, :, :skip.shape[2], :skip.shape[3]]
                x.append(skip)
            x = self.decoder[self.decoder_keys[("upconv", i, 1)]](torch.cat(x, 1))
            if i in self.scales:
                feats[i] = x
        return feats

    def forward(self, input_features):
        feats = self.trunk(input_features)
        return {("disp", s): self.decoder[self.decoder_keys[("dispconv", s)]](feats[s]) for s in self.scales}


class Monodepth2(nn.Module):
    def __init__(self, resnet_layers=18, cp_location=None, freeze=False, num_ch_dec=None, d_out=128, scales=range(4), pretrained=True,
                 pretrained_path=None):
        super().__init__()
        self.encoder = ResnetEncoder(resnet_layers, pretrained, 1)
        if pretrained_path is not None:
            # torchvision's resnet state dict (what the reference downloads, monodepth2.py:258 -> models.resnetXX(pretrained=True))
            sd = torch.load(pretrained_path, map_location="cpu")
            self.encoder.encoder.load_state_dict(sd.get("state_dict", sd) if isinstance(sd, dict) else sd)
        elif pretrained and cp_location is None:
            warnings.warn("Monodepth2(pretrained=True): the reference starts from ImageNet ResNet weights (monodepth2.py:258); none were given "
                          "(`pretrained_path` / `cp_location`), so this encoder starts from a random initialisation and a from-scratch run "
                          "will not reproduce the reference's quality", stacklevel=2)
        self.num_ch_enc = self.encoder.num_ch_enc
        self.d_out, self.scales = d_out, list(scales)
        self.decoder = Decoder(num_ch_enc=self.num_ch_enc, d_out=d_out, num_ch_dec=num_ch_dec, scales=self.scales)
        self.num_ch_dec = self.decoder.num_ch_dec
        self.latent_size = d_out
        if cp_location is not None:
            self.load_state_dict(torch.load(cp_location, map_location="cpu")["model"])
        if freeze:
            for p in self.parameters(True):
                p.requires_grad = False

    def _trunk(self, x):
        x = (x * .5 + .5).contiguous(memory_format=torch.channels_last)     # MIOpen NHWC kernels; the outputs come out channels-last
        return self.decoder.trunk(self.encoder(x))

    def forward(self, x):
        """images (B, 3, H, W) in [-1, 1] -> [features (B, d_out, H / 2^s, W / 2^s) for s in scales]  (monodepth2.py:279-291)."""
        feats = self._trunk(x)
        dec = self.decoder
        # scale 0's output convolution writes channels-last like every other layer here: a (B, d_out, H, W) view over (B, H, W, d_out)
        # memory, which the hand-over reads as it is (ABI 8: bts_project_features_cl; the map's gradient comes back in the same format, so
        # neither direction has a layout pass -- SURVEY 8 row f4)
        return [dec._conv(("dispconv", 0), feats[0].permute(0, 2, 3, 1)).permute(0, 3, 1, 2) if (s == 0 and dec.tail_is_fused(feats[0])) else
                dec.decoder[dec.decoder_keys[("dispconv", s)]](feats[s]) for s in self.scales]

    @classmethod
    def from_conf(cls, conf, **kw):
        return cls(cp_location=conf.get("cp_location", None), freeze=conf.get("freeze", False), num_ch_dec=conf.get("num_ch_dec", None),
                   d_out=conf.get("d_out", 128), resnet_layers=conf.get("resnet_layers", 18), pretrained=conf.get("pretrained", True),
                   pretrained_path=conf.get("pretrained_path", None))
