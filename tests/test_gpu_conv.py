"""GPU: the Monodepth2 decoder's tail as hand-written kernels (csrc/bts_conv.hip, bts_conv3x3_fwd / _bwd -- since
round 5 on the bf16 matrix pipe with every fp32 operand split into three exact bf16 terms) against
PyTorch's own ops on the same weights: ``ReflectionPad2d(1)`` + ``Conv2d(3 x 3)`` [+ ``ELU``] with ``F.interpolate(nearest, x2)`` in
front (models/common/model/layers.py:11-40, models/common/backbones/monodepth2.py:211-239).

Tolerances (the round-4 review's bar for this row): forward within 1e-5 absolute of an fp64 evaluation of the reference ops (outputs are
O(1)); gradients within 1e-4 of the largest entry of each tensor.  The backward has no atomics: two runs are bit-identical."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(x_nchw, w, b, up2, elu):
    if up2:
        x_nchw = F.interpolate(x_nchw, scale_factor=(2, 2), mode="nearest")
    y = F.conv2d(F.pad(x_nchw, (1, 1, 1, 1), mode="reflect"), w, b)
    return F.elu(y) if elu else y


CASES = [
    # N, Hs, Ws, up2, elu, out_nchw
    (2, 12, 70, False, True, False),       # ragged row end (70 = 64 + 6), ConvBlock
    (1, 8, 64, False, False, True),        # plain convolution writing NCHW (dispconv)
    (2, 6, 40, True, True, False),         # x2 upsampling in front (upconv(0,1)): output 12 x 80
    (1, 4, 4, False, True, False),         # the smallest frame: every pixel next to a border
    (3, 5, 131, False, False, False),      # odd sizes, three ragged tiles per row
    (1, 3, 33, True, False, True),         # up2 with odd source sizes, NCHW out
    # the tile geometry of the bf16 kernels (csrc/bts_conv.hip): 62 outputs per wave tile (60 for the data gradient behind an x2 upsampling),
    # 16 pixels per k-step of the weight gradient -- rows that end exactly on, one before and one after those boundaries
    (1, 5, 62, False, True, False),
    (1, 4, 63, False, False, True),
    (2, 4, 124, False, True, False),
    (1, 4, 125, False, False, False),
    (1, 4, 30, True, True, False),         # 60 wide behind the upsampling
    (1, 2, 31, True, False, True),         # 62 wide, the smallest height behind the upsampling
    (1, 3, 61, True, True, False),         # 122 = two data-gradient tiles + 2
    (1, 4, 16, False, False, False),
    (1, 4, 17, False, False, True),
    (2, 5, 33, False, False, False),
    (1, 4, 48, False, True, False),
]


@pytest.mark.parametrize("N,Hs,Ws,up2,elu,nchw", CASES)
def test_conv3x3_forward_and_backward_vs_torch(N, Hs, Ws, up2, elu, nchw):
    from behindthescenes_amd import native
    g = torch.Generator().manual_seed(N * 1000 + Hs * 10 + Ws)
    x = torch.randn(N, Hs, Ws, 64, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(64, 64, 3, 3, generator=g) * (2.0 / 576) ** 0.5).cuda().requires_grad_(True)
    b = (torch.randn(64, generator=g) * 0.1).cuda().requires_grad_(True)
    y = native.Conv3x3Function.apply(x, w, b, up2, elu, nchw)
    H, W = (2 * Hs, 2 * Ws) if up2 else (Hs, Ws)
    assert tuple(y.shape) == ((N, 64, H, W) if nchw else (N, H, W, 64))
    y_nchw = y if nchw else y.permute(0, 3, 1, 2)
    x64, w64, b64 = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    ref = _ref(x64.permute(0, 3, 1, 2), w64, b64, up2, elu)
    err = (y_nchw.detach().double() - ref.detach()).abs().max().item()
    assert err <= 1e-5, err          # (outputs of O(1) here: the review's absolute bar)
    gy = torch.randn(ref.shape, generator=g).cuda()
    ref.backward(gy.double())
    y.backward(gy if nchw else gy.permute(0, 2, 3, 1).contiguous())
    for name, got, want in (("d_x", x.grad, x64.grad), ("d_weight", w.grad, w64.grad), ("d_bias", b.grad, b64.grad)):
        top = want.abs().max().item()
        assert (got.double() - want).abs().max().item() <= 1e-4 * top, (name, (got.double() - want).abs().max().item(), top)
    # deterministic backward (no atomics): the same bits again
    first = [t.grad.clone() for t in (x, w, b)]
    for t in (x, w, b):
        t.grad = None
    y2 = native.Conv3x3Function.apply(x, w, b, up2, elu, nchw)
    assert torch.equal(y2, y)
    y2.backward(gy if nchw else gy.permute(0, 2, 3, 1).contiguous())
    assert all(torch.equal(a, t.grad) for a, t in zip(first, (x, w, b)))


def test_conv3x3_at_the_decoder_tail_size_vs_torch():
    """One full-size layer (2 x 192 x 640, the KITTI frame; every wave of the persistent grid takes several tiles) against torch's ops on
    the GPU, both measured against an fp64 evaluation: within 3e-6 of the largest output (576 products per output accumulated in fp32 in
    the MFMA's order: measured 1.1e-5 on outputs up to 7; torch's own fp32 convolution, which sums in blocks, 3e-6); gradients within
    1e-4 of the largest entry of the fp64 ones."""
    from behindthescenes_amd import native
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 192, 640, 64, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(64, 64, 3, 3, generator=g) * (2.0 / 576) ** 0.5).cuda().requires_grad_(True)
    b = (torch.randn(64, generator=g) * 0.1).cuda().requires_grad_(True)
    y = native.Conv3x3Function.apply(x, w, b, False, True, False)
    x64, w64, b64 = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    ref64 = _ref(x64.permute(0, 3, 1, 2), w64, b64, False, True)
    with torch.no_grad():
        ref32 = _ref(x.detach().permute(0, 3, 1, 2), w.detach(), b.detach(), False, True)
    e_hip = (y.detach().permute(0, 3, 1, 2).double() - ref64.detach()).abs().max().item()
    e_t32 = (ref32.double() - ref64.detach()).abs().max().item()
    assert e_hip <= 3e-6 * max(1.0, ref64.detach().abs().max().item()) and e_t32 <= e_hip + 1e-5, (e_hip, e_t32)
    gy = torch.randn(ref64.shape, generator=g).cuda() / ref64.numel() ** 0.5
    ref64.backward(gy.double()), y.backward(gy.permute(0, 2, 3, 1).contiguous())
    for got, want in ((x.grad, x64.grad), (w.grad, w64.grad), (b.grad, b64.grad)):
        assert (got.double() - want).abs().max().item() <= 1e-4 * want.abs().max().item()


def test_conv3x3_wide_dynamic_range_vs_fp64():
    """The bf16 three-term split is exact for every finite fp32 value (bf16 carries fp32's exponent: no range to scale, unlike an f16
    split): activations spread over twelve decades, each output within 2e-6 of the sum of the ABSOLUTE products feeding it (the fp32
    accumulator's share; an fp32-input convolution does no better), forward and weight gradient."""
    from behindthescenes_amd import native
    g = torch.Generator().manual_seed(11)
    N, H, W = 1, 8, 70
    x = (torch.randn(N, H, W, 64, generator=g) * 10.0 ** (torch.rand(N, H, W, 64, generator=g) * 12 - 6)).cuda().requires_grad_(True)
    w = (torch.randn(64, 64, 3, 3, generator=g) * 10.0 ** (torch.rand(64, 64, 3, 3, generator=g) * 4 - 3)).cuda().requires_grad_(True)
    y = native.Conv3x3Function.apply(x, w, None, False, False, False)
    x64, w64 = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    ref = _ref(x64.permute(0, 3, 1, 2), w64, None, False, False)
    mag = _ref(x64.detach().abs().permute(0, 3, 1, 2), w64.detach().abs(), None, False, False)
    assert ((y.detach().permute(0, 3, 1, 2).double() - ref.detach()).abs() <= 2e-6 * mag).all()
    gy = torch.randn(ref.shape, generator=g).cuda()
    ref.backward(gy.double()), y.backward(gy.permute(0, 2, 3, 1).contiguous())
    xa = x64.detach().abs().requires_grad_(True)
    wa = w64.detach().abs().requires_grad_(True)
    _ref(xa.permute(0, 3, 1, 2), wa, None, False, False).backward(gy.double().abs())
    assert ((w.grad.double() - w64.grad).abs() <= 2e-6 * wa.grad).all()
    assert ((x.grad.double() - x64.grad).abs() <= 2e-6 * xa.grad).all()
