"""The decoder tail's kernels alone: upconv(0,0), upconv(0,1) (x2 in front), dispconv(0) forward + backward at the
exp_kitti_360.yaml shapes, for rocprofv3 (`tools/profile.sh <tag> conv`) and for a quick timing:
    python tools/conv_probe.py [rounds = 5] [batch = 16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from behindthescenes_amd import native  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
H, W = 192, 640
g = torch.Generator().manual_seed(0)
x = torch.randn(B, H // 2, W // 2, 64, generator=g).cuda().requires_grad_(True)
ws = [(torch.randn(64, 64, 3, 3, generator=g) * (2.0 / 576) ** 0.5).cuda().requires_grad_(True) for _ in range(3)]
bs = [(torch.randn(64, generator=g) * 0.1).cuda().requires_grad_(True) for _ in range(3)]
gy = (torch.randn(B, 64, H, W, generator=g) / (B * 64 * H * W) ** 0.5).cuda()
ev, lay = [], []
for r in range(rounds + 2):
    e0, e1, e2, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(5))
    e0.record()
    y = native.Conv3x3Function.apply(x, ws[0], bs[0], False, True, False)
    ea.record()
    y = native.Conv3x3Function.apply(y, ws[1], bs[1], True, True, False)
    eb.record()
    f = native.Conv3x3Function.apply(y, ws[2], bs[2], False, False, True)
    e1.record()
    f.backward(gy)
    e2.record()
    torch.cuda.synchronize()
    if r >= 2:
        ev.append((e0.elapsed_time(e1), e1.elapsed_time(e2)))
        lay.append((e0.elapsed_time(ea), ea.elapsed_time(eb), eb.elapsed_time(e1)))
flop = 2.0 * 9 * 64 * 64 * B * (H * W * 2 + H * W / 4)
fw, bw = sum(a for a, _ in ev) / len(ev), sum(b for _, b in ev) / len(ev)
print(f"decoder tail, bs {B}, {H}x{W}: forward {fw:.3f} ms ({flop / fw / 1e9:.1f} TFLOP/s), backward {bw:.3f} ms ({2 * flop / bw / 1e9:.1f} TFLOP/s incl. the "
      f"elu' / transpose passes), {flop / 1e9:.0f} GFLOP forward")
print("forward per layer (incl. launch gaps): upconv(0,0) @ H/2 %.3f ms, upconv(0,1) x2 in front %.3f ms, dispconv(0) -> NCHW %.3f ms" % tuple(
    sorted(l[i] for l in lay)[len(lay) // 2] for i in range(3)))
