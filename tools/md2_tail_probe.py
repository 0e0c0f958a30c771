"""SURVEY 8 row f4, the evidence: what do the Monodepth2 decoder's LAST convolutions cost per training step?

    python tools/md2_tail_probe.py [batch = 16] [H = 192] [W = 640]

With d_out = 64 the decoder's channel widths are clamped to >= 64 (monodepth2.py:189-206), so the tail that produces the renderer's scale-0
feature map is   upconv(0,0): ConvBlock 64 -> 64 @ H/2 x W/2   ->  nearest x2  ->  upconv(0,1): ConvBlock 64 -> 64 @ H x W  ->
dispconv(0): Conv3x3 64 -> 64 @ H x W   (reflection pad 1, 3x3, ELU inside the ConvBlocks; models/common/model/layers.py:11-40).
Each layer is run ALONE (forward + backward, channels-last like behindthescenes_amd.monodepth2, MIOpen's kernels) under torch.profiler:
the device time of every kernel it launches, split into forward / backward by the autograd phase, against the layer's FLOP
(2 x 9 x Cin x Cout per output pixel; backward = data + weight gradient = 2 x forward) at the fp32 peak of 157.3 TFLOP/s."""
import sys

import torch
import torch.nn.functional as F
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from behindthescenes_amd.monodepth2 import Conv3x3, ConvBlock  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H = int(sys.argv[2]) if len(sys.argv) > 2 else 192
W = int(sys.argv[3]) if len(sys.argv) > 3 else 640
dev = torch.device("cuda")
torch.manual_seed(0)
LAYERS = [("upconv(0,0) ConvBlock 64->64", ConvBlock(64, 64), (H // 2, W // 2)),
          ("upsample x2 (nearest)", None, (H // 2, W // 2)),
          ("upconv(0,1) ConvBlock 64->64", ConvBlock(64, 64), (H, W)),
          ("dispconv(0) Conv3x3 64->64", Conv3x3(64, 64), (H, W))]
PEAK = 157.3e12


def dev_us(e):
    return getattr(e, "device_time_total", None) or getattr(e, "cuda_time_total", 0.0)


print(f"# batch {B}, {H}x{W}, channels-last fp32, one layer at a time, 5 profiled iterations after 3 warm-ups")
print(f"{'layer':34s} {'fwd ms':>8s} {'bwd ms':>8s} {'GFLOP f/b':>12s} {'fwd TF':>7s} {'bwd TF':>7s}   kernels (ms per iteration)")
tot_f = tot_b = 0.0
for name, mod, (h, w) in LAYERS:
    x = torch.randn(B, 64, h, w, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    if mod is not None:
        mod = mod.to(dev).to(memory_format=torch.channels_last)
        fn = mod
        flop = 2.0 * 9 * 64 * 64 * h * w * B
    else:
        def fn(t):
            return F.interpolate(t, scale_factor=(2, 2), mode="nearest")
        flop = 0.0
    for _ in range(3):
        y = fn(x)
        y.backward(torch.ones_like(y))
    g = torch.randn_like(fn(x).detach())
    torch.cuda.synchronize()
    it = 5
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof_f:
        for _ in range(it):
            y = fn(x)
        torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof_b:
        for _ in range(it):
            y = fn(x)
            y.backward(g)
        torch.cuda.synchronize()
    kf = {e.key: dev_us(e) / it / 1e3 for e in prof_f.key_averages() if dev_us(e) > 0 and e.device_type.name != "CPU"}
    kb_all = {e.key: dev_us(e) / it / 1e3 for e in prof_b.key_averages() if dev_us(e) > 0 and e.device_type.name != "CPU"}
    fwd = sum(kf.values())
    bwd = sum(kb_all.values()) - fwd
    tot_f, tot_b = tot_f + fwd, tot_b + bwd
    top = sorted(((k, v - kf.get(k, 0.0)) for k, v in kb_all.items()), key=lambda kv: -kv[1])[:4]
    desc = "; ".join(f"{k[:48]} {v:.3f}" for k, v in sorted(kf.items(), key=lambda kv: -kv[1])[:3]) + "  ||  " + "; ".join(f"{k[:48]} {v:.3f}" for k, v in top)
    tf_f = flop / (fwd * 1e-3) / 1e12 if flop and fwd else 0.0
    tf_b = 2 * flop / (bwd * 1e-3) / 1e12 if flop and bwd else 0.0
    print(f"{name:34s} {fwd:8.3f} {bwd:8.3f} {flop / 1e9:6.1f}/{2 * flop / 1e9:5.1f} {tf_f:7.1f} {tf_b:7.1f}   {desc}")
print(f"{'tail total':34s} {tot_f:8.3f} {tot_b:8.3f}   = {tot_f + tot_b:.3f} ms per step of the decoder tail (fwd + bwd), fp32 peak {PEAK / 1e12:.1f} TF")
