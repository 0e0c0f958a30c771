"""What does bts_project_features_bwd_tiles spend its time on?  The training shapes (KITTI-360: 16 x 64 x 192 x 640; KITTI-Raw: 8 frames), a
clustered tile pattern at several dirty fractions, each gradient alone and both; next to a plain fill of d_feat (the write floor).
    python tools/projbwd_probe.py [N] [cl]        (cl: the map in channels_last format -- ABI 8, bts_project_features_cl / _bwd_cl)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from behindthescenes_amd import native


def timed(fn, n=12):
    ts = []
    for i in range(n + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if i >= 2:
            ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    H, W = 192, 640
    spec = native.FieldSpec(C=64, d_hidden=64, n_blocks=0)
    g = torch.Generator(device="cuda").manual_seed(1)
    feat = torch.randn(N, 64, H, W, device="cuda", generator=g)
    if "cl" in sys.argv[2:]:
        feat = feat.contiguous(memory_format=torch.channels_last)
    mlp = torch.randn(spec.mlp_param_count(), device="cuda", generator=g) * 0.1
    tpi = native.proj_tile_count(spec, H, W)
    print(f"N={N} tiles/img={tpi} d_feat={feat.numel() * 4 / 1e6:.0f} MB  layout={'channels_last' if native.is_channels_last(feat) else 'nchw'}")
    print(f"fill(d_feat) {timed(lambda: torch.zeros_like(feat)):.4f} ms")
    for frac in (0.0, 0.1, 0.3, 1.0):
        tiles = torch.zeros(N, tpi, dtype=torch.uint8, device="cuda")
        if frac >= 1.0:
            tiles.fill_(1)
        elif frac > 0:
            # clusters: 2 x 2 neighbouring 16 x 4 blocks (an 8 x 8 patch's footprint and a bit), W / 16 blocks per block row
            n_cl = int(frac * N * tpi / 4)
            img = torch.randint(0, N, (n_cl,), device="cuda", generator=g)
            by = torch.randint(0, H // 4 - 1, (n_cl,), device="cuda", generator=g)
            bx = torch.randint(0, W // 16 - 1, (n_cl,), device="cuda", generator=g)
            for dy in range(2):
                for dx in range(2):
                    tiles[img, (by + dy) * (W // 16) + bx + dx] = 1
        d_proj = torch.zeros(N, H, W, 64, device="cuda")
        d_proj[tiles.bool()[:, native.proj_tile_map(H, W).cuda()]] = 0.01
        real = tiles.float().mean().item()
        t = timed(lambda: native.project_features(spec, feat, mlp, tiles=tiles))
        print(f"dirty {real:.3f}  forward (flagged tiles only)  {t:.4f} ms")
        for nf, nm in ((True, True), (True, False), (False, True)):
            t = timed(lambda: native.project_features_bwd(spec, feat, d_proj, mlp, need_feat=nf, need_mlp=nm, tiles=tiles, clear_after=False))
            print(f"dirty {real:.3f}  d_feat={int(nf)} d_mlp={int(nm)}  {t:.4f} ms")


main()
