"""A/B timing of differently built libraries on the BASELINE configs[1] kernel workload within ONE gpurun call (box-to-box spread is
+-4 %): every library renders the same frame `rounds` times in its own subprocess, interleaved over `passes` passes.
    python tools/lib_ab.py [--learn-empty] default noslp late_noslp ...      (names under behindthescenes_amd/variants/, or "default")
A name may carry modes after a colon: "default:jitter" = sample_coarse inside the kernel (BtsRenderArgs.jitter) instead of a z_samp
tensor, "default:nohint" = without the encoder-view hint (BtsFieldCfg.enc_render_view)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(rounds, learn_empty, modes=()):
    import torch
    import behindthescenes_amd as bts
    from behindthescenes_amd import native, synthetic as S
    H, W, K, V = 192, 640, 64, 2
    scene = S.synthetic_scene(1, V, H, W, 64, seed=1, intrinsics=S.K_KITTIRAW)
    net = S.build_net(scene, 64, 0, [0], learn_empty=learn_empty)
    ft = net.native_field()
    params = net.mlp_coarse.packed().detach()
    rays = bts.ImageRaySampler(3.0, 80.0, H, W).sample(None, scene["poses"].cuda(), scene["projs"].cuda())[0].reshape(-1, 8).contiguous()
    u = torch.rand(rays.shape[0], K, device="cuda")
    z = native.sample_coarse(rays, u, True)
    if "nohint" in modes:
        ft = native.FieldTensors(ft.spec, ft.proj_nhwc, ft.K_enc, ft.w2c_enc, ft.imgs_nhwc4, ft.K_r, ft.w2c_r, ft.empty_feature, enc_view=-1)
    kw = dict(jitter=u, lindisp=True) if "jitter" in modes else {}
    ts = []
    for r in range(rounds + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        native.render_fwd(ft, params, rays, None if kw else z, hard_alpha_cap=True, want_weights=True, want_alphas=True, want_invalid=True, **kw)
        e1.record()
        torch.cuda.synchronize()
        if r >= 2:
            ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f"{ts[len(ts) // 2]:.4f} {ts[0]:.4f}")


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    le = "--learn-empty" in sys.argv
    passes = int(os.environ.get("LIB_AB_PASSES", "3"))
    res = {a: [] for a in args}
    for _ in range(passes):
        for a in args:
            name, *modes = a.split(":")
            lib = os.path.join(ROOT, "behindthescenes_amd", "libbts_render.so" if name == "default" else f"variants/libbts_{name}.so")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "15"] + (["--learn-empty"] if le else []) + ["--modes=" + ",".join(modes)],
                               env=dict(os.environ, BTS_RENDER_LIB=lib, BTS_ALLOW_LIB_OVERRIDE="1", BTS_ALLOW_OLDER_ABI="1"), capture_output=True, text=True)
            if r.returncode:
                print(a, "FAILED", r.stderr[-500:])
                continue
            res[a].append(tuple(float(x) for x in r.stdout.split()[-2:]))
    for a, v in res.items():
        if v:
            print(f"{a:24s} median-of-medians {sorted(x[0] for x in v)[len(v) // 2]:.4f} ms   best {min(x[1] for x in v):.4f} ms   {v}")


if __name__ == "__main__":
    if "--child" in sys.argv:
        child(int(sys.argv[sys.argv.index("--child") + 1]), "--learn-empty" in sys.argv,
              tuple(m for a in sys.argv if a.startswith("--modes=") for m in a[8:].split(",") if m))
    else:
        main()
