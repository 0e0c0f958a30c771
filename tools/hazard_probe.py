"""Hazard bisection on the GPU box: renders fixed scenes with several differently scheduled builds of the library (one subprocess
per build, BTS_RENDER_LIB), checks run-to-run bit equality of every output, and diffs each build against the default one.  The
arithmetic of every variant is the same expression tree (-ffp-contract=off), so any bit difference is a hazard or a miscompile.

    python tools/hazard_probe.py                      # all libraries under behindthescenes_amd/variants/ + the default
    python tools/hazard_probe.py --child <lib> <out>  # (internal) render with one library and dump the outputs
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SCENES = {  # name -> (n, v, H, W, C, Hd, nb, K, ids_render, n_rays, seed)
    "arbiter_nv1": (1, 2, 192, 640, 64, 64, 0, 64, [0], 30000, 31),
    "nv2": (1, 3, 192, 640, 64, 64, 0, 64, [1, 2], 20000, 32),
    "k32_nv1": (2, 2, 96, 320, 64, 64, 0, 32, [1], 8192, 33),
    "re10k_nv2": (2, 3, 64, 96, 32, 32, 1, 48, [1, 2], 4096, 34),
}
REPS = int(os.environ.get("HAZARD_REPS", "4"))
ONLY = [t for t in os.environ.get("HAZARD_ONLY", "").split(",") if t]


def child(lib, out):
    import numpy as np
    import torch
    import behindthescenes_amd as bts
    from behindthescenes_amd import _lib, synthetic as S
    assert os.path.samefile(_lib.LIB_PATH, lib), (_lib.LIB_PATH, lib)
    _lib.load()
    res = {}
    for name, (n, v, H, W, C, Hd, nb, K, ids, n_rays, seed) in SCENES.items():
        g = torch.Generator().manual_seed(seed)
        scene = S.synthetic_scene(n, v, H, W, C, seed=seed, intrinsics=S.K_KITTIRAW if C == 64 else S.K_RE10K, smooth=True)
        conf = dict(z_near=1.0, z_far=100.0, code_mode="distance") if C == 32 else {}
        net = S.build_net(scene, d_hidden=Hd, n_blocks=nb, ids_render=ids, device="cuda", mlp_seed=seed, **conf)
        renderer = bts.NeRFRenderer(n_coarse=K, lindisp=True, hard_alpha_cap=(C == 64)).cuda().eval()
        zn, zf = (1.0, 100.0) if C == 32 else (3.0, 80.0)
        rays = bts.ImageRaySampler(zn, zf, H, W).sample(None, scene["poses"].cuda(), scene["projs"].cuda())[0]
        idx = torch.randperm(rays.shape[1], generator=g)[:n_rays].sort().values.cuda()
        rays = rays[:, idx].contiguous().reshape(-1, 8)
        u = torch.rand(rays.shape[0], K, generator=g).cuda()
        z = renderer.sample_coarse(rays, u)
        outs = []
        with torch.no_grad():
            for _ in range(REPS):
                w, rgb, depth, a, inv, _, rs = renderer.composite(net, rays, z, sb=n)
                outs.append([t.clone() for t in (w, rgb, depth, a, inv, rs)])
        torch.cuda.synchronize()
        det = all(torch.equal(x, y) for o in outs[1:] for x, y in zip(outs[0], o))
        res[name] = dict(deterministic=det)
        np.savez(os.path.join(out, f"{name}.npz"), **{k: t.cpu().numpy() for k, t in zip(("w", "rgb", "depth", "a", "inv", "rs"), outs[0])})
    json.dump(res, open(os.path.join(out, "det.json"), "w"))


def main():
    import numpy as np
    vdir = os.path.join(ROOT, "behindthescenes_amd", "variants")
    libs = {"default": os.path.join(ROOT, "behindthescenes_amd", "libbts_render.so")}
    for f in sorted(os.listdir(vdir)) if os.path.isdir(vdir) else []:
        if f.endswith(".so"):
            libs[f[len("libbts_"):-3]] = os.path.join(vdir, f)
    if ONLY:
        libs = {k: v for k, v in libs.items() if k in ONLY or k == "default"}
    tmp = os.environ.get("TMPDIR", "/tmp")
    report = {}
    for tag, lib in libs.items():
        out = os.path.join(tmp, f"hazard_{tag}")
        os.makedirs(out, exist_ok=True)
        env = dict(os.environ, BTS_RENDER_LIB=lib, BTS_ALLOW_LIB_OVERRIDE="1")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib, out], env=env, capture_output=True, text=True)
        if r.returncode != 0:
            report[tag] = dict(error=r.stderr[-2000:])
            continue
        report[tag] = dict(det=json.load(open(os.path.join(out, "det.json"))))
    ref = os.path.join(tmp, "hazard_default")
    for tag in libs:
        if tag == "default" or "error" in report[tag]:
            continue
        diff = {}
        for name in SCENES:
            a, b = np.load(os.path.join(ref, f"{name}.npz")), np.load(os.path.join(tmp, f"hazard_{tag}", f"{name}.npz"))
            d = {}
            for k in a.files:
                ne = a[k].view(np.uint32) != b[k].view(np.uint32)
                cnt = int(ne.sum())
                d[k] = dict(mismatch=cnt, of=int(ne.size), max_abs=float(np.abs(a[k] - b[k]).max()) if cnt else 0.0)
                if cnt and k in ("w", "a") and a[k].ndim == 2:
                    d[k]["sample_hist"] = ne.sum(axis=0).tolist()
                if cnt and k in ("rgb", "rs", "w"):
                    where = np.argwhere(ne)
                    rows = np.unique(where[:, 0])
                    d[k]["rays"] = int(rows.size)
                    d[k]["first_rays"] = rows[:24].tolist()
                    d[k]["ray_mod64_hist"] = np.bincount(rows % 64, minlength=64).tolist()
                    if k == "rs":
                        d[k]["sample_hist"] = np.bincount(where[:, 1], minlength=a[k].shape[1]).tolist()
                        ex = where[:6]
                        d[k]["examples"] = [dict(at=e.tolist(), ref=float(a[k][tuple(e)]), got=float(b[k][tuple(e)])) for e in ex]
            diff[name] = d
        report[tag]["vs_default"] = diff
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(report, open(os.path.join(ROOT, "gpurun_out", "hazard_probe.json"), "w"), indent=1)
    for tag, r in report.items():
        if "error" in r:
            print(tag, "ERROR", r["error"][-400:])
            continue
        print(tag, "deterministic:", {k: v["deterministic"] for k, v in r["det"].items()})
        for name, d in r.get("vs_default", {}).items():
            print("   ", name, {k: (v["mismatch"], v["max_abs"]) for k, v in d.items() if v["mismatch"]})
            for k in ("w", "a"):
                if d.get(k, {}).get("sample_hist"):
                    print("       ", k, "mismatches per sample index:", d[k]["sample_hist"])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3])
    else:
        main()
