"""Debug helper (test infrastructure, not collected by pytest): renders the golden cases with two builds of the library and prints
where their outputs differ bitwise.   python tests/variant_diff.py <libA.so> <libB.so> [case ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(out, cases):
    import torch
    import behindthescenes_amd as bts
    from tests._cases import Case
    from tests._hip_helpers import net_from_case
    res = {}
    for name in cases:
        c = Case(name)
        net = net_from_case(c, "cuda")
        renderer = bts.NeRFRenderer(n_coarse=c.meta["K"], lindisp=True, hard_alpha_cap=c.hard_cap).cuda().eval()
        with torch.no_grad():
            o = renderer.composite(net, c.rays.reshape(-1, 8).cuda(), c.z_samp.cuda(), coarse=True, sb=c.rays.shape[0])
            ft = net.native_field()
            saved = bts.native.render_fwd(ft, net.mlp_coarse.packed().detach(), c.rays.reshape(-1, 8).cuda().contiguous(), c.z_samp.cuda().contiguous(),
                                          hard_alpha_cap=c.hard_cap, want_saved=True)
        res[name] = dict(w=o[0].cpu(), rgb=o[1].cpu(), depth=o[2].cpu(), a=o[3].cpu(), inv=o[4].cpu(), rs=o[6].cpu(), sigma_raw=saved["sigma_raw"].cpu(),
                         ref_a=c.t["out_alphas"], meta=dict(K=c.meta["K"], rays=tuple(c.rays.shape)))
    torch.save(res, out)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3:])
        sys.exit(0)
    import torch
    libs = sys.argv[1:3]
    cases = sys.argv[3:] or ["kitti_train", "kitti_eval", "kitti_single", "re10k_train", "odd_cfg"]
    outs = []
    for i, lib in enumerate(libs):
        f = f"/tmp/variant_diff_{i}.pt"
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", f] + cases, env=dict(os.environ, BTS_RENDER_LIB=os.path.abspath(lib), BTS_ALLOW_LIB_OVERRIDE="1"),
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(f))
    for name in cases:
        A, B = outs[0][name], outs[1][name]
        print(f"== {name} {A['meta']}")
        for k in ("sigma_raw", "a", "w", "rgb", "depth", "rs", "inv"):
            ne = A[k] != B[k]
            if ne.any():
                idx = ne.nonzero()
                d = (A[k].double() - B[k].double()).abs()
                rel = d / A[k].double().abs().clamp_min(1e-30)
                print(f"  {k}: {int(ne.sum())} of {ne.numel()} differ, max |d| {float(d.max()):.3e} (rel {float(rel[ne].max()):.3e}, median rel of differing {float(rel[ne].median()):.3e}); first {idx[:6].tolist()}")
                if k == "a":
                    ea, eb = (A[k] - A["ref_a"]).abs(), (B[k] - B["ref_a"]).abs()
                    print(f"     |alpha - golden|: A max {float(ea.max()):.3e} at {tuple(ea.argmax().item() // ea.shape[1:][0] for _ in [0])}, B max {float(eb.max()):.3e}; A>1e-5: {int((ea > 1e-5).sum())}, B>1e-5: {int((eb > 1e-5).sum())}")
                for i in idx[:4]:
                    t = tuple(i.tolist())
                    extra = f" ref {float(A['ref_a'][t]):.8g}" if k == "a" else ""
                    print(f"     at {t}: A {float(A[k][t]):.9g}  B {float(B[k][t]):.9g}{extra}")
