"""Which samples change between launches of a (not shipped) gather-order build?  Renders the RE10K determinism case `runs` times with the
library named by BTS_RENDER_LIB and compares the per-sample pre-softplus densities (independent per sample: a corrupted gather block shows
up as exactly the samples it fed) with the first launch.
    BTS_ALLOW_LIB_OVERRIDE=1 BTS_RENDER_LIB=.../variants/libbts_fetchlate.so python tools/late_probe.py [runs] [case]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import behindthescenes_amd as bts
from behindthescenes_amd import native
from tests.test_gpu_determinism import _scene, CASES

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
case = sys.argv[2] if len(sys.argv) > 2 else "re10k_nv2"
n, v, H, W, C, Hd, nb, K, ids, n_rays, conf = CASES[case]
net, renderer, rays, z = _scene(bts, n, v, H, W, C, Hd, nb, K, ids, n_rays, seed=40 + len(case), **conf)
ft = net.native_field()
params = net.mlp_coarse.packed().detach()
ref = None
bad_launches, groups, lens = 0, collections.Counter(), collections.Counter()
with torch.no_grad():
    for r in range(runs):
        s = native.render_fwd(ft, params, rays, z, hard_alpha_cap=False, want_saved=True)["sigma_raw"]
        if ref is None:
            ref = s.clone()
            continue
        d = (s != ref)
        if d.any():
            bad_launches += 1
            for ray in d.any(1).nonzero().flatten().tolist():
                ks = d[ray].nonzero().flatten().tolist()
                lens[len(ks)] += 1
                groups[(ks[0] % 64) // 8] += 1
                if sum(lens.values()) <= 6:
                    print(f"  launch {r} ray {ray}: samples {ks}  |d| {float((s[ray] - ref[ray]).abs().max()):.3e}")
print(f"{os.path.basename(os.environ.get('BTS_RENDER_LIB', 'libbts_render.so'))} [{case}]: {bad_launches} of {runs - 1} launches differ from the first; "
      f"samples changed per ray {dict(lens)}; first changed sample's group of 8 (= DMA instruction j + 4 x point tile) {dict(groups)}")
