"""Golden fixture for the occupancy profile FROM THE REAL REFERENCE.  Run in the build container only:

    python -B tests/golden/gen_golden_profile.py

`scripts/inference_setup.py` cannot be imported (hydra, cv2, matplotlib, a dataset on disk at module level), so the two functions
that ARE the profile -- `get_pts` (:84-97) and `render_profile` (:201-229) -- are cut out of the reference's source file as text and
executed unmodified, with `OUT_RES` set to a small grid and `device` to the CPU, against the reference's own BTSNet (imported through
oracle/ref_shim.py).  Nothing of the reference's source enters the repository; the outputs do: tests/golden/profile.npz pins
oracle.profile_points / oracle.occupancy_profile (tests/test_oracle_golden.py) and bts_occupancy_profile (tests/test_gpu_parity.py).
"""
import ast
import math
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np
import torch

from oracle import bts_oracle as O
from oracle.ref_shim import load_reference, REFERENCE_ROOT
from gen_golden import ref_conf, load_mlp_into, mlp_arrays

torch.set_num_threads(4)


def reference_functions(names, namespace):
    """exec the named top-level functions of scripts/inference_setup.py, unmodified, in `namespace`"""
    path = os.path.join(REFERENCE_ROOT, "scripts", "inference_setup.py")
    src = open(path).read()
    tree = ast.parse(src)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), namespace)
    missing = [n for n in names if n not in namespace]
    assert not missing, missing
    return namespace


class dotdict(dict):
    __getattr__ = dict.get


def main():
    ref = load_reference()
    cfg = O.FieldConfig(learn_empty=True)
    seed, n, v, H, W, C, Hd = 77, 1, 3, 48, 160, 64, 64
    g = torch.Generator().manual_seed(seed)
    scene = O.synthetic_scene(n, v, H, W, C, seed=seed, intrinsics=O.K_KITTI360, baseline=0.6, smooth=True)
    mlp = O.init_mlp(C + 39, Hd, 0, gen=g)
    mlp.b_out = torch.tensor([-1.5])        # densities of 0.1 - 1: the running sums cross the threshold inside the grid
    empty = torch.randn(C, generator=g)
    net = ref.make_net(ref_conf(cfg, 0, Hd), [scene["feat"]])
    load_mlp_into(net, mlp)
    with torch.no_grad():
        net.empty_feature.copy_(empty)
    net.eval()
    net.encode(scene["images"], scene["projs"], scene["poses"], ids_encoder=[0], ids_render=[0, 1])
    out_res = dotdict(X_RANGE=(-6, 6), Y_RANGE=(.0, .75), Z_RANGE=(16, 3), P_RES_ZX=(24, 40), P_RES_Y=64)
    ns = reference_functions(["get_pts", "render_profile"], dict(torch=torch, math=math, OUT_RES=out_res, device="cpu"))
    with torch.no_grad():
        q_pts = ns["get_pts"](out_res.X_RANGE, out_res.Y_RANGE, out_res.Z_RANGE, out_res.P_RES_ZX[1], out_res.P_RES_Y, out_res.P_RES_ZX[0])
        profile = ns["render_profile"](net, None)
        # the per-point values behind it (the reference's own field query), for diagnosis and for the threshold margin
        _, invalid, sigma = net.forward(q_pts.reshape(1, -1, 3))
    arrays = dict(q_pts=q_pts, profile=profile, sigma=sigma.reshape(-1), invalid=invalid[0].float(), empty_feature=empty,
                  images=scene["images"], feat=scene["feat"], projs=scene["projs"], poses=scene["poses"], **mlp_arrays(mlp))
    meta = dict(x_range=out_res.X_RANGE, y_range=out_res.Y_RANGE, z_range=out_res.Z_RANGE, x_res=out_res.P_RES_ZX[1], y_res=out_res.P_RES_Y,
                z_res=out_res.P_RES_ZX[0], ids_render=[0, 1], threshold=8.0)
    np.savez_compressed(os.path.join(HERE, "profile.npz"), **{k: t.numpy() for k, t in arrays.items()},
                        **{f"meta_{k}": np.asarray(v_) for k, v_ in meta.items()})
    print("profile.npz:", tuple(q_pts.shape), "profile", tuple(profile.shape), "values", sorted(set(np.round(profile.numpy().ravel(), 4)))[:8], "...",
          "invalid fraction %.3f" % float(invalid.float().mean()))


if __name__ == "__main__":
    main()
