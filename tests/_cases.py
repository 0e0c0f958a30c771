"""Loads the golden fixtures (tests/golden/*.npz, produced by the real reference) into oracle containers."""
import ast
import os

import numpy as np
import torch

from oracle import bts_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RENDER_CASES = ["kitti_train", "kitti_eval", "kitti_single", "re10k_train", "odd_cfg"]
GRAD_CASES = ["kitti_train", "re10k_train"]


def robust_ray_mask(st, rays, z_samp, margin=1e-4):
    """Rays none of whose samples sits within `margin` of a frustum border test (|x|,|y| == 1, z == EPS) in any view: on those
    the boolean `invalid` flags cannot flip under 1-ulp differences in the projection (SURVEY.md section 7 hazard iv).  Border
    pixels of every rendered frame project EXACTLY onto |x| = 1 or |y| = 1 of their own view, so they are never robust."""
    n = rays.shape[0]
    r = rays.reshape(-1, 8)
    pts = (r[:, None, :3] + z_samp.unsqueeze(2) * r[:, None, 3:6]).reshape(n, -1, 3)
    w2c = torch.cat((st.w2c_enc.unsqueeze(1), st.w2c_r), dim=1)
    Ks = torch.cat((st.K_enc.unsqueeze(1), st.K_r), dim=1)
    xy, z, _, _ = O.project(pts, w2c, Ks)
    near = ((xy.abs() - 1).abs() < margin).any(-1, keepdim=True) | ((z - O.EPS).abs() < margin)
    near = near.any(dim=1).reshape(r.shape[0], -1)     # (B, K)
    return ~near.any(dim=1)


class Case:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, f"{name}.npz"))
        self.name = name
        self.meta = ast.literal_eval(str(z["meta"]))
        self.t = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}
        m = self.meta
        self.cfg = O.FieldConfig(d_min=m["d_min"], d_max=m["d_max"], inv_z=m["inv_z"], code_mode=m["code_mode"],
                                 num_freqs=m["num_freqs"], freq_factor=m["freq_factor"], learn_empty=m["learn_empty"],
                                 empty_empty=m["empty_empty"])
        blocks = [(self.t[f"blk{i}_w0"], self.t[f"blk{i}_b0"], self.t[f"blk{i}_w1"], self.t[f"blk{i}_b1"])
                  for i in range(m["nb"])]
        self.mlp = O.MlpParams(self.t["w_in"], self.t["b_in"], blocks, self.t["w_out"], self.t["b_out"])
        self.scene = dict(images=self.t["images"], feat=self.t["feat"], projs=self.t["projs"], poses=self.t["poses"])
        self.state = O.make_state(self.scene, m["ids_render"], self.cfg, self.t.get("empty_feature"))
        self.rays = self.t["rays"]          # (n, B', 8)
        self.z_samp = self.t["z_samp"]      # (n*B', K)
        self.hard_cap = m["hard_cap"]

    def robust_ray_mask(self, margin=1e-4):
        return robust_ray_mask(self.state, self.rays, self.z_samp, margin)

    def well_conditioned_colour_mask(self, min_z=0.1):
        """(B, K, nv) True where the point's depth in that render view is >= min_z.  Closer to the camera plane the perspective
        divide amplifies fp32 rounding so much that the reference's own fp32 colour tap is off by > 1e-5 from an fp64 evaluation
        (measured: 1.6e-5 at z = 0.019, up to 2.3e-4 in the re10k fixture), so a 1e-5 per-sample tolerance is meaningless there."""
        st = self.state
        n = self.rays.shape[0]
        rays = self.rays.reshape(-1, 8)
        pts = (rays[:, None, :3] + self.z_samp.unsqueeze(2) * rays[:, None, 3:6]).reshape(n, -1, 3)
        _, z, _, _ = O.project(pts, st.w2c_r, st.K_r)                    # (n, nv, P, 1)
        K = self.z_samp.shape[1]
        return (z[..., 0] >= min_z).permute(0, 2, 1).reshape(rays.shape[0], K, -1)


class ProfileCase:
    """tests/golden/profile.npz: the reference's own get_pts / render_profile (scripts/inference_setup.py:84-97, 201-229, executed from
    the reference's source text by tests/golden/gen_golden_profile.py) on a small grid and a seeded scene."""

    def __init__(self):
        z = np.load(os.path.join(GOLDEN, "profile.npz"))
        self.t = {k: torch.from_numpy(z[k]) for k in z.files if not k.startswith("meta_")}
        self.meta = {k[5:]: z[k].tolist() for k in z.files if k.startswith("meta_")}
        self.cfg = O.FieldConfig(learn_empty=True)
        self.mlp = O.MlpParams(self.t["w_in"], self.t["b_in"], [], self.t["w_out"], self.t["b_out"])
        self.scene = dict(images=self.t["images"], feat=self.t["feat"], projs=self.t["projs"], poses=self.t["poses"])
        self.state = O.make_state(self.scene, self.meta["ids_render"], self.cfg, self.t["empty_feature"])

    def decided_columns(self, margin=2e-4):
        """(Z, X) bool: columns none of whose running sums (of the REFERENCE's densities, invalid -> 1) comes within `margin` of the
        threshold -- on the others a last-bit difference of one density may move the count by one level."""
        q = self.t["q_pts"]
        Y, Z, X, _ = q.shape
        a = self.t["sigma"].clone()
        a[self.t["invalid"].reshape(Y * Z * X, -1).any(-1)] = 1
        cs = torch.cumsum(a.reshape(Y, Z, X).double(), 0)
        return ((cs - self.meta["threshold"]).abs() > margin).all(0)
