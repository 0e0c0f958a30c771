"""Ray samplers with the reference's constructors and ``sample`` / ``reconstruct`` contracts
(models/bts/model/ray_sampler.py:7-321).  Ray generation (gen_rays + unproj_map, models/common/util/util.py:113-149,
244-273) runs in the HIP kernel ``bts_gen_rays`` -- one launch for all n*v frames instead of a Python loop over n; the
random patch / pixel choice keeps torch's CPU generator semantics of the reference (``torch.randint`` on the CPU)."""
import torch

from . import native


def gen_rays(poses, width, height, z_near, z_far, focal=None, c=None, norm_dir=True, projs=None):
    """(V, H, W, 8) rays for poses (V,4,4); intrinsics either as ``projs`` (V,3,3) or focal (V,2) + c (V,2)."""
    V = poses.shape[0]
    if projs is None:
        projs = torch.zeros((V, 3, 3), device=poses.device, dtype=torch.float32)
        projs[:, 0, 0], projs[:, 1, 1] = focal[:, 0], focal[:, 1]
        projs[:, 0, 2], projs[:, 1, 2] = c[:, 0], c[:, 1]
        projs[:, 2, 2] = 1.0
    return native.gen_rays(poses.float().contiguous(), projs.float().contiguous(), height, width, float(z_near), float(z_far),
                           norm_dir)


def _all_rays(poses, projs, h, w, z_near, z_far, norm_dir=True):
    n, v = poses.shape[:2]
    rays = native.gen_rays(poses.reshape(n * v, 4, 4).float().contiguous(), projs.reshape(n * v, 3, 3).float().contiguous(),
                           h, w, float(z_near), float(z_far), norm_dir)
    return rays.view(n, v, h, w, 8)


class RaySampler:
    def sample(self, images, poses, projs):
        raise NotImplementedError

    def reconstruct(self, render_dict):
        raise NotImplementedError

    @staticmethod
    def _reshape_all(render_dict, lead, channels, rgb_gt_shape=None):
        """Shared body of the three reconstruct() variants: view every per-ray tensor as lead + (...)."""
        for key in ("coarse", "fine"):
            part = render_dict[key]
            n = part["rgb"].shape[0]
            v = part["rgb"].shape[-1] // channels
            part["rgb"] = part["rgb"].view(n, *lead, v, channels)
            part["depth"] = part["depth"].view(n, *lead)
            if "weights" not in part:
                # lean training outputs (NeRFRenderer.lean_training_outputs): the per-sample tensors never left the kernel, the
                # loss' invalid-ray mask comes as per-ray, per-view reductions
                for k in ("invalid_wsum", "invalid_any"):
                    part[k] = part[k].view(n, *lead, v)
                render_dict[key] = part
                continue
            K = part["weights"].shape[-1]
            part["weights"] = part["weights"].view(n, *lead, K)
            part["invalid"] = part["invalid"].view(n, *lead, K, v)
            if "alphas" in part:
                part["alphas"] = part["alphas"].view(n, *lead, K)
            if "z_samps" in part:
                part["z_samps"] = part["z_samps"].view(n, *lead, K)
            if "rgb_samps" in part:
                part["rgb_samps"] = part["rgb_samps"].view(n, *lead, K, v, channels)
            render_dict[key] = part
        if "rgb_gt" in render_dict and render_dict["rgb_gt"] is not None:
            g = render_dict["rgb_gt"]
            render_dict["rgb_gt"] = g.view(g.shape[0], *lead, channels)
        return render_dict


class RandomRaySampler(RaySampler):
    def __init__(self, ray_batch_size, z_near, z_far, channels=3):
        self.ray_batch_size, self.z_near, self.z_far, self.channels = ray_batch_size, z_near, z_far, channels

    def sample(self, images, poses, projs):
        n, v, c, h, w = images.shape
        rays = _all_rays(poses, projs, h, w, self.z_near, self.z_far).view(n, -1, 8)
        gt = images.permute(0, 1, 3, 4, 2).reshape(n, -1, self.channels)
        idx = torch.stack([torch.randint(0, v * h * w, (self.ray_batch_size,)) for _ in range(n)]).to(rays.device)
        return (torch.gather(rays, 1, idx.unsqueeze(-1).expand(-1, -1, 8)),
                torch.gather(gt, 1, idx.unsqueeze(-1).expand(-1, -1, self.channels)))

    def reconstruct(self, render_dict, channels=None):
        channels = self.channels if channels is None else channels
        return self._reshape_all(render_dict, (render_dict["coarse"]["rgb"].shape[1],), channels)


class PatchRaySampler(RaySampler):
    def __init__(self, ray_batch_size, z_near, z_far, patch_size, channels=3):
        self.ray_batch_size, self.z_near, self.z_far = ray_batch_size, z_near, z_far
        if isinstance(patch_size, int):
            self.patch_size_x, self.patch_size_y = patch_size, patch_size
        elif hasattr(patch_size, "__len__") and len(patch_size) == 2:
            self.patch_size_y, self.patch_size_x = patch_size[0], patch_size[1]
        else:
            raise ValueError("Invalid format for patch size")
        self.channels = channels
        assert (ray_batch_size % (self.patch_size_x * self.patch_size_y)) == 0
        self._patch_count = self.ray_batch_size // (self.patch_size_x * self.patch_size_y)

    def draw_patches(self, n, v, h, w, rows=None):
        """The reference's CPU RNG draws, in its order (ray_sampler.py:141-143): per sample v, y, x vectors.  ``rows``: 3 n one-dimensional
        integer CPU tensors (v, y, x of sample 0, then of sample 1, ...) to draw into instead -- ``t.random_(0, hi)`` is what
        ``torch.randint(0, hi, ...)`` runs underneath and consumes the generator identically for int32 and int64 (tests/test_protocol_cpu.py)."""
        P = self._patch_count
        if rows is not None:
            hy, hx = h - self.patch_size_y, w - self.patch_size_x
            for i in range(n):
                rows[3 * i].random_(0, v), rows[3 * i + 1].random_(0, hy), rows[3 * i + 2].random_(0, hx)
            return None
        pv, py, px = [], [], []
        for _ in range(n):
            pv.append(torch.randint(0, v, (P,)))
            py.append(torch.randint(0, h - self.patch_size_y, (P,)))
            px.append(torch.randint(0, w - self.patch_size_x, (P,)))
        return torch.stack(pv), torch.stack(py), torch.stack(px)

    def sample(self, images, poses, projs, patches=None):
        """Rays and ground-truth colours of ``ray_batch_size`` / (ph*pw) random patches per sample (ray_sampler.py:125-162).  The patch
        coordinates come from the CPU RNG in the reference's order; one HIP pass (bts_patch_rays) then produces the rays of exactly
        those pixels and gathers their colours -- the reference builds all v*H*W rays per sample and slices them in a Python loop."""
        n, v, c, h, w = images.shape
        dev = images.device
        pv, py, px = self.draw_patches(n, v, h, w) if patches is None else patches
        if patches is not None:   # the draws live on the host anyway: out-of-frame patches would read out of bounds on the device
            for t, hi, nme in ((pv, v, "view"), (py, h - self.patch_size_y + 1, "y0"), (px, w - self.patch_size_x + 1, "x0")):
                if t.numel() and (int(t.min()) < 0 or int(t.max()) >= hi):
                    raise ValueError(f"patch {nme} outside [0, {hi})")
        # one small host-to-device copy, from pinned memory and asynchronous: a pageable source would make the host wait for everything
        # the stream still holds (the previous step's backward) before the copy even starts
        idx = torch.stack((pv, py, px)).to(torch.int32)
        idx = idx.pin_memory().to(dev, non_blocking=True) if dev.type == "cuda" else idx.to(dev)
        all_rays, all_rgb_gt = native.patch_rays(poses.detach().float().contiguous(), projs.detach().float().contiguous(),
                                                 images.detach().float().contiguous(), idx[0], idx[1], idx[2],
                                                 self.patch_size_y, self.patch_size_x, self.z_near, self.z_far)
        return all_rays, all_rgb_gt

    def reconstruct(self, render_dict, channels=None):
        channels = self.channels if channels is None else channels
        return self._reshape_all(render_dict, (self._patch_count, self.patch_size_y, self.patch_size_x), channels)


class ImageRaySampler(RaySampler):
    def __init__(self, z_near, z_far, height=None, width=None, channels=3, norm_dir=True):
        self.z_near, self.z_far, self.height, self.width = z_near, z_far, height, width
        self.channels, self.norm_dir = channels, norm_dir

    def sample(self, images, poses, projs):
        n, v = poses.shape[:2]
        if self.height is None:
            self.height, self.width = images.shape[-2:]
        all_rays = _all_rays(poses, projs, self.height, self.width, self.z_near, self.z_far, self.norm_dir).view(n, -1, 8)
        all_rgb_gt = None
        if images is not None:
            all_rgb_gt = images.reshape(n, -1, self.channels, self.height, self.width).permute(0, 1, 3, 4, 2).reshape(n, -1, self.channels)
        return all_rays, all_rgb_gt

    def reconstruct(self, render_dict, channels=None):
        channels = self.channels if channels is None else channels
        n_pts = render_dict["coarse"]["rgb"].shape[1]
        v_in = n_pts // (self.height * self.width)
        return self._reshape_all(render_dict, (v_in, self.height, self.width), channels)
