"""Worker of tests/test_gpu_ddp.py: one of WORLD_SIZE processes sharing cuda:0.  Wraps the REAL task (BTSNet + NeRFRenderer, i.e. the
RenderFunction / ProjectFunction autograd nodes and the packed-parameter cat) in DistributedDataParallel through
behindthescenes_amd.parallel.wrap_ddp and writes its gradients / sharded-render outputs for the parent to compare."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import behindthescenes_amd as bts  # noqa: E402
from behindthescenes_amd import parallel, synthetic as S  # noqa: E402

N_TOTAL, V, H, W, C, K = 4, 3, 48, 160, 64, 32


class SharedMaps(torch.nn.Module):
    """feature maps of ALL samples; a rank reads only its own rows (the stand-in for a CNN whose weights are shared)"""

    def __init__(self, feats):
        super().__init__()
        self.feats = torch.nn.ParameterList([torch.nn.Parameter(feats.clone())])
        self.latent_size, self.scales, self.rows = feats.shape[1], [0], slice(0, feats.shape[0])

    def forward(self, x):
        return [self.feats[0][self.rows]]


class Task(torch.nn.Module):
    def __init__(self, net, renderer):
        super().__init__()
        self.net, self.renderer = net, renderer

    def forward(self, images, projs, poses, rays, z, c_rgb):
        self.net.encode(images, projs, poses, ids_encoder=[0], ids_render=[1, 2])
        w, rgb, depth, *_ = self.renderer.composite(self.net, rays.reshape(-1, 8), z, sb=images.shape[0])
        return ((rgb * c_rgb).sum() + 0.05 * depth.sum()) / images.shape[0]


def build(scene):
    torch.manual_seed(0)
    net = bts.BTSNet(S.field_conf(C, 64, 0, H, W))
    S.init_mlp_(net.mlp_coarse, seed=7)
    net.encoder = SharedMaps(scene["feat"])
    return Task(net.cuda().train(), bts.NeRFRenderer(n_coarse=K, lindisp=True, hard_alpha_cap=True).cuda())


def main(out_dir):
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("BTS_TEST_BACKEND", "gloo")
    # gloo: every rank shares cuda:0 (RCCL refuses two ranks on one device); nccl (= RCCL): one rank per device, as a real run
    local = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or backend == "nccl":
        dist.init_process_group(backend, rank=rank, world_size=world, **(dict(device_id=dev) if backend == "nccl" else {}))
    scene = S.synthetic_scene(N_TOTAL, V, H, W, C, seed=3, intrinsics=S.K_KITTI360, smooth=True)
    g = torch.Generator().manual_seed(11)
    task = build(scene)
    sampler = bts.PatchRaySampler(ray_batch_size=512, z_near=3.0, z_far=80.0, patch_size=8)
    torch.manual_seed(5)
    images, projs, poses = scene["images"].cuda(), scene["projs"].cuda(), scene["poses"].cuda()
    rays, _ = sampler.sample(images[:, :1] * .5 + .5, poses[:, :1], projs[:, :1])           # (4, 512, 8), same draws on every rank
    z = task.renderer.sample_coarse(rays.reshape(-1, 8), torch.rand(N_TOTAL * 512, K, generator=g).cuda()).view(N_TOTAL, 512, K)
    c_rgb = torch.randn(N_TOTAL, 512, 6, generator=g).cuda()
    s, e = parallel.shard_range(N_TOTAL, rank, world)
    task.net.encoder.rows = slice(s, e)
    # backend nccl: a real DistributedDataParallel even at world size 1 (force) -- reducer, bucket views and the RCCL all-reduce then run
    # through RenderFunction / ProjectFunction on the single-GPU box exactly as they will on a node
    model = parallel.wrap_ddp(task, dev, force=backend == "nccl")
    if world > 1 or backend == "nccl":
        assert isinstance(model, torch.nn.parallel.DistributedDataParallel)
    if world == 1 and dist.is_initialized():    # one RCCL rank: exercise the collectives the multi-GPU paths use (barrier, all-reduce, all-gather)
        t = torch.ones(4, device=dev)
        dist.all_reduce(t), dist.barrier()
        assert parallel.all_gather_cat(t, 0, 4, 1) is t and float(t.sum()) == 4.0
    loss = model(images[s:e], projs[s:e], poses[s:e], rays[s:e], z[s:e].reshape(-1, K), c_rgb[s:e].reshape(-1, 6))
    loss.backward()
    m = task.net.mlp_coarse
    grads = dict(w_in=m.lin_in.weight.grad, b_in=m.lin_in.bias.grad, w_out=m.lin_out.weight.grad, b_out=m.lin_out.bias.grad,
                 feat=task.net.encoder.feats[0].grad)
    # ---- ray-sharded inference with the real kernel: every rank holds the whole field, renders a slice of the rays, all-gathers
    task.net.encoder.rows = slice(0, N_TOTAL)
    task.eval()
    wrapped = task.renderer.bind_parallel(task.net).eval()
    with torch.no_grad():
        task.net.encode(images, projs, poses, ids_encoder=[0], ids_render=[1, 2])
        orig = task.renderer.sample_coarse
        task.renderer.sample_coarse = lambda r, u=None: z.reshape(-1, K)[_rows(r, rays)]      # deterministic depths for both calls
        full = wrapped(rays, want_weights=True, want_alphas=True)["coarse"]
        shard = parallel.render_sharded(wrapped, rays, want_weights=True, want_alphas=True)["coarse"]
        task.renderer.sample_coarse = orig
    same = {k: bool(torch.equal(full[k], shard[k])) for k in full}
    torch.save(dict(grads={k: v.cpu() for k, v in grads.items()}, loss=float(loss), same=same, backend=backend), os.path.join(out_dir, f"rank{rank}_of{world}.pt"))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def _rows(r, rays):
    """indices (into the flattened full ray set) of the rays `r` (SB*b, 8) -- shards are contiguous slices along the ray axis"""
    sb, btot = rays.shape[:2]
    b = r.shape[0] // sb
    first = (rays[0, :, :].reshape(btot, 8) == r.view(sb, b, 8)[0, 0]).all(-1).nonzero()[0, 0]
    return (torch.arange(sb, device=r.device).view(-1, 1) * btot + first + torch.arange(b, device=r.device).view(1, -1)).reshape(-1)


if __name__ == "__main__":
    main(sys.argv[1])
