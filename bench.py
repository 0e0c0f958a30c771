"""bench.py -- rendered rays/sec of the fused HIP renderer on BASELINE.json's configs[1]:
KITTI eval_depth forward pass, bs=1, 192x640 frames, 64 samples/ray, both stereo views' rays rendered from the single
encoder view (245 760 rays, 15.7 M field queries per step), fp32.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = what the reference's evaluator does per frame after the CNN (models/bts/evaluator.py:60-79): hand the feature
map / frames over to the renderer (`encode` without a CNN: layout kernels only), draw the stratified jitter, render all
rays (`renderer(all_rays, want_weights=True, want_alphas=True)`), `reconstruct`, `distance_to_z`.  Inputs are synthetic and
resident in HBM before the timed region.  N > 1: every rank renders its own frame (frames are independent -> weak scaling,
no collective on the data path); value = total rays of all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (bts::render_kernel_p = bts_render_fwd), timed live with HIP
events on the launch stream inside the timed region; `roofline.traffic` is the HBM byte count of the committed rocprofv3 PMC passes
of the same workload (profiles/<round>/traffic.json, written by tools/profile.sh), null when absent; `cpu_baseline` times the CPU
oracle port on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_POINT = 2 * (103 * 64 + 64)          # 13 312 FLOP per field query (KITTI MLP)
PEAK_FP32_MATRIX_TFLOPS = 157.3               # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / fp32 vector peak
H, W, K, C, HD, V = 192, 640, 64, 64, 64, 2


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=("eval", "train"), default="eval",
                    help="eval (default): BASELINE configs[1], the headline line.  train: configs[2] (exp_kitti_360.yaml shapes), the "
                         "renderer's share of a training step, forward + backward")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gpu-eager-baseline", action="store_true",
                    help="also time the oracle's torch ops eagerly on the GPU (the reference's own code path on this device) -> "
                         "cpu_baseline.gpu_eager_port; off by default")
    ap.add_argument("--cpu-rows", type=int, default=24, help="image rows of view 0 rendered by the CPU oracle sample")
    return ap.parse_args()


def cpu_baseline(scene, net, rows, device="cpu"):
    """The oracle ("port" of the reference algorithm, same torch CPU ops) on a bounded sample: `rows` full image rows of
    both views (rows*640*2 rays, K=64), best of 2 after one warm-up.  The only place bench.py touches ``oracle/``."""
    from oracle import bts_oracle as O
    cfg = O.FieldConfig()                       # z in [3, 80], inv_z, code_mode z (eval_depth.yaml)
    m = net.mlp_coarse
    mlp = O.MlpParams(w_in=m.lin_in.weight.detach().cpu().clone(), b_in=m.lin_in.bias.detach().cpu().clone(),
                      w_out=m.lin_out.weight.detach().cpu().clone(), b_out=m.lin_out.bias.detach().cpu().clone())
    rays = O.image_rays(scene["poses"], scene["projs"], H, W, cfg.d_min, cfg.d_max).view(1, V, H, W, 8)
    r0 = (H - rows) // 2
    rays = rays[:, :, r0:r0 + rows].reshape(1, -1, 8).contiguous()
    g = torch.Generator().manual_seed(1)
    u = torch.rand(rays.shape[1], K, generator=g)
    st = O.make_state(scene, [0], cfg)
    if device != "cpu":   # the same torch ops, eagerly, on the GPU: what the reference's own code path costs on this device
        rays, u = rays.to(device), u.to(device)
        st = O.FieldState(*[None if t is None else t.to(device) for t in (st.feat, st.K_enc, st.w2c_enc, st.imgs, st.K_r, st.w2c_r, st.empty_feature)])
        mlp = O.MlpParams(mlp.w_in.to(device), mlp.b_in.to(device), [], mlp.w_out.to(device), mlp.b_out.to(device))
    best = float("inf")
    with torch.no_grad():
        for i in range(3):
            t0 = time.perf_counter()
            z = O.sample_coarse(rays.reshape(-1, 8), K, True, u)
            O.composite(rays.reshape(-1, 8), z, 1, st, mlp, cfg, hard_alpha_cap=True)
            if device != "cpu":
                torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if i > 0:
                best = min(best, dt)
    n_rays = rays.shape[1]
    if device != "cpu":
        return dict(value=n_rays / best, unit="rays/s", kind="port", device="MI355X, PyTorch-ROCm eager (the oracle's torch ops on cuda:0)",
                    sample=f"{rows} rows x {W} px x {V} views = {n_rays} rays x {K} samples, renderer only, best of 2")
    return dict(value=n_rays / best, unit="rays/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{rows} rows x {W} px x {V} views = {n_rays} rays x {K} samples, renderer only, best of 2")


def train_workload(args, world, rank, dev):
    """BASELINE configs[2] (configs/exp_kitti_360.yaml): bs 16 per GPU, 8 frames per sample (4 loss + 4 render views), 4096 patch rays
    (64 patches of 8x8) per sample, 64 samples per ray.  One step = the renderer's share of `trainer.py:208-259` + backward: encode
    hand-over (no CNN), PatchRaySampler.sample, G = project(F), render with saved activations and every output the trainer asks for
    (weights, alphas, rgb_samps), reconstruct, the photometric loss (l1+ssim, weight-guided invalid mask, edge-aware smoothness: one
    HIP pass incl. its gradient), backward through bts_render_bwd and bts_project_features_bwd (MLP and feature-map gradients), and
    under N > 1 the all-reduce of the MLP gradient (the only exchange of the path; the CNN's DDP bucket is not part of it)."""
    import behindthescenes_amd as bts
    from behindthescenes_amd import native, parallel, synthetic as S
    n, Vt, Kt, NV = 16, 8, 64, 4
    scene = S.synthetic_scene(n, Vt, H, W, C, seed=2000 + rank, intrinsics=S.K_KITTI360, baseline=0.6, smooth=True)
    net = bts.BTSNet(S.field_conf(C, HD, 0, H, W))
    S.init_mlp_(net.mlp_coarse, seed=7)
    net.encoder = bts.FeatureMapEncoder((H, W), C, num_views=n)
    S.set_feature_map(net, scene["feat"])
    net = net.to(dev).train()
    renderer = bts.NeRFRenderer.from_conf(dict(n_coarse=Kt, lindisp=True, hard_alpha_cap=True)).to(dev).train()
    sampler = bts.PatchRaySampler(ray_batch_size=4096, z_near=3.0, z_far=80.0, patch_size=8)
    images, projs, poses = scene["images"].to(dev), scene["projs"].to(dev), scene["poses"].to(dev)
    ids_loss, ids_render = [0, 1, 2, 3], [4, 5, 6, 7]
    kern = {"fwd": [], "bwd": []}
    orig_fwd, orig_bwd = native.render_fwd, native.render_bwd

    def timed(fn, key):
        def f(*a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            kern[key].append((e0, e1))
            return out
        return f

    wrapped = renderer.bind_parallel(net).train()
    crit = bts.ReconstructionLoss({"criterion": "l1+ssim", "invalid_policy": "weight_guided", "lambda_edge_aware_smoothness": 0.001})

    def step():   # trainer.py:208-259 after the CNN, then base_trainer.py:297
        net.zero_grad(set_to_none=True)
        images_ip = images * .5 + .5
        net.encode(images, projs, poses, ids_encoder=[0], ids_render=ids_render, images_alt=images_ip)
        all_rays, all_rgb_gt = sampler.sample(images_ip[:, :4], poses[:, :4], projs[:, :4])       # ids_loss = first four frames
        rd = wrapped(all_rays, want_weights=True, want_alphas=True, want_rgb_samps=True)
        rd["fine"] = dict(rd["coarse"])
        rd["rgb_gt"], rd["rays"] = all_rgb_gt, all_rays
        rd = sampler.reconstruct(rd)
        loss, _ = crit(dict(coarse=[rd["coarse"]], fine=[rd["fine"]], rgb_gt=rd["rgb_gt"]))
        loss.backward()
        parallel.all_reduce_mean_([p.grad for p in net.mlp_coarse.parameters() if p.grad is not None])

    for _ in range(args.warmup):
        step()
    native.render_fwd, native.render_bwd = timed(orig_fwd, "fwd"), timed(orig_bwd, "bwd")
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    native.render_fwd, native.render_bwd = orig_fwd, orig_bwd
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = t.item()
    n_rays = n * 4096
    ms = {k: sum(a.elapsed_time(b) for a, b in v) / max(len(v), 1) for k, v in kern.items()}
    if rank == 0:
        flop = 3 * n_rays * Kt * FLOP_PER_POINT          # training = 3x forward (dX + dW)
        achieved = flop / ((ms["fwd"] + ms["bwd"]) * 1e-3) / 1e12
        print(json.dumps({
            "metric": "training-step rays/sec after the CNN: render forward + loss + backward (KITTI-360 shapes)", "value": world * n_rays * args.steps / elapsed,
            "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "exp_kitti_360.yaml shapes: bs=16/GPU, 8 frames (4 loss + 4 render views), 4096 patch rays (8x8) per sample, "
                                   "64 samples/ray, everything after the CNN (feature-map encoder stand-in): sample, render, photometric loss, backward",
                       "rays_per_step_per_gpu": n_rays, "samples_per_ray": Kt, "parallelism": f"batch x{world}"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MATRIX_TFLOPS, "traffic": None,
                         "kernel": "bts_render_fwd + bts_render_bwd (render_kernel_p, render_bwd_kernel, scatter_dg_kernel)",
                         "kernel_ms": ms["fwd"] + ms["bwd"], "fwd_ms": ms["fwd"], "bwd_ms": ms["bwd"], "algorithmic_flop_per_launch": flop,
                         "note": "algorithmic 3 x 13 312 FLOP / sample; the backward is bound by L2 float atomics and LDS, not by MFMA"},
        }))


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    import behindthescenes_amd as bts
    from behindthescenes_amd import _lib
    from behindthescenes_amd import synthetic as S

    _lib.load()
    if args.workload == "train":
        train_workload(args, world, rank, dev)
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    Z_NEAR, Z_FAR = 3.0, 80.0                   # eval_depth.yaml
    scene = S.synthetic_scene(1, V, H, W, C, seed=1000 + rank, intrinsics=S.K_KITTIRAW)
    net = bts.BTSNet(S.field_conf(C, HD, 0, H, W, z_near=Z_NEAR, z_far=Z_FAR))
    S.init_mlp_(net.mlp_coarse, seed=7)
    S.set_feature_map(net, scene["feat"])
    net = net.to(dev).eval()
    wrapped = bts.NeRFRenderer.from_conf(dict(n_coarse=K, lindisp=True, hard_alpha_cap=True)).bind_parallel(net).eval().to(dev)
    sampler = bts.ImageRaySampler(Z_NEAR, Z_FAR)
    images, projs, poses = scene["images"].to(dev), scene["projs"].to(dev), scene["poses"].to(dev)
    n_rays = V * H * W

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    kernel_events = []

    # time the dominant kernel alone: wrap the C-ABI forward launch with events on the launch stream
    from behindthescenes_amd import native
    orig_render_fwd = native.render_fwd

    def timed_render_fwd(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig_render_fwd(*a, **kw)
        e1.record()
        kernel_events.append((e0, e1))
        return out

    def step():
        with torch.no_grad():
            net.encode(images, projs, poses, ids_encoder=[0], ids_render=[0])
            all_rays, all_rgb_gt = sampler.sample(images * .5 + .5, poses, projs)
            rd = wrapped(all_rays, want_weights=True, want_alphas=True)
            rd["fine"] = dict(rd["coarse"])
            rd["rgb_gt"] = all_rgb_gt
            rd = sampler.reconstruct(rd)
            depth_z = bts.distance_to_z(rd["coarse"]["depth"], projs)
        return depth_z

    for _ in range(args.warmup):
        step()
    native.render_fwd = timed_render_fwd
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        step()
        ev[i][1].record()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    native.render_fwd = orig_render_fwd
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = t.item()

    kernel_ms = sum(a.elapsed_time(b) for a, b in kernel_events) / max(len(kernel_events), 1)
    traffic, traffic_detail = None, None
    prof = sorted(d for d in os.listdir(os.path.join(ROOT, "profiles")) if os.path.exists(os.path.join(ROOT, "profiles", d, "traffic.json"))) \
        if os.path.isdir(os.path.join(ROOT, "profiles")) else []
    if prof:
        tj = json.load(open(os.path.join(ROOT, "profiles", prof[-1], "traffic.json")))
        traffic = tj["fetch_bytes"] + tj["write_bytes"]          # HBM bytes per launch of the render kernel
        traffic_detail = {"fetch_bytes": tj["fetch_bytes"], "write_bytes": tj["write_bytes"],
                          "source": f"profiles/{prof[-1]}/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, "
                                    "FETCH_SIZE doubled per MI355X_MICROARCH.md)"}
    step_ms = elapsed * 1e3 / args.steps
    value = world * n_rays * args.steps / elapsed
    if rank == 0:
        flop_per_launch = n_rays * K * FLOP_PER_POINT
        achieved = flop_per_launch / (kernel_ms * 1e-3) / 1e12
        out = {
            "metric": "rendered rays/sec (192x640x64 samples)", "value": value, "unit": "rays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "KITTI eval_depth.yaml forward, bs=1/GPU, 192x640, 2 views x 122880 rays, 64 samples/ray, "
                                   "nv=1, want_weights+alphas, renderer only (feature-map encoder stand-in)",
                       "rays_per_step_per_gpu": n_rays, "samples_per_ray": K, "parallelism": f"frames x{world}"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MATRIX_TFLOPS, "traffic": traffic, "traffic_detail": traffic_detail, "kernel": "bts::render_kernel_p<64,64,0,1,true,true>",
                         "kernel_ms": kernel_ms, "algorithmic_flop_per_launch": flop_per_launch,
                         "note": "algorithmic FLOP (13 312 / sample, SURVEY 8d); the kernel executes 5 248 / sample (projected features, "
                                 "DESIGN.md section 3), 39 of its 40 lin_in rows on the f16 matrix pipe in split precision, the bias row as the fp32 C operand"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, net, args.cpu_rows)
            if args.gpu_eager_baseline:
                out["cpu_baseline"]["gpu_eager_port"] = cpu_baseline(scene, net, 4 * args.cpu_rows, device=dev)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
