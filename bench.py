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

Prints ONE JSON line (rank 0).  Next to the headline it carries `others` (N = 1: BASELINE configs[2], [3], [4], [4] at K = 128 and the
occupancy profile, each = `bench.py --workload X --steps CHILD_STEPS --warmup CHILD_WARMUP` in a child process) and, under torch.distributed.run, `ddp_train`
(KITTI-Raw shapes with the Monodepth2 encoder through DistributedDataParallel: the path's one collective, with `allreduce_ms`);
`--no-others` skips both.  `roofline` is for the dominant kernel (bts::render_kernel_p = bts_render_fwd) TOGETHER WITH bts::project_kernel
(the feature half of lin_in, hoisted out of the render kernel by the declared projected-feature shortcut), both timed live with HIP
events on the launch stream inside the timed region: `achieved` / `frac` = the FLOP those two kernels EXECUTE / their summed time -- a
true fraction of the fp32 vector peak; the algorithmic figure of SURVEY 8d (13 312 FLOP / sample, what a kernel without the shortcut
would have to execute) is reported beside it as `algorithmic_tflops` / `frac_algorithmic` and may exceed 1.  Training lines: `kernel_ms`
= EVERY library call of the step (event pairs around each C-ABI entry point), `gpu_busy_frac` = kernel_ms / ms_per_step, `frac` = the
algorithmic 3 x forward FLOP / kernel_ms.  `roofline.traffic` is the HBM byte count of the committed rocprofv3 PMC passes of the same
workload (profiles/<round>/traffic.json, written by tools/profile.sh), null when absent; `cpu_baseline` times the CPU oracle port on a
bounded sample of the same workload.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_POINT = 2 * (103 * 64 + 64)          # 13 312 FLOP per field query (KITTI MLP), the ALGORITHMIC figure
EXEC_FLOP_PER_POINT = 2 * (40 * 64 + 64) + 2 * 4 * 64   # what the kernel executes: 40 PE / bias rows x 64 on the matrix pipe + lin_out + the 4-tap blend of G
PROJECT_FLOP_PER_TEXEL = 2 * 64 * 64          # bts::project_kernel: G = F . w_in[:, :C]^T, once per texel of the map instead of once per sample
# wave64 VALU issue on one SIMD-32 of gfx950, measured (tools/ubench/valu_issue.hip -> profiles/r05*/valu_issue*.txt): ns per wave-instruction
# per SIMD at TWO resident waves (what the render kernels run at), wall-clock, for an FMA-like and a MUL-like instruction mix
VALU_NS_PER_INST_2WAVES = {"fma_like": 1.88, "mul_like": 1.14}
PEAK_FP32_MATRIX_TFLOPS = 157.3               # MI355X_MICROARCH.md: fp32 vector / fp32-input MFMA peak (256 CU x 256 FLOP/clk x 2.4 GHz)
DTYPE = "f32 (lin_in as 3-term f16 split products on the f16 MFMA, f32 accumulate; everything else f32)"
H, W, K, C, HD, V = 192, 640, 64, 64, 64, 2
# timed / untimed steps of every `others` sub-record (each one a child process, see run_child); tests read these, never a literal.
# 40 warm-up steps: the device needs ~25 ms of continuous work to reach its steady state (profiles/r06q), and ten 0.65 ms steps of
# exp_kitti_raw.yaml's shapes are a quarter of that
CHILD_STEPS, CHILD_WARMUP = 40, 40


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=("eval", "train", "kitti_raw", "re10k", "profile", "glue"), default="eval",
                    help="eval (default): BASELINE configs[1], the headline line.  train: configs[2] (exp_kitti_360.yaml shapes), kitti_raw: "
                         "configs[3] (exp_kitti_raw.yaml, bs 8 / GPU), re10k: configs[4] (exp_re10k.yaml, 256x384, four scales per step): the "
                         "renderer's share of a training step, forward + loss + backward; profile: SURVEY 8f.3, the 64 x 256 x 256 occupancy grid of "
                         "scripts/inference_setup.py (render_profile) as one fused pass, density queries/s")
    ap.add_argument("--samples", type=int, default=0, help="re10k: samples per ray (default 48 = the yaml; BASELINE.json quotes 128)")
    ap.add_argument("--feat-layout", choices=("nchw", "nhwc"), default="nhwc",
                    help="memory format of the stand-in feature maps of the training workloads: nhwc (default since round 6) = torch channels_last, "
                         "what the SHIPPED Monodepth2 hands over (MIOpen's NHWC kernels, bts_conv3x3_fwd) -- read as it is through "
                         "bts_project_features_cl (ABI 8), tile flags as 16 x 4 blocks (ABI 9); nchw = what a plain nn.Conv2d stack returns (the "
                         "reference's encoder on CUDA; rounds 1 - 5 benched this).  The line of one layout carries the other's ms_per_step as "
                         "`other_layout`")
    ap.add_argument("--no-other-layout", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--encoder", choices=("feature_map", "monodepth2"), default="feature_map",
                    help="training workloads: feature_map (default) = learnable stand-in for the CNN output (the renderer's share of the step); "
                         "monodepth2 = the shipped Monodepth2 (ResNet of the yaml, random weights): whole step incl. the CNN")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--entries", action="store_true",
                    help="training workloads, A/B: the entry-by-entry sequence of the reference's trainer (encode, sample, render, reconstruct, loss: "
                         "~20 library calls + torch glue per step) instead of FusedTrainStep's two calls (bts_train_step_fwd / _bwd, ABI 7)")
    ap.add_argument("--concurrent-scales", action="store_true",
                    help="re10k workload, A/B: the four scales' kernel chains side by side on queues of the library (BtsTrainStep.concurrent_scales) "
                         "instead of one after the other on the caller's stream; measured slower by 1 %, profiles/r05i")
    ap.add_argument("--shard", choices=("frames", "rays"), default="frames",
                    help="eval workload under torch.distributed.run: frames (default) = every rank renders its own frame, no collective; rays = ONE "
                         "frame, its rays split over the ranks (parallel.render_sharded) and the per-ray outputs all-gathered inside the timed "
                         "region (SURVEY 8e's second axis; the reference's dead DataParallel hook, nerf.py:454-456)")
    ap.add_argument("--dense-proj-grad", action="store_true",
                    help="training workloads, A/B: the gradient of the projected map as a dense autograd tensor (zero fill + full read) instead of "
                         "the kept (d_proj, tile flags) pair (native.SPARSE_PROJ_GRAD)")
    ap.add_argument("--dense-projection", action="store_true",
                    help="training workloads, A/B: project the whole feature map per render instead of the tiles the render's samples read "
                         "(NeRFRenderer.sparse_projection)")
    ap.add_argument("--tile-stats", action="store_true",
                    help="training workloads, diagnostic: how much of the projected map's gradient a step touches (texels, 16-texel segments, "
                         "64-texel tiles, 8 x 8 blocks); prints to stderr and exits")
    ap.add_argument("--no-others", action="store_true",
                    help="eval workload: skip the `others` sub-records (every other BASELINE config + the occupancy profile, 10 steps each) and, "
                         "under torch.distributed.run, the `ddp_train` sub-record (KITTI-Raw shapes with the Monodepth2 encoder: the real gradient bucket)")
    ap.add_argument("--ops-profile", action="store_true",
                    help="training workloads: run 3 steps under torch.profiler and print the host-side op table (no JSON line)")
    ap.add_argument("--full-outputs", action="store_true",
                    help="train workload: also materialise weights / alphas / invalid / rgb_samps per sample as the reference trainer's dict "
                         "has them (default: lean_training_outputs -- the loss reads per-ray reductions from the render kernel's epilogue)")
    ap.add_argument("--no-gpu-eager-baseline", action="store_true",
                    help="skip ref_gpu_baseline: the oracle's torch ops eagerly on the GPU (the reference's own code path on this device, the "
                         "denominator of north_star's >= 10x target); on by default at N = 1")
    ap.add_argument("--cpu-child", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-rows", type=int, default=24, help="image rows of view 0 rendered by the CPU oracle sample")
    return ap.parse_args()


def park_gc():
    """Collect now, then park the cyclic collector: a generation-2 pass over a process with torch loaded takes ~30 ms, and when it falls
    inside a timed loop of a 0.7 ms step it is 0.3 ms per step of a 100-step run (profiles/r05t; nothing in a 20-step run).  -> the
    function that restores the previous state.  BTS_BENCH_KEEP_GC=1 leaves the collector on."""
    import gc
    gc.collect()
    was_on = gc.isenabled()
    if os.environ.get("BTS_BENCH_KEEP_GC") != "1":
        gc.disable()
    return gc.enable if was_on else (lambda: None)


def gc_parked():
    """what the JSON line records as `gc_parked`: was the cyclic collector off during the timed loops of this process?"""
    return os.environ.get("BTS_BENCH_KEEP_GC") != "1"


# ------------------------------------------------------------------------------------------------------------------------------------
# launch glue: ONE implementation for every workload (and for the CPU stand-in `--workload glue`, which tests/test_distributed_cpu.py
# launches at world size 2 over gloo, so that a first multi-GPU run cannot die in this code)
# ------------------------------------------------------------------------------------------------------------------------------------
def launch_context(env=None):
    """-> (world, rank, local_rank, launched): the torch.distributed.run contract (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    env = os.environ if env is None else env
    world, rank, local_rank = int(env.get("WORLD_SIZE", "1")), int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0"))
    launched = "RANK" in env and "WORLD_SIZE" in env and "MASTER_PORT" in env
    return world, rank, local_rank, launched


def init_group(launched, local_rank, use_cuda=True):
    """One process per GPU over RCCL (backend "nccl", `device_id` so that the communicator is bound to this rank's device) -- also at world
    size 1, so that barrier / MAX all-reduce / DDP run exactly as at N > 1.  Without CUDA (the glue workload on a CPU box): gloo.
    -> the device of this rank."""
    if use_cuda:
        torch.cuda.set_device(local_rank if launched else 0)
    if launched:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if use_cuda:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    return torch.device("cuda", local_rank if launched else 0) if use_cuda else torch.device("cpu")


def _sync(dev):
    if dev.type == "cuda":
        torch.cuda.synchronize()


# HIP event pairs inside the timed region on every EVENT_EVERY-th step (default: every step).  Round 6 suspected the pairs themselves of
# slowing the loop (an event record between two kernels is a barrier packet) and measured it: 1.10 ms per eval step with pairs on every
# step, on every 4th, on every 10th and on the first only -- no difference (profiles/r06q/event_every.txt).  What did cost 0.12 ms per step
# was the collector pass between warm-up and timing (see timed_region).  BTS_BENCH_EVENT_EVERY keeps the experiment repeatable.
EVENT_EVERY = int(os.environ.get("BTS_BENCH_EVENT_EVERY", "1"))


def timed_region(step, steps, dev, events=None, before=None, timer=None, warmup=0, unpark_gc=None):
    """The contract's timed region: barrier + synchronize, EXACTLY `steps` calls of step(), synchronize + barrier, and the MAXIMUM of the
    wall time over the ranks.  `events`: one (start, end) HIP event pair per step, recorded on the current stream around the call -- on
    the sampled steps (every EVENT_EVERY-th), like `timer`'s pairs (a KernelTimer, installed by the caller).
    `before`: runs between the opening barrier and the clock (e.g. reset_peak_memory_stats).
    `warmup`: the W untimed steps, run HERE -- behind the collector pass, directly in front of the opening synchronize.  Until round 6 the
    workloads warmed up first and park_gc() came after: its gc.collect() is ~30 ms of host time with the GPU idle, and the first steps of
    the timed region then ran on a device that had dropped out of its steady state -- a fixed ~2.4 ms per timed region, i.e. 1.10 ms per
    step over 20 steps against 0.98 over 200 (profiles/r06q/event_every.txt: independent of how many steps carry event pairs).  The warm-up
    exists so that timing starts in steady state; a pause between the two defeats it.
    -> (seconds, the last step's return value); timed_region.sampled = the indices of the sampled steps."""
    dist = torch.distributed
    if unpark_gc is None:               # (a caller that has GPU work of its own in front of the warm-up parks the collector before THAT)
        unpark_gc = park_gc()           # (in front of the warm-up and the barrier: the collection takes a different time on every rank)
    if timer is not None:
        timer.enabled = False
    for _ in range(warmup):
        step()
    _sync(dev)
    if dist.is_initialized():
        dist.barrier()
        _sync(dev)
    if before is not None:
        before()
    last = None
    t0 = time.perf_counter()
    sampled = []
    for i in range(steps):
        on = i % EVENT_EVERY == 0
        if timer is not None:
            timer.enabled = on
        if on:
            sampled.append(i)
        if events is not None and on:
            events[i][0].record()
        last = step()
        if events is not None and on:
            events[i][1].record()
    timed_region.sampled = sampled
    if timer is not None:
        timer.enabled = True
    _sync(dev)
    unpark_gc()
    if dist.is_initialized():
        dist.barrier()
        _sync(dev)
    elapsed = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    return elapsed, last


def parallelism(kind, world):
    """config.parallelism of the JSON line: what is split over the ranks"""
    return {"frames": f"frames x{world}", "rays": f"rays x{world} of one frame", "batch": f"batch x{world}"}[kind]


def glue_workload(args, world, rank, dev):
    """NOT a benchmark: the launch glue (launch_context, init_group, timed_region, the JSON line's contract keys) around a stand-in step
    that runs wherever torch runs.  tests/test_distributed_cpu.py starts it as `python -m torch.distributed.run --nproc-per-node 2 bench.py
    --gpus 2 --workload glue` on the CPU container (gloo); rank r's step sleeps (r + 1) ms so that the MAX over the ranks is checkable."""
    x = torch.ones(64, 64, device=dev)

    def step():
        time.sleep(1e-3 * (rank + 1))
        return (x @ x).sum()
    elapsed, _ = timed_region(step, args.steps, dev, warmup=args.warmup)
    if rank != 0:
        return None
    return {"metric": "launch glue stand-in (not a measurement)", "value": world * args.steps / elapsed, "unit": "steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "gc_parked": gc_parked(),
            "config": {"workload": "glue", "parallelism": parallelism(args.shard, world), "backend": torch.distributed.get_backend() if
                       torch.distributed.is_initialized() else None}}


class KernelTimer:
    """HIP event pairs (on the launch stream) around the C-ABI entry points of behindthescenes_amd.native: every bts:: kernel of a step
    lies inside exactly one pair (nested entry points -- distance_to_z -> invert_small -- count once, at the outermost).  What the pairs
    do NOT contain: torch's own kernels (the jitter draw, torch.cat of the parameters, DDP) and host gaps between entry points."""
    ENTRIES = ("nchw_to_nhwc", "nhwc_to_nchw", "pack_rgb", "gen_rays", "patch_rays", "photometric_loss", "sample_coarse", "invert_small",
               "distance_to_z", "project_features", "mark_sampled_tiles", "project_features_bwd", "render_fwd", "render_bwd", "field_query",
               "occupancy_profile", "train_step_fwd", "train_step_bwd", "eval_frame")

    def __init__(self):
        from behindthescenes_amd import native
        self.native, self.orig, self.ev, self.depth = native, {}, {k: [] for k in self.ENTRIES}, 0
        self.enabled = True          # timed_region switches the pairs on for the SAMPLED steps only (see EVENT_EVERY)

    def _wrap(self, name, fn):
        def f(*a, **kw):
            if self.depth or not self.enabled:
                return fn(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.depth += 1
            try:
                e0.record()
                out = fn(*a, **kw)
                e1.record()
            finally:
                self.depth -= 1
            self.ev[name].append((e0, e1))
            return out
        return f

    def install(self):
        for k in self.ENTRIES:
            self.orig[k] = getattr(self.native, k)
            setattr(self.native, k, self._wrap(k, self.orig[k]))

    def remove(self):
        for k, fn in self.orig.items():
            setattr(self.native, k, fn)
        self.orig = {}

    def ms_per_step(self, steps):
        """{entry point: ms per step}, only the ones that ran (call after a synchronize)"""
        return {k: sum(a.elapsed_time(b) for a, b in v) / max(steps, 1) for k, v in self.ev.items() if v}


def cpu_baseline(scene, net, rows, device="cpu", learn_empty=True, threads=None):
    """The oracle ("port" of the reference algorithm: the very torch CPU ops the reference calls; the reference tree itself does not
    exist on the GPU box) on a bounded sample: `rows` full image rows of both views (rows*640*2 rays, K=64), best of 2 after one
    warm-up, for each thread count of a sweep -- the best one is reported (more threads than ~16 LOSE on 30 720-ray chunks).
    device != "cpu": the same ops eagerly on the GPU = the reference's own code path on this device, the denominator of north_star's
    ">= 10x the reference GPU path".  The only place bench.py touches ``oracle/``."""
    from oracle import bts_oracle as O
    cfg = O.FieldConfig(learn_empty=learn_empty)       # z in [3, 80], inv_z, code_mode z, learn_empty default (eval_depth.yaml)
    m = net.mlp_coarse
    mlp = O.MlpParams(w_in=m.lin_in.weight.detach().cpu().clone(), b_in=m.lin_in.bias.detach().cpu().clone(),
                      w_out=m.lin_out.weight.detach().cpu().clone(), b_out=m.lin_out.bias.detach().cpu().clone())
    rays = O.image_rays(scene["poses"], scene["projs"], H, W, cfg.d_min, cfg.d_max).view(1, V, H, W, 8)
    r0 = (H - rows) // 2
    rays = rays[:, :, r0:r0 + rows].reshape(1, -1, 8).contiguous()
    g = torch.Generator().manual_seed(1)
    u = torch.rand(rays.shape[1], K, generator=g)
    st = O.make_state(scene, [0], cfg, net.empty_feature.detach().cpu() if learn_empty else None)
    if device != "cpu":
        rays, u = rays.to(device), u.to(device)
        st = O.FieldState(*[None if t is None else t.to(device) for t in (st.feat, st.K_enc, st.w2c_enc, st.imgs, st.K_r, st.w2c_r, st.empty_feature)])
        mlp = O.MlpParams(mlp.w_in.to(device), mlp.b_in.to(device), [], mlp.w_out.to(device), mlp.b_out.to(device))
    n_rays = rays.shape[1]

    def best_of(n_rep):
        best = float("inf")
        with torch.no_grad():
            for i in range(n_rep + 1):
                t0 = time.perf_counter()
                z = O.sample_coarse(rays.reshape(-1, 8), K, True, u)
                O.composite(rays.reshape(-1, 8), z, 1, st, mlp, cfg, hard_alpha_cap=True)
                if device != "cpu":
                    torch.cuda.synchronize()
                if i > 0:
                    best = min(best, time.perf_counter() - t0)
        return best

    sample = f"{rows} rows x {W} px x {V} views = {n_rays} rays x {K} samples, renderer only, best of 2"
    if device != "cpu":
        return dict(value=n_rays / best_of(3), unit="rays/s", kind="port", device="MI355X, PyTorch-ROCm eager: the oracle's torch ops on cuda:0 "
                    "(= the reference's own op sequence; the reference tree is absent on the GPU box)", sample=sample.replace("best of 2", "best of 3"))
    if threads is not None:      # child of the sweep below: one thread count in a fresh process (clean OpenMP pool)
        return dict(value=n_rays / best_of(2))
    nproc = os.cpu_count() or 8
    sweep = {}
    for t in sorted({c for c in (8, 16, 32, 64, nproc) if c <= nproc}):
        env = dict(os.environ, OMP_NUM_THREADS=str(t), MKL_NUM_THREADS=str(t))
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-child", str(t), "--cpu-rows", str(rows)], env=env, capture_output=True,
                               text=True, timeout=75)
            sweep[t] = float(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else None
        except (subprocess.TimeoutExpired, ValueError, IndexError):
            sweep[t] = None            # slower than 75 s for three passes over the sample: not the best count anyway
    done = {k: v for k, v in sweep.items() if v}
    if not done:
        return dict(value=None, unit="rays/s", kind="port", sample=sample, note="every thread count timed out")
    cores = max(done, key=done.get)
    return dict(value=done[cores], unit="rays/s", cores=cores, kind="port", sample=sample, host_cpus=nproc,
                thread_sweep={str(k): (None if v is None else round(v, 1)) for k, v in sweep.items()},
                note="oracle = CPU restatement with the reference's own torch ops (validated against the unmodified reference: same rays/s "
                     "within 2 % at 8 threads); one fresh process per thread count, best count reported (null = more than 75 s)")


def cpu_child(args):
    """One thread count of the CPU sweep in its own process: the bench scene and MLP rebuilt from their seeds (no GPU touched)."""
    import behindthescenes_amd as bts
    from behindthescenes_amd import synthetic as S
    torch.set_num_threads(args.cpu_child)
    scene = S.synthetic_scene(1, V, H, W, C, seed=1000, intrinsics=S.K_KITTIRAW)
    torch.manual_seed(4242)
    net = bts.BTSNet(S.field_conf(C, HD, 0, H, W, z_near=3.0, z_far=80.0, learn_empty=True))
    S.init_mlp_(net.mlp_coarse, seed=7)
    print(cpu_baseline(scene, net, args.cpu_rows, threads=args.cpu_child)["value"])


TRAIN_WORKLOADS = {
    # BASELINE configs[2]: configs/exp_kitti_360.yaml (bs 16, kitti360-mono: 4 loss + 4 render frames, 4096 patch rays, 64 samples)
    "train": dict(yaml="exp_kitti_360.yaml", n=16, V=8, H=192, W=640, C=64, HD=64, NB=0, K=64, ids_loss=[0, 1, 2, 3], ids_render=[4, 5, 6, 7],
                  rays=4096, z=(3.0, 80.0), intr="K_KITTI360", baseline=0.6, hard_cap=True, code_mode="z", scales=1, resnet=50,
                  num_ch_dec=[32, 32, 64, 128, 256]),
    # BASELINE configs[3]: configs/exp_kitti_raw.yaml at 8 samples per GPU (yaml: 16 over the node; BASELINE: bs 8 / GPU), stereo pair x 2
    # time steps = 2 loss + 2 render frames, 2048 patch rays, 64 samples
    "kitti_raw": dict(yaml="exp_kitti_raw.yaml", n=8, V=4, H=192, W=640, C=64, HD=64, NB=0, K=64, ids_loss=[0, 1], ids_render=[2, 3],
                      rays=2048, z=(3.0, 80.0), intr="K_KITTIRAW", baseline=0.54, hard_cap=True, code_mode="z", scales=1, resnet=50,
                      num_ch_dec=[32, 32, 64, 128, 256]),
    # BASELINE configs[4]: configs/exp_re10k.yaml (bs 24, 3 frames: 1 loss + 2 render, 256x384, C = 32, one ResnetBlockFC of width 32,
    # distance code, z in [1, 100], no alpha cap, 1024 patch rays, K = 48; prediction_mode unset -> "multiscale": FOUR renders per step,
    # trainer.py:220-242)
    "re10k": dict(yaml="exp_re10k.yaml", n=24, V=3, H=256, W=384, C=32, HD=32, NB=1, K=48, ids_loss=[0], ids_render=[1, 2], rays=1024,
                  z=(1.0, 100.0), intr="K_RE10K", baseline=0.2, hard_cap=False, code_mode="distance", scales=4, resnet=18,
                  num_ch_dec=[32, 32, 64, 128, 256]),
}


def train_byte_model(cfg, Kt, shifts, flagged_tiles):
    """HBM bytes of ONE fused training step as the design moves them, for `traffic_ratio`.  Two parts:
      algorithmic  what a perfectly fused step would have to move -- the step's inputs read once and its outputs written once: F of
                   every flagged 64-texel tile, the render frames, the loss patches' colours, one jitter value per sample; the DENSE
                   feature-map gradient (autograd's contract), the per-ray outputs the caller gets (rays, rgb_gt, rgb, depth, two
                   invalid-ray reductions per view);
      state        what the three-pass design parks in HBM between its kernels, each tensor counted once per write and once per read it
                   needs: the projected tiles G (project: write; render, pass A: read), the rgb0-packed render frames (write), per sample
                   z / sigma_raw / trans / rgb_samps (render: write; pass A: read) and g_s (pass A: write; passes B, C: read), the relu
                   gates (plain MLP: 12 B per sample and channel word) or the u0 rows (ResnetBlockFC: 4 d_hidden B per sample) between
                   pass A and passes B / C, dG tiles (scatter: read-modify-write; projection backward: read + the write that returns them to zero).
    `flagged_tiles[s]` = tiles of scale s one step's samples touch (counted on the device after the timed loop)."""
    n, V, H, W, C, HD, NB = cfg["n"], cfg["V"], cfg["H"], cfg["W"], cfg["C"], cfg["HD"], cfg["NB"]
    nv, B = len(cfg["ids_render"]), cfg["n"] * cfg["rays"]
    S = B * Kt                                                    # samples per render
    alg = n * nv * H * W * 3 * 4 + B * 3 * 4 + B * 8 * 4          # render frames, the patches' colours, the rays
    state = n * nv * H * W * 4 * 4                                # rgb0-packed frames
    parts = {}
    for s, sh in enumerate(shifts):
        tile_px = flagged_tiles[s] * 64
        dense = n * C * (H >> sh) * (W >> sh) * 4
        alg += tile_px * C * 4 + S * 4 + dense + B * (nv * 3 + 1 + 2 * nv) * 4          # F tiles, jitter, dF, per-ray outputs
        g = tile_px * HD * 4
        per_sample = (3 * 4 + nv * 3 * 4) * 2 + 4 * 3                                   # z, sigma_raw, trans, rgb_samps w + r; g_s w + 2 r
        per_sample += (4 * HD) * 3 if NB else (HD // 32) * 4 * 3 + (HD * 8) // 64 * 2   # u0 rows w + 2 r | gate words w + r (per-sample + per-channel forms)
        state += 3 * g + S * per_sample + 4 * g + tile_px * C * 4                        # G w + 2 r; dG rmw (2) + r + zero w; F re-read by the projection backward
        parts[f"scale{s}"] = dict(flagged_tiles=int(flagged_tiles[s]), tile_fraction=flagged_tiles[s] / max(1, n * (((H >> sh) * (W >> sh) + 63) // 64)))
    return dict(algorithmic_bytes=int(alg), state_bytes=int(state), detail=parts)


def train_cpu_baseline(cfg, net, scene, rank):
    """cpu_baseline of a training workload: the oracle (the reference's torch ops on the CPU) forward + backward of the render for ONE
    batch sample (bounded: cfg.rays rays x K samples, scale 0), best of 2 after a warm-up, min(16, nproc) threads (the eval sweep's
    best count).  A port, not the target: it says what the CPU path costs per ray, nothing about kernel quality."""
    from oracle import bts_oracle as O
    threads = min(16, os.cpu_count() or 8)
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        ocfg = O.FieldConfig(d_min=cfg["z"][0], d_max=cfg["z"][1], code_mode=cfg["code_mode"])
        m = net.mlp_coarse
        one = {k: v[:1] for k, v in scene.items()}
        g = torch.Generator().manual_seed(1)
        rays = O.image_rays(one["poses"][:, cfg["ids_loss"]], one["projs"][:, cfg["ids_loss"]], cfg["H"], cfg["W"], *cfg["z"])
        rays = rays[:, torch.randperm(rays.shape[1], generator=g)[:cfg["rays"]]].contiguous()
        u = torch.rand(cfg["rays"], cfg["K"], generator=g)
        c_rgb = torch.randn(cfg["rays"], 3 * len(cfg["ids_render"]), generator=g)
        best = float("inf")
        for i in range(3):
            t0 = time.perf_counter()
            params = [p.detach().cpu().clone().requires_grad_(True) for p in
                      [m.lin_in.weight, m.lin_in.bias] + sum([[b.fc_0.weight, b.fc_0.bias, b.fc_1.weight, b.fc_1.bias] for b in m.blocks], []) +
                      [m.lin_out.weight, m.lin_out.bias]]
            blocks = [tuple(params[2 + 4 * j: 6 + 4 * j]) for j in range(len(m.blocks))]
            mlp = O.MlpParams(params[0], params[1], blocks, params[-2], params[-1])
            feat = one["feat"].clone().requires_grad_(True)
            st = O.make_state(dict(one, feat=feat), cfg["ids_render"], ocfg)
            z = O.sample_coarse(rays.reshape(-1, 8), cfg["K"], True, u)
            w, rgb, depth, *_ = O.composite(rays.reshape(-1, 8), z, 1, st, mlp, ocfg, hard_alpha_cap=cfg["hard_cap"])
            ((rgb * c_rgb).sum() + 0.05 * depth.sum()).backward()
            if i > 0:
                best = min(best, time.perf_counter() - t0)
        return dict(value=cfg["rays"] / best, unit="rays/s", cores=threads, kind="port", host_cpus=os.cpu_count(),
                    sample=f"1 batch sample: {cfg['rays']} rays x {cfg['K']} samples, nv = {len(cfg['ids_render'])}, oracle render forward + "
                           "autograd backward (MLP + feature map), no loss term, one scale, best of 2",
                    note="oracle = CPU restatement with the reference's own torch ops")
    finally:
        torch.set_num_threads(prev)


def train_workload(args, world, rank, dev):
    """The training configs of BASELINE.json (TRAIN_WORKLOADS).  One step = the renderer's share of `trainer.py:208-259` + backward:
    encode hand-over (no CNN unless --encoder monodepth2), PatchRaySampler.sample, G = project(F), render with saved activations and
    every output the trainer asks for (weights, alphas, rgb_samps) once per scale (multiscale: net.set_scale(s), trainer.py:220-242),
    reconstruct, the photometric loss (l1+ssim, weight-guided invalid mask, edge-aware smoothness: one HIP pass per scale incl. its
    gradient), backward through bts_render_bwd and bts_project_features_bwd (MLP and feature-map gradients); the task is wrapped through
    parallel.wrap_ddp (DistributedDataParallel + RCCL all-reduce of the gradients when N > 1)."""
    import behindthescenes_amd as bts
    from behindthescenes_amd import native, parallel, synthetic as S
    cfg = dict(TRAIN_WORKLOADS[args.workload])
    if args.samples:
        cfg["K"] = args.samples
    n, Vt, Kt, Hh, Ww, Cc, Hd, Nb = cfg["n"], cfg["V"], cfg["K"], cfg["H"], cfg["W"], cfg["C"], cfg["HD"], cfg["NB"]
    n_scales = cfg["scales"]
    scene = S.synthetic_scene(n, Vt, Hh, Ww, Cc, seed=2000 + rank, intrinsics=getattr(S, cfg["intr"]), baseline=cfg["baseline"], smooth=True)
    conf = S.field_conf(Cc, Hd, Nb, Hh, Ww, z_near=cfg["z"][0], z_far=cfg["z"][1], code_mode=cfg["code_mode"])
    if args.encoder == "monodepth2":
        conf["encoder"] = dict(type="monodepth2", freeze=False, pretrained=False, resnet_layers=cfg["resnet"], num_ch_dec=cfg["num_ch_dec"], d_out=Cc)
        torch.manual_seed(99)
        net = bts.BTSNet(conf)
        if n_scales == 1:
            net.encoder.scales = [0]          # prediction_mode default renders scale 0 only; the other output convolutions still run
    else:
        net = bts.BTSNet(conf)
        net.encoder = bts.FeatureMapEncoder((Hh, Ww), Cc, num_views=n, n_scales=n_scales, pyramid=n_scales > 1, channels_last=args.feat_layout == "nhwc")
        S.set_feature_map(net, scene["feat"])
    S.init_mlp_(net.mlp_coarse, seed=7)
    net = net.to(dev).train()
    renderer = bts.NeRFRenderer.from_conf(dict(n_coarse=Kt, lindisp=True, hard_alpha_cap=cfg["hard_cap"],
                                               lean_training_outputs=not args.full_outputs)).to(dev).train()
    renderer.sparse_projection = not args.dense_projection
    sampler = bts.PatchRaySampler(ray_batch_size=cfg["rays"], z_near=cfg["z"][0], z_far=cfg["z"][1], patch_size=8)
    images, projs, poses = scene["images"].to(dev), scene["projs"].to(dev), scene["poses"].to(dev)
    ids_loss, ids_render = cfg["ids_loss"], cfg["ids_render"]
    wrapped = renderer.bind_parallel(net).train()
    crit = bts.ReconstructionLoss({"criterion": "l1+ssim", "invalid_policy": "weight_guided", "lambda_edge_aware_smoothness": 0.001})
    fused = bts.FusedTrainStep(wrapped, sampler, crit, multiscale=n_scales > 1, fused=not args.entries and not args.full_outputs,
                               concurrent_scales=args.concurrent_scales)

    class Task(torch.nn.Module):
        """trainer.py:208-259 after the data loader, as ONE module so that DistributedDataParallel wraps it exactly as idist.auto_model wraps
        the reference's BTSWrapper (trainer.py:418): the gradient all-reduce of its parameters (MLP + feature maps here; + the CNN in
        a real run) is DDP's bucketed RCCL all-reduce, overlapped with the backward.  The body is behindthescenes_amd.FusedTrainStep:
        encoder, then bts_train_step_fwd (hand-over, patch rays, per scale: flagged-tile projection, render, photometric loss) -- or,
        with --entries, the same kernels entry by entry as the reference's trainer spells them."""

        def __init__(self):
            super().__init__()
            self.step = fused

        def forward(self, images, projs, poses):
            return self.step(images, projs, poses, ids_encoder=[0], ids_render=ids_render, ids_loss=ids_loss)[0]

    task = Task()
    if args.encoder == "feature_map":
        # the stand-in feature maps are per-sample DATA (what the CNN would output), not shared weights: their gradient stays on the rank
        # (in a real run it flows on into the local CNN backward); the all-reduce carries the MLP here and MLP + CNN in a real run
        torch.nn.parallel.DistributedDataParallel._set_params_and_buffers_to_ignore_for_model(
            task, [k for k, _ in task.named_parameters() if ".encoder.feats." in k])
    # a real DistributedDataParallel whenever a process group exists (also at world size 1 under torch.distributed.run: the reducer and
    # the RCCL all-reduce then run on the single-GPU box as they will on a node)
    # (Monodepth2 with one rendered scale: the output convolutions of scales 1-3 run but stay outside the loss' graph)
    model = parallel.wrap_ddp(task, dev, force=True, find_unused_parameters=args.encoder == "monodepth2" and n_scales == 1)
    is_ddp = isinstance(model, torch.nn.parallel.DistributedDataParallel)

    def step():   # base_trainer.py:287-297
        net.zero_grad(set_to_none=True)
        model(images, projs, poses).backward()

    native.SPARSE_PROJ_GRAD = not args.dense_proj_grad
    if args.tile_stats or args.ops_profile:      # (the diagnostics below want a warmed-up process; the timed path warms up inside timed_region)
        for _ in range(args.warmup):
            step()
    if args.tile_stats:
        orig_pb = native.project_features_bwd

        def spy(spec, feat, d_proj, mlp_params, *a, tiles=None, **kw):
            on = (d_proj != 0).any(dim=-1)                                  # (N, H, W)
            N, Hm, Wm = on.shape
            flat = on.reshape(N, -1)
            pad = (-flat.shape[1]) % 64
            flat = torch.nn.functional.pad(flat, (0, pad))
            blk = on[:, :Hm // 8 * 8, :Wm // 8 * 8].reshape(N, Hm // 8, 8, Wm // 8, 8)
            print(f"map {tuple(d_proj.shape)}: texels {on.float().mean().item():.3f}  16-texel segments "
                  f"{flat.reshape(N, -1, 16).any(-1).float().mean().item():.3f}  64 x 1 tiles {flat.reshape(N, -1, 64).any(-1).float().mean().item():.3f}"
                  f"  flagged {'-' if tiles is None else round(tiles.float().mean().item(), 3)}  8x8 blocks {blk.any(4).any(2).float().mean().item():.3f}"
                  f"  16x4 blocks {on[:, :Hm // 4 * 4, :Wm // 16 * 16].reshape(N, Hm // 4, 4, Wm // 16, 16).any(4).any(2).float().mean().item():.3f}",
                  file=sys.stderr)
            return orig_pb(spec, feat, d_proj, mlp_params, *a, tiles=tiles, **kw)
        native.project_features_bwd = spy
        step()
        torch.cuda.synchronize()
        native.project_features_bwd = orig_pb
        return
    if args.ops_profile:      # diagnostic: what the host issues around the HIP kernels (launch-bound steps)
        from torch.profiler import profile, ProfilerActivity
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
            for _ in range(3):
                step()
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60), file=sys.stderr)
        # who issues the small torch kernels: ops that launch something, grouped by the innermost frames of this repository
        by = {}
        for e in prof.events():
            if e.device_time_total <= 0 or not e.name.startswith("aten::") or e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::"):
                continue
            frames = [f for f in (e.stack or []) if ROOT in f][:2] or ["(no python frame)"]
            parent = e.cpu_parent.name if e.cpu_parent is not None else "-"
            k = (e.name, f"shapes {e.input_shapes}  under {parent}  " + " <- ".join(f.replace(ROOT + "/", "") for f in frames))
            c = by.setdefault(k, [0, 0.0])
            c[0] += 1
            c[1] += e.device_time_total
        print("\nop, calls per step, device us per step, issued from", file=sys.stderr)
        for (name, where), (cnt, us) in sorted(by.items(), key=lambda kv: -kv[1][1]):
            print(f"{name:28s} {cnt / 3:6.1f} {us / 3:9.1f}  {where}", file=sys.stderr)
        return
    timer = KernelTimer()
    timer.install()
    elapsed, _ = timed_region(step, args.steps, dev, before=torch.cuda.reset_peak_memory_stats, timer=timer, warmup=args.warmup)
    n_event_steps = len(timed_region.sampled)
    timer.remove()
    # for the byte model of `traffic_ratio`: the 64-texel tiles of every scale's map the LAST step's samples touched (the forward's flags stay
    # in the arena until the next hand-over clears them), outside the timed region
    flagged_tiles, step_shifts = None, None
    try:
        from behindthescenes_amd import train_step as TS
        arenas = [a for pool in TS._ARENAS.values() for a in pool]
        if len(arenas) == 1:
            flagged_tiles = [int(sc["tiles"].ne(0).sum().item()) for sc in arenas[0].scales]
            step_shifts = list(arenas[0].key[12])
    except Exception:
        pass
    allreduce = None
    if is_ddp:
        # the gradient all-reduce under a profiler range: device time of the RCCL kernels of two more steps (outside the timed region)
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step(), step()
            torch.cuda.synchronize()
        coll = [e for e in prof.key_averages() if any(t in e.key.lower() for t in ("nccl", "rccl"))]
        dev_us = lambda e: getattr(e, "device_time_total", None) or getattr(e, "cuda_time_total", 0.0)
        ignored = set(getattr(model, "parameters_to_ignore", ()) or ())
        grad_bytes = sum(p.numel() * p.element_size() for k, p in model.module.named_parameters() if p.requires_grad and k not in ignored)
        allreduce = dict(allreduce_ms=sum(dev_us(e) for e in coll) / 2 / 1e3, kernels=sorted({e.key[:60] for e in coll}), calls_per_step=sum(e.count for e in coll) / 2,
                         bucket_bytes=grad_bytes, world=world, backend=torch.distributed.get_backend(),
                         note="device time of the RCCL kernels per step (torch.profiler, two steps after the timed region); at world 1 the "
                              "collective is a local copy -- the number says the path ran, not what xGMI costs")
    n_rays = n * cfg["rays"]
    # per STEP and entry point: the four scales of re10k are four calls each on the entry-by-entry path
    ms = timer.ms_per_step(n_event_steps)
    if rank == 0:
        flop_pt = 2 * ((Cc + 39) * Hd + Nb * 2 * Hd * Hd + Hd)     # 13 312 (KITTI MLP) / 8 704 (RE10K MLP) per field query
        flop = 3 * n_rays * Kt * flop_pt * n_scales                # training = 3x forward (dX + dW); every scale renders all rays
        kernel_ms = sum(ms.values())                               # EVERY library call of the step
        fwd_ms = sum(v for k, v in ms.items() if k in ("render_fwd", "train_step_fwd"))
        bwd_ms = sum(v for k, v in ms.items() if k in ("render_bwd", "train_step_bwd"))
        step_ms = elapsed * 1e3 / args.steps
        achieved = flop / (kernel_ms * 1e-3) / 1e12
        # HBM bytes and issue counters of every training kernel from the committed rocprofv3 PMC passes (tools/profile.sh <tag> <workload>)
        traffic, counters = None, {}
        pdir = os.path.join(ROOT, "profiles")
        # tools/profile.sh writes traffic_<mode>.json: mode "train" for configs[2], "bwd_re10k" / "bwd_kitti_raw" (bts_render_fwd +
        # bts_render_bwd of the configs[4] / [3] shape) for the others
        # (a pass at another K than the yaml's carries the K in its name: tools/profile.sh <tag> bwd_re10k 128 -> traffic_bwd_re10k_k128.json;
        # a line never shows the counters of another K)
        ksfx = f"_k{args.samples}" if args.samples and args.samples != TRAIN_WORKLOADS[args.workload]["K"] else ""
        byte_model, traffic_ratio, traffic_over_model = None, None, None
        sname = f"traffic_step_{args.workload}{ksfx}" + ("" if args.feat_layout == "nhwc" else f"_{args.feat_layout}") + ".json"
        sprof = sorted(d for d in (os.listdir(pdir) if os.path.isdir(pdir) else []) if os.path.exists(os.path.join(pdir, d, sname)))
        if sprof and args.encoder == "feature_map":
            # tools/profile_step.sh: rocprofv3 --pmc passes of THIS command (the fused step), every bts:: kernel of the step summed
            tj = json.load(open(os.path.join(pdir, sprof[-1], sname)))
            traffic = tj["step"]["fetch_bytes"] + tj["step"]["write_bytes"]
            counters = {k: {f: round(v[f], 4) for f in ("launches_per_step", "kernel_ms", "fetch_bytes", "write_bytes", "l2_hit", "tb_per_s") if f in v}
                        for k, v in tj["kernels"].items()}
            counters["source"] = (f"profiles/{sprof[-1]}/{sname}: rocprofv3 --pmc passes of `bench.py --workload {args.workload}` (tools/profile_step.sh); "
                                  "`traffic` = fetch + write bytes of every bts:: kernel of one fused step; FETCH_SIZE doubled per MI355X_MICROARCH.md")
            if flagged_tiles is not None:
                byte_model = train_byte_model(cfg, Kt, step_shifts, flagged_tiles)
                traffic_ratio = traffic / byte_model["algorithmic_bytes"]
                traffic_over_model = traffic / (byte_model["algorithmic_bytes"] + byte_model["state_bytes"])
        else:
            tnames = ["traffic_train.json"] if args.workload == "train" and not ksfx else [f"traffic_{args.workload}{ksfx}.json", f"traffic_bwd_{args.workload}{ksfx}.json"]
            prof = sorted((d, t) for d in (os.listdir(pdir) if os.path.isdir(pdir) else []) for t in tnames if os.path.exists(os.path.join(pdir, d, t)))
            if prof:
                tname = prof[-1][1]
                prof = [prof[-1][0]]
                tj = json.load(open(os.path.join(pdir, prof[-1], tname)))
                render = [k for k in ("render_kernel_p", "rows_kernel", "scatter_kernel", "dwpe_kernel", "rowsb_kernel", "dwpe_rows_kernel") if k in tj and "fetch_bytes" in tj[k]]
                traffic = sum(tj[k]["fetch_bytes"] + tj[k]["write_bytes"] for k in render) if render else None
                counters = {k: {f: v[f] for f in ("kernel_ms_rocprof", "fetch_bytes", "write_bytes", "valu_busy", "mfma_busy", "wait_frac", "l2_hit") if f in v}
                            for k, v in tj.items()}
                counters["source"] = (f"profiles/{prof[-1]}/{tname}: rocprofv3 --pmc passes of tools/train_probe.py (same shapes); `traffic` = "
                                      "fetch + write bytes of " + ", ".join(render) + " per launch; FETCH_SIZE doubled per MI355X_MICROARCH.md")
        what = (f"{cfg['yaml']} shapes: bs={n}/GPU, {Vt} frames ({len(ids_loss)} loss + {len(ids_render)} render views), {Hh}x{Ww}, {cfg['rays']} patch "
                f"rays (8x8) per sample, {Kt} samples/ray, C={Cc}, d_hidden={Hd}, {Nb} ResnetBlockFC, code {cfg['code_mode']}, "
                + (f"{n_scales} renders per step (multiscale, trainer.py:220-242), " if n_scales > 1 else "")
                + ("everything after the CNN (feature-map encoder stand-in): " if args.encoder == "feature_map" else
                   f"whole step incl. the shipped Monodepth2 (ResNet-{cfg['resnet']}, random weights), "
                   + "hand-over F -> bts_project_features: ")
                + "sample, render, photometric loss, backward; "
                + ("every per-sample output of the reference trainer's dict materialised (--full-outputs)" if args.full_outputs else
                   "lean_training_outputs: the loss' invalid-ray mask reads per-ray reductions from the render epilogue (SURVEY 8f.1)"))
        out = {
            "metric": f"training-step rays/sec ({cfg['yaml']} shapes): render forward + loss + backward" + (" + CNN" if args.encoder != "feature_map" else ""),
            "value": world * n_rays * args.steps / elapsed,
            "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic", "gc_parked": gc_parked(),
            "config": {"workload": what, "rays_per_step_per_gpu": n_rays, "samples_per_ray": Kt, "renders_per_step": n_scales,
                       "parallelism": parallelism("batch", world), "peak_hbm_bytes": torch.cuda.max_memory_allocated(),
                       "feat_layout": "nhwc (channels_last: Monodepth2's hand-over)" if args.encoder != "feature_map" else args.feat_layout},
            "roofline": {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MATRIX_TFLOPS, "traffic": traffic, "event_steps": n_event_steps,
                         **({"algorithmic_bytes": byte_model["algorithmic_bytes"], "state_bytes": byte_model["state_bytes"], "traffic_ratio": traffic_ratio,
                             "traffic_over_model": traffic_over_model, "byte_model": byte_model["detail"]} if byte_model else {}),
                         "kernel": ("bts_train_step_fwd + bts_train_step_bwd: every bts:: kernel of the step (hand-over, patch rays, tile flags, "
                                    "projection, render, loss, the backward's passes, projection backward)" if fused.last_path == "fused" else
                                    "every library call of the step, entry by entry: " + ", ".join(sorted(ms))),
                         "kernel_ms": kernel_ms, "fwd_ms": fwd_ms, "bwd_ms": bwd_ms, "entry_ms": {k: round(v, 4) for k, v in sorted(ms.items())},
                         "gpu_busy_frac": kernel_ms / step_ms, "frac_step": flop / (step_ms * 1e-3) / 1e12 / PEAK_FP32_MATRIX_TFLOPS,
                         "algorithmic_flop_per_step": flop, "path": fused.last_path,
                         "counters": counters,
                         "note": f"`achieved` / `frac` = the algorithmic 3 x {flop_pt} FLOP / sample (SURVEY 8d) over kernel_ms = the summed "
                                 "HIP-event time of EVERY library call of a step (nothing that carries priced FLOPs is left out: the feature "
                                 "half of lin_in and its two gradients live in the projection kernels); frac_step = the same over the whole "
                                 "step; gpu_busy_frac = kernel_ms / ms_per_step (the rest: torch's jitter draw / parameter packing / autograd, "
                                 "host gaps).  fwd_ms / bwd_ms = the two render phases (the whole phase on the fused path).  Backward passes: "
                                 "DESIGN.md section 3"},
        }
        if world == 1 and not args.no_cpu_baseline and args.encoder == "feature_map":
            out["cpu_baseline"] = train_cpu_baseline(cfg, net, scene, rank)
        if allreduce is not None:
            out["allreduce"] = allreduce
        if world == 1 and args.encoder == "feature_map" and not args.no_other_layout and not args.entries:
            # the same step with the stand-in maps in the OTHER memory format (20 steps after 5, same process): both are always on the line
            import copy
            a2 = copy.copy(args)
            a2.feat_layout, a2.steps, a2.warmup, a2.no_cpu_baseline, a2.no_other_layout = ("nchw" if args.feat_layout == "nhwc" else "nhwc"), 20, 5, True, True
            torch.cuda.empty_cache()
            try:
                o2 = train_workload(a2, world, rank, dev)
                out["other_layout"] = {"feat_layout": a2.feat_layout, "ms_per_step": o2["ms_per_step"], "kernel_ms": o2["roofline"]["kernel_ms"],
                                       "steps": a2.steps, "warmup": a2.warmup}
            except Exception as e:      # a side figure never costs the line
                out["other_layout"] = {"feat_layout": a2.feat_layout, "error": f"{type(e).__name__}: {e}"[:300]}
        return out
    return None


def profile_workload(args, world, rank, dev):
    """the occupancy profile of scripts/inference_setup.py:201-229 -- 64 x 256 x 256 = 4.19 M density queries of the
    KITTI-360 field per frame, sigma := 1 on invalid points, running sum over the 64 vertical levels, count(<= 8) / 64.  One step =
    `net.occupancy_profile(grid, 64)` (bts_occupancy_profile: ONE kernel, no per-point tensor in HBM).  Beside it, once: the
    reference's own flow on the HIP field query (84 chunks of 50 000 points through net.forward + torch post-processing)."""
    import behindthescenes_amd as bts
    from behindthescenes_amd import synthetic as S
    Hh, Ww = 192, 640
    scene = S.synthetic_scene(1, 2, Hh, Ww, C, seed=3000 + rank, intrinsics=S.K_KITTI360, baseline=0.6, smooth=True)
    torch.manual_seed(4242)
    net = bts.BTSNet(S.field_conf(C, HD, 0, Hh, Ww, learn_empty=True))
    S.init_mlp_(net.mlp_coarse, seed=7)
    with torch.no_grad():
        net.mlp_coarse.lin_out.bias.fill_(-2.0)      # densities of 0.1 - 1: the running sums cross the threshold inside the grid
    S.set_feature_map(net, scene["feat"])
    net = net.to(dev).eval()
    grid = S.profile_points()
    Y, Z, X, _ = grid.shape
    pts = grid.reshape(1, -1, 3).to(dev).contiguous()
    with torch.no_grad():
        net.encode(scene["images"].to(dev), scene["projs"].to(dev), scene["poses"].to(dev), ids_encoder=[0], ids_render=[0, 1])

    def step():
        return net.occupancy_profile(pts, Y)

    def reference_flow():   # inference_setup.py:205-228 on the HIP field query
        sig, inv = [], []
        for f in range(0, pts.shape[1], 50000):
            _, i_, s_ = net(pts[:, f:f + 50000].contiguous())
            sig.append(s_), inv.append(i_)
        sigmas, invalid = torch.cat(sig, 1), torch.cat(inv, 1)
        sigmas[torch.any(invalid > 0, dim=-1)] = 1
        a = sigmas.reshape(Y, Z, X)
        return (torch.cumsum(a, 0) <= 8).float().sum(0) / Y

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    elapsed, prof = timed_region(step, args.steps, dev, events=ev, warmup=args.warmup)
    kernel_ms = sum(ev[i][0].elapsed_time(ev[i][1]) for i in timed_region.sampled) / len(timed_region.sampled)
    n_pts = Y * Z * X
    if rank == 0:
        ref_ms, agree = None, None
        if world == 1:
            with torch.no_grad():
                reference_flow()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                ref_prof = reference_flow()
                torch.cuda.synchronize()
                ref_ms = (time.perf_counter() - t1) * 1e3
            agree = float((ref_prof.reshape(-1) == prof.reshape(-1)).float().mean())
        flop = n_pts * FLOP_PER_POINT                 # algorithmic
        exec_flop = n_pts * EXEC_FLOP_PER_POINT       # what the query kernel executes (the projected map is built by encode, outside the step)
        achieved = exec_flop / (kernel_ms * 1e-3) / 1e12
        out = {
            "metric": "density-field queries/sec (64x256x256 occupancy profile, scripts/inference_setup.py)", "value": world * n_pts * args.steps / elapsed,
            "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic", "gc_parked": gc_parked(),
            "config": {"workload": "occupancy profile of one KITTI-360 frame: 64 x 256 x 256 = 4 194 304 query points (get_pts defaults), nv = 2 "
                                   "views flag invalid points, learn_empty=True, one fused pass (bts_occupancy_profile)", "points_per_step_per_gpu": n_pts,
                       "parallelism": parallelism("frames", world),
                       "reference_flow_on_hip_queries_ms": ref_ms, "columns_equal_to_reference_flow": agree},
            "roofline": {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MATRIX_TFLOPS,
                         "traffic": None, "kernel": "bts::query_kernel_p<64,64,0,2> (profile mode)", "kernel_ms": kernel_ms,
                         "executed_flop_per_launch": exec_flop, "algorithmic_flop_per_launch": flop,
                         "algorithmic_tflops": flop / (kernel_ms * 1e-3) / 1e12, "frac_algorithmic": flop / (kernel_ms * 1e-3) / 1e12 / PEAK_FP32_MATRIX_TFLOPS,
                         "note": "`achieved` / `frac` = the 5 760 FLOP / query the kernel EXECUTES (the projected-feature form, DESIGN.md section 3, like "
                                 "the render kernel) against the fp32 vector = fp32-input-MFMA peak; `frac_algorithmic` prices SURVEY 8d's 13 312 FLOP / "
                                 "query and exceeds 1 by construction"},
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import bts_oracle as O
            threads = min(16, os.cpu_count() or 8)
            prev = torch.get_num_threads()
            torch.set_num_threads(threads)
            small = O.profile_points(x_res=64, z_res=64)
            m = net.mlp_coarse
            mlp = O.MlpParams(m.lin_in.weight.detach().cpu(), m.lin_in.bias.detach().cpu(), [], m.lin_out.weight.detach().cpu(), m.lin_out.bias.detach().cpu())
            st = O.make_state(scene, [0, 1], O.FieldConfig(learn_empty=True), net.empty_feature.detach().cpu())
            best = float("inf")
            with torch.no_grad():
                for i in range(3):
                    t1 = time.perf_counter()
                    O.occupancy_profile(small, st, mlp, O.FieldConfig(learn_empty=True))
                    if i:
                        best = min(best, time.perf_counter() - t1)
            torch.set_num_threads(prev)
            out["cpu_baseline"] = dict(value=small.numel() / 3 / best, unit="points/s", cores=threads, kind="port", host_cpus=os.cpu_count(),
                                       sample="64 x 64 x 64 = 262 144 points of the same grid, oracle render_profile (50 000-point chunks), best of 2")
        return out
    return None


def main():
    args = parse()
    if args.cpu_child:
        return cpu_child(args)
    world, rank, local_rank, launched = launch_context()
    if args.workload == "glue":         # the launch glue alone, wherever torch runs (CPU: gloo) -- see glue_workload
        dev = init_group(launched, local_rank, use_cuda=torch.cuda.is_available())
        out = glue_workload(args, world, rank, dev)
        if out is not None:
            print(json.dumps(out))
        if launched:
            torch.distributed.destroy_process_group()
        return
    dev = init_group(launched, local_rank)

    import behindthescenes_amd as bts
    from behindthescenes_amd import _lib
    from behindthescenes_amd import synthetic as S

    _lib.load()
    if args.workload != "eval":
        out = (profile_workload if args.workload == "profile" else train_workload)(args, world, rank, dev)
        if out is not None:
            print(json.dumps(out))
        if launched:
            torch.distributed.destroy_process_group()
        return
    Z_NEAR, Z_FAR = 3.0, 80.0                   # eval_depth.yaml
    scene = S.synthetic_scene(1, V, H, W, C, seed=1000 + rank, intrinsics=S.K_KITTIRAW)
    torch.manual_seed(4242)   # BTSNet draws its empty_feature from the global generator
    # eval_depth.yaml does not set learn_empty, so the reference runs with BTSNet's default learn_empty=True (models_bts.py:24)
    net = bts.BTSNet(S.field_conf(C, HD, 0, H, W, z_near=Z_NEAR, z_far=Z_FAR, learn_empty=True))
    S.init_mlp_(net.mlp_coarse, seed=7)
    S.set_feature_map(net, scene["feat"])
    net = net.to(dev).eval()
    wrapped = bts.NeRFRenderer.from_conf(dict(n_coarse=K, lindisp=True, hard_alpha_cap=True)).bind_parallel(net).eval().to(dev)
    sampler = bts.ImageRaySampler(Z_NEAR, Z_FAR)
    images, projs, poses = scene["images"].to(dev), scene["projs"].to(dev), scene["poses"].to(dev)
    n_rays = V * H * W

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    from behindthescenes_amd import native, parallel
    timer = KernelTimer()       # event pairs around every C-ABI entry point of the step (render_fwd = the dominant kernel)
    shard_rays = launched and args.shard == "rays"
    if shard_rays:
        # SURVEY 8e, second axis: ONE frame (the same scene on every rank), its rays split over the ranks, per-ray outputs all-gathered
        scene = S.synthetic_scene(1, V, H, W, C, seed=1000, intrinsics=S.K_KITTIRAW)
        S.set_feature_map(net, scene["feat"].to(dev))
        images, projs, poses = scene["images"].to(dev), scene["projs"].to(dev), scene["poses"].to(dev)

    frame = bts.FusedEvalFrame(wrapped, sampler, fused=not args.entries and not shard_rays)

    def step():
        # evaluator.py:60-79 after the CNN.  Default: behindthescenes_amd.FusedEvalFrame = ONE library call (bts_eval_frame: hand-over, rays of
        # every pixel, render with sample_coarse inside, distance_to_z); --entries / --shard rays: the same kernels entry by entry
        if frame.fused:
            return frame(images, projs, poses, ids_encoder=[0], ids_render=[0])["coarse"][0]["depth"]
        with torch.no_grad():
            net.encode(images, projs, poses, ids_encoder=[0], ids_render=[0])
            all_rays, all_rgb_gt = sampler.sample(images * .5 + .5, poses, projs)
            if shard_rays:    # every rank holds the field; rank r renders rays [r B / N, (r + 1) B / N) and gathers depth / rgb / weights / alphas
                rd = parallel.render_sharded(wrapped, all_rays, rank, world, want_weights=True, want_alphas=True)
            else:
                rd = wrapped(all_rays, want_weights=True, want_alphas=True)
            rd["fine"] = dict(rd["coarse"])
            rd["rgb_gt"] = all_rgb_gt
            rd = sampler.reconstruct(rd)
            depth_z = bts.distance_to_z(rd["coarse"]["depth"], projs)
        return depth_z

    # Everything ELSE this workload measures on the GPU runs first, the warm-up and the timed region last: the device takes ~25 ms of
    # continuous work to reach its steady state after the idle seconds of the set-up, and the driver's W = 5 warm-up frames are 5 ms
    # (profiles/r06q: 20 timed steps after 5 warm-up frames 1.05 ms each, after 25 and after 100: 0.99).  So the reference split below
    # -- the same kernels entry by entry, each library call in its own event pair; it used to FOLLOW the timed region (3 + 10 frames) --
    # now precedes it, with 20 untimed frames of its own in front of its 10 (so that it, too, measures a device in steady state), and the
    # collector pass (park_gc: ~30 ms of host time with the GPU idle) precedes both.  The run's parts are ordered so that the timed
    # region measures the steady state; profiles/r06q: 1.05 -> 0.99 ms per step at the driver's --steps 20 --warmup 5.
    split_ms = None
    unpark_gc = park_gc()     # (the collector pass -- ~30 ms of host time with the GPU idle -- in front of ALL of it)
    if frame.fused:
        frame.fused = False
        for _ in range(20):
            step()
        t2 = KernelTimer()
        t2.install()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        t2.remove()
        split_ms = t2.ms_per_step(10)
        frame.fused = True
    timer.install()
    elapsed, _ = timed_region(step, args.steps, dev, timer=timer, warmup=args.warmup, unpark_gc=unpark_gc)
    n_event_steps = len(timed_region.sampled)
    timer.remove()
    # the same loop once more with Python's cyclic collector ON (what an evaluation loop that does not park it sees), outside the timed
    # region: reported beside the headline as `ms_per_step_gc_on` (round-5 advice: the headline's loop runs with the collector parked)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(args.steps):
        step()
    torch.cuda.synchronize()
    gc_on_ms = (time.perf_counter() - t1) * 1e3 / args.steps

    entry_ms = timer.ms_per_step(n_event_steps)
    one_call = "eval_frame" in entry_ms
    if one_call:
        # the frame is ONE library call: its event pair holds every bts:: kernel of the frame (render_kernel_p, project_kernel and five
        # small ones); the roofline prices the two kernels that carry FLOPs over the WHOLE call's time.  (`split_ms`: the same kernels
        # entry by entry, each in its own event pair -- measured in front of the warm-up, see above)
        kernel_ms, project_ms = entry_ms["eval_frame"], 0.0
    else:
        kernel_ms = entry_ms.get("render_fwd", float("nan"))          # bts::render_kernel_p alone
        project_ms = entry_ms.get("project_features", 0.0)            # bts::project_kernel: the feature half of lin_in, once per texel
        split_ms = entry_ms
    # counters of the same kernel on the same workload from the committed rocprofv3 PMC passes (tools/profile.sh -> profiles/<tag>/)
    traffic, counters = None, {}
    prof = sorted(d for d in os.listdir(os.path.join(ROOT, "profiles")) if os.path.exists(os.path.join(ROOT, "profiles", d, "traffic.json"))) \
        if os.path.isdir(os.path.join(ROOT, "profiles")) else []
    if prof:
        tj = json.load(open(os.path.join(ROOT, "profiles", prof[-1], "traffic.json")))
        traffic = tj["fetch_bytes"] + tj["write_bytes"]          # HBM bytes per launch of the render kernel
        counters = {k: tj[k] for k in ("fetch_bytes", "write_bytes", "valu_busy", "mfma_busy", "wait_frac", "valu_insts_per_ray", "mfma_insts_per_ray",
                                       "kernel_ms_rocprof", "l2_hit") if k in tj}
        counters["source"] = (f"profiles/{prof[-1]}/traffic.json: rocprofv3 --pmc passes of tools/kernel_probe.py (same kernel, same workload); FETCH_SIZE "
                              "doubled per MI355X_MICROARCH.md; valu_busy = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES, "
                              "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles)")
    step_ms = elapsed * 1e3 / args.steps
    value = (1 if shard_rays else world) * n_rays * args.steps / elapsed
    if rank == 0:
        rays_launch = n_rays // world if shard_rays else n_rays            # rays one launch of the render kernel processes on this rank
        flop_per_launch = rays_launch * K * FLOP_PER_POINT                  # algorithmic
        exec_flop = rays_launch * K * EXEC_FLOP_PER_POINT + H * W * PROJECT_FLOP_PER_TEXEL      # what render_kernel_p + project_kernel execute
        exec_ms = kernel_ms + project_ms
        achieved = exec_flop / (exec_ms * 1e-3) / 1e12
        render_ms = split_ms.get("render_fwd", kernel_ms)                  # the render kernel alone (entry-by-entry event pair)
        algorithmic = flop_per_launch / (render_ms * 1e-3) / 1e12
        # how much of the kernel's time the SIMDs need just to ISSUE its VALU instructions: wave-instructions per SIMD x the measured issue
        # time per instruction at two resident waves (tools/ubench/valu_issue.hip), over the kernel time
        valu_issue = None
        if counters.get("valu_insts_per_ray"):
            per_simd = counters["valu_insts_per_ray"] * rays_launch / 1024
            valu_issue = {k: per_simd * ns * 1e-6 / render_ms for k, ns in VALU_NS_PER_INST_2WAVES.items()}
        out = {
            "metric": "rendered rays/sec (192x640x64 samples)", "value": value, "unit": "rays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE, "data": "synthetic", "gc_parked": gc_parked(), "ms_per_step_gc_on": gc_on_ms,
            "config": {"workload": "KITTI eval_depth.yaml forward, bs=1/GPU, 192x640, 2 views x 122880 rays, 64 samples/ray, nv=1, "
                                   "learn_empty=True (the yaml's effective default), want_weights+alphas, renderer only (feature-map "
                                   "encoder stand-in)" + (": ONE frame, its rays sharded over the ranks + all-gather of the per-ray outputs" if shard_rays else ""),
                       "rays_per_step_per_gpu": rays_launch, "samples_per_ray": K, "parallelism": parallelism("rays" if shard_rays else "frames", world),
                       **({"all_gather_bytes_per_step": n_rays * (3 + 1 + 3 * K) * 4,
                           "all_gather": "rgb, depth, weights, alphas, invalid of all rays on every rank (what the un-sharded call returns)"} if shard_rays else {})},
            "roofline": {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MATRIX_TFLOPS, "traffic": traffic,
                         "kernel": "bts::render_kernel_p<64,64,0,1,true,true> + bts::project_kernel<64,64>" + (" inside bts_eval_frame (one call: + cameras, "
                                   "rgb0 packing, rays, two small inverses, distance_to_z)" if one_call else ""),
                         "kernel_ms": kernel_ms, "project_ms": project_ms, "render_kernel_ms": render_ms, "event_steps": n_event_steps,
                         "entry_ms_entry_by_entry": {k: round(v, 4) for k, v in sorted(split_ms.items())} if one_call else None,
                         "executed_flop_per_launch": exec_flop, "algorithmic_flop_per_launch": flop_per_launch,
                         "algorithmic_tflops": algorithmic, "frac_algorithmic": algorithmic / PEAK_FP32_MATRIX_TFLOPS,
                         "valu_issue_frac": valu_issue, "entry_ms": {k: round(v, 4) for k, v in sorted(entry_ms.items())},
                         "gpu_busy_frac": sum(entry_ms.values()) / step_ms, "counters": counters,
                         "note": "bound: VALU issue + latency at 2 waves / SIMD (not the matrix pipe, not HBM).  `achieved` / `frac` = the FLOP the "
                                 "render kernel (5 760 / sample: 40 encoding + bias rows x 64 on the matrix pipe, lin_out, the 4-tap blend of G) "
                                 "and bts::project_kernel (2 x 64 x 64 / texel: the feature half of lin_in, hoisted out of the sample loop by "
                                 "the declared projected-feature shortcut, DESIGN.md section 3) EXECUTE, over their summed event time, against "
                                 "the fp32 vector = fp32-input-MFMA peak: a true fraction (with the one-call frame the time is the whole call's: "
                                 "every bts:: kernel of the frame; `render_kernel_ms` = the render kernel alone, from ten entry-by-entry frames "
                                 "after the timed region).  `algorithmic_tflops` / `frac_algorithmic` price "
                                 "SURVEY 8d's 13 312 FLOP / sample over the render kernel's time: what a kernel without the shortcut would have "
                                 "to sustain; above 1 by construction.  `valu_issue_frac` = VALU wave-instructions per SIMD (PMC) x the measured "
                                 "issue time per instruction at two waves per SIMD (tools/ubench/valu_issue.hip; FMA-like / MUL-like mix) / "
                                 "kernel time.  The 39 lin_in rows run as 3 f16 MFMAs each (f16 pipe peak 2.5 PF: < 1 % used)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, net, args.cpu_rows)
            if not args.no_gpu_eager_baseline:
                ref = cpu_baseline(scene, net, 4 * args.cpu_rows, device=dev)
                ref["ours_over_ref"] = value / ref["value"]
                out["ref_gpu_baseline"] = ref
    # ---- the other BASELINE configs on the same line (every rank runs them: their barriers are collectives)
    del net, wrapped, scene, images
    torch.cuda.empty_cache()
    # Under torch.distributed.run the sub-record `ddp_train` is the first thing in this file that runs a real collective over RCCL at N > 1
    # (no multi-GPU node has been available to any round).  Should it never return -- a rank that failed while its peers wait inside an
    # all-reduce --, the headline measured above must still reach the driver: a watchdog prints the line without it and ends the process.
    watchdog = None
    if launched and not args.no_others:
        import threading

        def bail():
            if rank == 0:
                out["ddp_train"] = {"error": "did not finish within 300 s (a collective that never completed?); the headline above stands"}
                print(json.dumps(out), flush=True)
            os._exit(0)
        watchdog = threading.Timer(300.0, bail)
        watchdog.daemon = True
        watchdog.start()
    subs = sub_records(args, world, rank, dev, launched)
    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        out.update(subs)
        print(json.dumps(out))
    if launched:
        torch.distributed.destroy_process_group()


def _condense(rec):
    """what a sub-record keeps of a workload's own JSON line"""
    r = rec["roofline"]
    keep = {"metric": rec["metric"], "value": rec["value"], "unit": rec["unit"], "steps": rec["steps"], "warmup": rec["warmup"],
            "ms_per_step": rec["ms_per_step"], "workload": rec["config"]["workload"],
            **({"feat_layout": rec["config"]["feat_layout"]} if "feat_layout" in rec["config"] else {}),
            **({"other_layout": rec["other_layout"]} if "other_layout" in rec else {}),
            "roofline": {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "frac_step", "frac_algorithmic", "gpu_busy_frac", "traffic", "algorithmic_bytes", "state_bytes", "traffic_ratio", "traffic_over_model", "kernel", "kernel_ms", "fwd_ms", "bwd_ms", "entry_ms", "path") if k in r}}
    if "allreduce" in rec:
        keep["allreduce"] = rec["allreduce"]
    return keep


def sub_records(args, world, rank, dev, launched):
    """`others`: BASELINE configs[2..4] (+ configs[4] at BASELINE.json's K = 128) and the occupancy profile, each measured by the same
    code as `bench.py --workload X` (its own process each, 10 steps after 3 warm-ups, no CPU baseline) so that the ONE line the driver
    records carries every workload; N = 1 only.  `ddp_train` (whenever a process group exists, i.e. under torch.distributed.run): the KITTI-Raw training shapes
    WITH the shipped Monodepth2 encoder -- the real ~140 MB gradient bucket through DistributedDataParallel's RCCL all-reduce, the one
    exchange step of the path (trainer.py:418); the eval headline next to it is collective-free by construction (independent frames)."""
    import copy
    out = {}
    if args.no_others:
        return out

    def run(workload, steps=10, warmup=3, **kw):
        a = copy.copy(args)
        a.workload, a.steps, a.warmup, a.no_cpu_baseline, a.samples, a.encoder = workload, steps, warmup, True, 0, "feature_map"
        for k, v in kw.items():
            setattr(a, k, v)
        try:
            rec = (profile_workload if workload == "profile" else train_workload)(a, world, rank, dev)
            return _condense(rec) if rec is not None else None
        except Exception as e:      # a sub-record never costs the headline
            return {"error": f"{type(e).__name__}: {e}"[:500]}
        finally:
            torch.cuda.empty_cache()

    def run_child(workload, samples=0):
        """`python bench.py --workload X --steps CHILD_STEPS --warmup CHILD_WARMUP` in its own process -- the very command a reader would run by hand (in this
        process the allocator and host state the previous workload leaves behind cost the next one up to a millisecond per step).  40 + 40
        steps: with 10 + 3 a 0.7 ms step still carries the process' first-use costs (0.82 vs 0.71 ms for kitti_raw, profiles/r05z vs r05t);
        the extra steps are ~0.1 s of a child process that spends seconds importing torch."""
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK",
                                                                  "LOCAL_WORLD_SIZE", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", workload, "--steps", str(CHILD_STEPS), "--warmup", str(CHILD_WARMUP), "--no-cpu-baseline", "--no-others"]
        if samples:
            cmd += ["--samples", str(samples)]
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                return {"error": (r.stderr or r.stdout)[-500:]}
            return _condense(json.loads(line[-1]))
        except Exception as e:
            return {"error": f"{type(e).__name__}: {e}"[:500]}

    if world == 1:
        others = {"train": run_child("train"), "kitti_raw": run_child("kitti_raw"), "re10k": run_child("re10k"),
                  "re10k_k128": run_child("re10k", 128), "profile": run_child("profile")}
        out["others"] = others
    if launched:
        out["ddp_train"] = run("kitti_raw", steps=5, warmup=2, encoder="monodepth2")
    return out


if __name__ == "__main__":
    main()
