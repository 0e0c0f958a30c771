"""GPU: the data-parallel training path and the ray-sharded inference path with the REAL kernels under torch.distributed (round-2
verdict item 7).  No multi-GPU box is available to the tests, so two processes share cuda:0: RCCL refuses two ranks on one device,
hence the gloo backend -- what is under test is the integration, not the transport: that DistributedDataParallel's hooks fire
for parameters that reach the loss only through RenderFunction / ProjectFunction (custom autograd nodes over a torch.cat of the MLP
parameters, gradient_as_bucket_view=True), that the averaged gradients equal a single process on the union batch, and that
render_sharded reproduces the un-sharded render bit for bit."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, out_dir, backend="gloo"):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                   LOCAL_RANK=str(rank if backend == "nccl" else 0), BTS_TEST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_ddp_worker.py"), str(out_dir)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [torch.load(os.path.join(out_dir, f"rank{r}_of{world}.pt")) for r in range(world)]


def _check_ddp(single, ranks):
    assert all(all(r["same"].values()) for r in ranks + [single]), [r["same"] for r in ranks]
    for k, ref in single["grads"].items():
        a, b = ranks[0]["grads"][k], ranks[1]["grads"][k]
        assert torch.equal(a, b), f"{k}: ranks hold different gradients after the all-reduce"
        err = (a - ref).abs().max().item() / ref.abs().max().item()
        assert err <= 2e-5, (k, err)       # float atomics / bucket summation order: not bit-exact, far below the 1e-4 gradient bar
    assert abs((ranks[0]["loss"] + ranks[1]["loss"]) / 2 - single["loss"]) <= 1e-5 * abs(single["loss"])


def test_ddp_gradients_equal_the_single_process_union_batch_and_sharded_render_is_exact(tmp_path):
    _check_ddp(_run(1, tmp_path)[0], _run(2, tmp_path))


def test_ddp_over_rccl_one_rank_per_device(tmp_path):
    """The same comparison over backend "nccl" (= RCCL), one rank per GPU, `device_id=`, gradient_as_bucket_view -- what a real
    multi-GPU run uses.  Needs two devices: on the single-GPU test box the RCCL path is still exercised at world size 1
    (init_process_group("nccl", device_id), all-reduce, barrier, all_gather_cat) and the 2-rank comparison is reported as NOT RUN."""
    single = _run(1, tmp_path, backend="nccl")[0]
    assert single["backend"] == "nccl" and all(single["same"].values())
    # world size 1 over RCCL wraps a REAL DistributedDataParallel (wrap_ddp(force=True)): its gradients are the bare module's
    bare = _run(1, tmp_path, backend="gloo")[0]
    for k, ref in bare["grads"].items():
        assert (single["grads"][k] - ref).abs().max().item() <= 2e-5 * ref.abs().max().item(), k
    if torch.cuda.device_count() < 2:
        pytest.skip(f"{torch.cuda.device_count()} GPU visible: RCCL ran at world size 1 only; the 2-rank RCCL comparison needs 2 devices (NOT RUN)")
    _check_ddp(single, _run(2, tmp_path, backend="nccl"))


@pytest.mark.parametrize("workload", ["eval", "eval_shard_rays", "train", "train_monodepth2"])
def test_bench_runs_under_torch_distributed_run(workload):
    """The driver's multi-GPU launch line at N = 1: `python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1`.  bench.py
    initialises the process group whenever it is launched that way (backend nccl = RCCL, device_id), so its barrier / MAX all-reduce
    / DDP path executes here exactly as it will on a node; prints ONE JSON line with the contract's keys.  The training workloads wrap a
    REAL DistributedDataParallel at world size 1 (wrap_ddp(force=True)): reducer, bucket views and the RCCL all-reduce run through
    RenderFunction / ProjectFunction; `train_monodepth2` puts the shipped encoder's ~140 MB gradient bucket on it.  `eval` (the default
    line) carries `others` (every other BASELINE config + the occupancy profile) and `ddp_train` (KITTI-Raw shapes + Monodepth2)."""
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    extra = ["--workload", "train", "--encoder", "monodepth2"] if workload == "train_monodepth2" else ["--workload", workload]
    if workload == "eval_shard_rays":       # SURVEY 8e's second axis: ONE frame, its rays over the ranks, all-gather inside the timed region
        extra = ["--workload", "eval", "--shard", "rays", "--no-others"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port",
           str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["value"] > 0 and out["roofline"]["frac"] > 0
    if workload == "eval_shard_rays":
        assert out["config"]["parallelism"] == "rays x1 of one frame" and out["config"]["all_gather_bytes_per_step"] == 245760 * (4 + 3 * 64) * 4
        assert 0 < out["roofline"]["frac"] <= 1.0 and out["roofline"]["frac_algorithmic"] > 1.0
        return
    if workload == "eval":
        assert 0 < out["roofline"]["frac"] <= 1.0 and out["roofline"]["frac_algorithmic"] > 1.0
        assert set(out["others"]) == {"train", "kitti_raw", "re10k", "re10k_k128", "profile"}, out.get("others")
        for k, rec in out["others"].items():
            assert "error" not in rec and rec["value"] > 0 and rec["roofline"]["frac"] > 0 and rec["steps"] == 10, (k, rec)
        sub = out["ddp_train"]
        assert "error" not in sub and sub["value"] > 0, sub
        ar = sub["allreduce"]
    else:
        ar = out["allreduce"]
    # (one rank: RCCL may complete an in-place all-reduce without launching a kernel -- the reducer ran either way)
    assert ar["backend"] == "nccl" and ar["allreduce_ms"] >= 0 and ar["world"] == 1, ar
    if workload != "train":
        assert ar["bucket_bytes"] > 50e6, ar       # Monodepth2 (ResNet-50 encoder + decoder) + the MLP: the real gradient bucket
