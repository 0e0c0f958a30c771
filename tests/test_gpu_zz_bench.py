"""GPU, collected LAST (tests/conftest.py orders the files; the name sorts last too): bench.py launched the way the driver launches it.
These cases start whole bench processes and assert on the printed line's metadata -- a hiccup here must never stand between `pytest -x`
and a parity file (round 5: one stale literal in this test left 136 parity tests unrun on the driver's box)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("workload", ["eval", "eval_shard_rays", "train", "train_monodepth2"])
def test_bench_runs_under_torch_distributed_run(workload):
    """The driver's multi-GPU launch line at N = 1: `python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1`.  bench.py
    initialises the process group whenever it is launched that way (backend nccl = RCCL, device_id), so its barrier / MAX all-reduce
    / DDP path executes here exactly as it will on a node; prints ONE JSON line with the contract's keys.  The training workloads wrap a
    REAL DistributedDataParallel at world size 1 (wrap_ddp(force=True)): reducer, bucket views and the RCCL all-reduce run through
    RenderFunction / ProjectFunction; `train_monodepth2` puts the shipped encoder's ~140 MB gradient bucket on it.  `eval` (the default
    line) carries `others` (every other BASELINE config + the occupancy profile) and `ddp_train` (KITTI-Raw shapes + Monodepth2)."""
    import json
    import bench          # the child-run step counts are bench.py's constants, never a literal here (round-5 verdict, weak 1)
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
            assert "error" not in rec and rec["value"] > 0 and rec["roofline"]["frac"] > 0 and rec["steps"] == bench.CHILD_STEPS and rec["warmup"] == bench.CHILD_WARMUP, (k, rec)
        sub = out["ddp_train"]
        assert "error" not in sub and sub["value"] > 0, sub
        ar = sub["allreduce"]
    else:
        ar = out["allreduce"]
    # (one rank: RCCL may complete an in-place all-reduce without launching a kernel -- the reducer ran either way)
    assert ar["backend"] == "nccl" and ar["allreduce_ms"] >= 0 and ar["world"] == 1, ar
    if workload != "train":
        assert ar["bucket_bytes"] > 50e6, ar       # Monodepth2 (ResNet-50 encoder + decoder) + the MLP: the real gradient bucket
