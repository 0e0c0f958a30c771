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
