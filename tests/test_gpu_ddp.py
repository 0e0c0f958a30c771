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


def _run(world, out_dir):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_ddp_worker.py"), str(out_dir)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [torch.load(os.path.join(out_dir, f"rank{r}_of{world}.pt")) for r in range(world)]


def test_ddp_gradients_equal_the_single_process_union_batch_and_sharded_render_is_exact(tmp_path):
    single = _run(1, tmp_path)[0]
    ranks = _run(2, tmp_path)
    assert all(all(r["same"].values()) for r in ranks + [single]), [r["same"] for r in ranks]
    for k, ref in single["grads"].items():
        a, b = ranks[0]["grads"][k], ranks[1]["grads"][k]
        assert torch.equal(a, b), f"{k}: ranks hold different gradients after the all-reduce"
        err = (a - ref).abs().max().item() / ref.abs().max().item()
        assert err <= 2e-5, (k, err)       # float atomics / bucket summation order: not bit-exact, far below the 1e-4 gradient bar
    assert abs((ranks[0]["loss"] + ranks[1]["loss"]) / 2 - single["loss"]) <= 1e-5 * abs(single["loss"])
