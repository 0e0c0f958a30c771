"""World-size-2 (and 3) gloo tests of the multi-GPU glue on CPU: ray sharding + all-gather reconstruct exactly what the un-sharded call
returns (ragged ray counts included), DDP over the renderer's own parameter modules averages gradients like one process on the
union of the data, and the flattened mean reduction is exact.  The HIP kernels are not involved (no GPU here): ``render`` is a
deterministic stand-in with the renderer's output dict layout."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from behindthescenes_amd import parallel
from behindthescenes_amd.mlp import ResnetFC


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_render(rays, want_weights=False):
    """per-ray outputs with the renderer's layout, a pure function of the ray"""
    sb, b, _ = rays.shape
    depth = rays[..., :3].norm(dim=-1) + rays[..., 6]
    out = dict(rgb=torch.stack((rays[..., 3], rays[..., 4], rays[..., 5]), -1).repeat(1, 1, 2), depth=depth,
               invalid=(rays[..., 0:1, None] > 0).float().expand(sb, b, 4, 2).contiguous())
    if want_weights:
        out["weights"] = rays[..., :4].abs()
    return {"coarse": out}


def _worker(rank, world, port, n_rays, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, dev = parallel.init_distributed("gloo")
    assert (r, w, dev.type) == (rank, world, "cpu")
    g = torch.Generator().manual_seed(0)
    rays = torch.randn(2, n_rays, 8, generator=g)
    full = _fake_render(rays, want_weights=True)["coarse"]
    got = parallel.render_sharded(_fake_render, rays, want_weights=True)["coarse"]
    ok = all(torch.equal(got[k], full[k]) for k in full) and set(got) == set(full)
    # DDP over the renderer's MLP module (same parameter names / shapes as in BTSNet): gradient = mean over ranks
    torch.manual_seed(1)
    mlp = ResnetFC(103, d_out=1, n_blocks=0, d_hidden=64)
    ddp = parallel.wrap_ddp(mlp)
    x = torch.randn(4 * world, 103, generator=g)
    s, e = parallel.shard_range(x.shape[0], rank, world)
    ddp(x[s:e]).square().mean().backward()
    ref = ResnetFC(103, d_out=1, n_blocks=0, d_hidden=64)
    ref.load_state_dict(mlp.state_dict())
    ref(x).square().mean().backward()
    ok_ddp = all(torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-7) for a, b in zip(mlp.parameters(), ref.parameters()))
    t = [torch.full((3,), float(rank)), torch.tensor(float(rank) * 2)]
    parallel.all_reduce_mean_(t)
    mean = (world - 1) / 2
    ok_mean = torch.allclose(t[0], torch.full((3,), mean)) and abs(t[1].item() - 2 * mean) < 1e-12
    results[rank] = (ok, ok_ddp, ok_mean)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_rays", [(2, 1000), (2, 37), (3, 10)])
def test_sharded_render_ddp_and_metric_reduction(world, n_rays):
    port = _free_port()
    with mp.Manager() as m:
        results = m.dict()
        mp.spawn(_worker, args=(world, port, n_rays, results), nprocs=world, join=True)
        assert len(results) == world
        for rank in range(world):
            assert results[rank] == (True, True, True), (rank, results[rank])


def _encoder_worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    parallel.init_distributed("gloo")
    from behindthescenes_amd.monodepth2 import Monodepth2
    torch.manual_seed(0)
    enc = Monodepth2(resnet_layers=18, num_ch_dec=[16, 16, 32, 64, 128], d_out=16, pretrained=False)
    ddp = parallel.wrap_ddp(enc)
    g = torch.Generator().manual_seed(rank)
    ok = True
    try:
        for _ in range(2):     # the reducer checks at the START of the second iteration that every gradient of the first arrived
            enc.zero_grad(set_to_none=True)
            x = torch.rand(1, 3, 64, 96, generator=g) * 2 - 1
            sum(f.square().mean() for f in ddp(x)).backward()
    except RuntimeError as e:
        ok = str(e)[:300]
    results[rank] = ok
    dist.destroy_process_group()


def test_two_ddp_steps_with_the_shipped_monodepth2_encoder():
    """ResNet.fc is kept for strict checkpoint loading but never evaluated; as a trainable parameter it made
    DistributedDataParallel raise 'Expected to have finished reduction in the prior iteration' in the second step."""
    port = _free_port()
    with mp.Manager() as m:
        results = m.dict()
        mp.spawn(_encoder_worker, args=(2, port, results), nprocs=2, join=True)
        assert dict(results) == {0: True, 1: True}, dict(results)


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 64, 122880):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(e - s for s, e in spans) - min(e - s for s, e in spans) <= 1


def test_bench_launch_glue_under_torch_distributed_run_world_2():
    """The driver's multi-GPU launch line -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W` -- at N = 2 on this CPU container: bench.py's `--workload glue` runs the launch
    code every workload shares (launch_context's env parsing, init_group, timed_region's barriers and MAX all-reduce, the parallelism
    strings, rank 0 printing ONE JSON line) around a stand-in step; without CUDA the group is gloo.  Rank r's step sleeps (r + 1) ms: the
    reported time must be the SLOWER rank's (the MAX over the ranks), value = the units of ALL ranks over it."""
    import json
    import subprocess
    import sys
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "2", "--workload", "glue"]
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]            # rank 0 only
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in out, k
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["warmup"] == 2 and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert out["config"]["backend"] == "gloo" and out["config"]["parallelism"] == "frames x2"
    assert 2.0 <= out["ms_per_step"] < 20.0, out["ms_per_step"]            # rank 1 sleeps 2 ms per step, rank 0 only 1 ms: the MAX
    assert abs(out["value"] - 2 * 20 / (out["ms_per_step"] * 20e-3)) <= 1e-6 * out["value"]
    # the pieces, in process
    assert bench.launch_context({}) == (1, 0, 0, False)
    assert bench.launch_context(dict(RANK="3", LOCAL_RANK="1", WORLD_SIZE="8", MASTER_PORT="1")) == (8, 3, 1, True)
    assert bench.launch_context(dict(RANK="0", WORLD_SIZE="1")) == (1, 0, 0, False)        # no rendezvous port: not a distributed launch
    assert [bench.parallelism(k, 8) for k in ("frames", "rays", "batch")] == ["frames x8", "rays x8 of one frame", "batch x8"]
