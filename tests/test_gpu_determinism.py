"""Determinism and second-schedule tests (round-2 verdict item 1).

The forward has no atomics, so N runs on the same inputs must be BIT-identical -- for every NVMAX instantiation (nv = 1, 2, 4, 8)
and both ray mappings (one ray per wave, several short rays per wave).  Round 1 shipped two "schedule-dependent wrong-result
hazards"; both were one hardware erratum (packed-FP32 operand select next to a wide MFMA, tools/ubench/pk_opsel_lanes.hip,
DESIGN.md section 3): wrong values in lanes 48-63, timing-dependent.  These tests would have caught it:
  * bit equality over 20 runs at the BASELINE configs[1] shape;
  * the parity suite re-run against differently scheduled builds of the same sources (-O2, per-region scheduling fences);
  * the micro-benchmark itself: the safe forms stay exact, the lint refuses the unsafe one."""
import os
import re
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNS = 20


@pytest.fixture(scope="module")
def hip():
    import behindthescenes_amd as bts
    from behindthescenes_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    _lib.load()
    return bts


def _scene(hip, n, v, H, W, C, Hd, nb, K, ids, n_rays, seed, **conf):
    from behindthescenes_amd import synthetic as S
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)   # BTSNet draws its empty_feature from the global generator
    scene = S.synthetic_scene(n, v, H, W, C, seed=seed, intrinsics=S.K_KITTIRAW if C == 64 else S.K_RE10K, smooth=True)
    net = S.build_net(scene, d_hidden=Hd, n_blocks=nb, ids_render=ids, device="cuda", mlp_seed=seed, **conf)
    zn, zf = conf.get("z_near", 3.0), conf.get("z_far", 80.0)
    rays = hip.ImageRaySampler(zn, zf, H, W).sample(None, scene["poses"].cuda(), scene["projs"].cuda())[0]
    if n_rays is not None:
        idx = torch.randperm(rays.shape[1], generator=g)[:n_rays].sort().values.cuda()
        rays = rays[:, idx].contiguous()
    rays = rays.reshape(-1, 8).contiguous()
    renderer = hip.NeRFRenderer(n_coarse=K, lindisp=True, hard_alpha_cap=(C == 64)).cuda().eval()
    z = renderer.sample_coarse(rays, torch.rand(rays.shape[0], K, generator=g).cuda())
    return net, renderer, rays, z


CASES = {  # name: (n, v, H, W, C, Hd, nb, K, ids_render, n_rays, conf)      one-ray-per-wave iff K > 32
    "cfg2_nv1_oneray": (1, 2, 192, 640, 64, 64, 0, 64, [0], None, {}),               # BASELINE configs[1] at full size
    "cfg2_nv1_learn_empty": (1, 2, 192, 640, 64, 64, 0, 64, [0], 60000, dict(learn_empty=True)),
    "nv2_oneray": (1, 3, 192, 640, 64, 64, 0, 64, [1, 2], 40000, {}),
    "nv4_oneray": (2, 5, 192, 640, 64, 64, 0, 64, [1, 2, 3, 4], 16384, {}),            # KITTI-360 training shape
    "nv8_oneray": (1, 9, 96, 320, 64, 64, 0, 64, list(range(1, 9)), 8192, {}),
    "nv1_shared_wave": (2, 2, 96, 320, 64, 64, 0, 32, [1], 16384, {}),                 # K = 32: two rays per wave
    "nv2_shared_wave": (2, 3, 96, 320, 64, 64, 0, 16, [1, 2], 16384, {}),              # K = 16: four rays per wave
    "nv4_shared_wave": (1, 5, 96, 320, 64, 64, 0, 8, [1, 2, 3, 4], 16384, {}),         # K = 8: eight rays per wave
    "k24_idle_lanes": (1, 2, 96, 320, 64, 64, 0, 24, [1], 16384, {}),                 # K = 24: 8 idle lanes per ray (two rays per wave)
    "k12_idle_lanes_re10k": (2, 3, 64, 96, 32, 32, 1, 12, [1, 2], 4096, dict(z_near=1.0, z_far=100.0, code_mode="distance")),
    "k48_idle_lanes": (1, 2, 96, 320, 64, 64, 0, 48, [0], 8192, {}),                   # K = 48: 16 idle lanes (one ray per wave)
    "re10k_nv2": (2, 3, 256, 384, 32, 32, 1, 48, [1, 2], 24576, dict(z_near=1.0, z_far=100.0, code_mode="distance")),
    "re10k_k128": (1, 3, 256, 384, 32, 32, 1, 128, [1, 2], 8192, dict(z_near=1.0, z_far=100.0, code_mode="distance")),
}


@pytest.mark.parametrize("name", list(CASES))
def test_forward_is_bit_deterministic(hip, name):
    n, v, H, W, C, Hd, nb, K, ids, n_rays, conf = CASES[name]
    net, renderer, rays, z = _scene(hip, n, v, H, W, C, Hd, nb, K, ids, n_rays, seed=40 + len(name), **conf)
    ref = None
    # the RE10K instantiations are where the one order of the gather ring that is NOT shipped lost its run-to-run determinism in round 2
    # (1 - 23 of 24 576 rays, i.e. ~1e-4 per ray and launch): 200 launches each instead of 20
    runs = int(os.environ.get("BTS_DETERMINISM_RUNS", 200 if name.startswith("re10k") or name.endswith("re10k") else RUNS))
    with torch.no_grad():
        for r in range(runs):
            out = renderer.composite(net, rays, z, sb=n)
            out = [t for t in out if t is not None]
            if ref is None:
                ref = [t.clone() for t in out]
                assert all(torch.isfinite(t).all() for t in ref)
                continue
            for i, (a, b) in enumerate(zip(ref, out)):
                if not torch.equal(a, b):
                    bad = (a != b).reshape(a.shape[0], -1).any(-1).nonzero().flatten()
                    where = (a != b).nonzero()[:8].tolist()
                    raise AssertionError(f"{name}: output {i} of run {r} differs from run 0 in {bad.numel()} rays, first at {where}")


def _variant(tag):
    path = os.path.join(ROOT, "behindthescenes_amd", "variants", f"libbts_{tag}.so")
    assert os.path.exists(path), f"{path} missing: __graft_entry__.build() builds the schedule variants"
    return path


@pytest.mark.parametrize("tag", ["o2", "regionbarrier"])
def test_parity_holds_on_a_second_schedule(tag):
    """The reference goldens, the fp64 arbiter, the ragged / training shapes and the gradient goldens against a differently scheduled
    build of the same sources (its own process: the library is chosen at load time through BTS_RENDER_LIB)."""
    env = dict(os.environ, BTS_RENDER_LIB=_variant(tag), BTS_ALLOW_LIB_OVERRIDE="1")
    sel = "(golden or test_fp64_arbiter or ragged_and_training or single_ray) and not real_training"
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-k", sel, "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_grad.py")],
                       env=env, capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert re.search(r"\b\d+ passed", r.stdout), r.stdout[-500:]


@pytest.mark.parametrize("tag,name", [("o2", "cfg2_nv1_learn_empty"), ("regionbarrier", "cfg2_nv1_learn_empty"),
                                      ("gatherregs", "cfg2_nv1_learn_empty"), ("fetchearly", "cfg2_nv1_learn_empty"),
                                      ("fetchearly", "re10k_nv2")])
def test_second_schedule_matches_the_shipped_build_bit_for_bit(hip, tag, name, tmp_path):
    """Same expression tree, -ffp-contract=off: a different instruction schedule must not change a single bit.  gatherregs: the
    round-1 gather (two register buffers per lane) against the shipped gather through LDS -- same blend order, same bits.
    fetchearly: the other order of the gather ring's step (rows of block T + 1 requested before block T + 3 goes out, the order
    rounds 1 - 2 shipped) -- on the RE10K shape too, where the late order used to differ from run to run before gl_issue took one
    dependency per row piece (tools/ubench/lds_dma_overtake.hip)."""
    code = f"""
import sys, torch
sys.path.insert(0, {ROOT!r})
import behindthescenes_amd as bts
from tests.test_gpu_determinism import _scene, CASES
n, v, H, W, C, Hd, nb, K, ids, n_rays, conf = CASES[{name!r}]
net, renderer, rays, z = _scene(bts, n, v, H, W, C, Hd, nb, K, ids, n_rays, seed=7, **conf)
with torch.no_grad():
    out = [t.cpu() for t in renderer.composite(net, rays, z, sb=n) if t is not None]
torch.save(out, sys.argv[1])
"""
    outs = []
    for lib in (None, _variant(tag)):
        f = tmp_path / f"{'shipped' if lib is None else tag}.pt"
        env = dict(os.environ)
        env.pop("BTS_RENDER_LIB", None)
        if lib:
            env["BTS_RENDER_LIB"], env["BTS_ALLOW_LIB_OVERRIDE"] = lib, "1"
        r = subprocess.run([sys.executable, "-c", code, str(f)], env=env, capture_output=True, text=True, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(f))
    for i, (a, b) in enumerate(zip(*outs)):
        assert torch.equal(a, b), f"output {i}: {(a != b).sum().item()} of {a.numel()} values differ between the shipped build and {tag}"


def test_packed_fp32_erratum_microbenchmark():
    """tools/ubench/pk_opsel_lanes: every operand selection the shipped code can contain is exact next to wide MFMAs; the form the
    lint refuses (low result <- src1's high register) is the one that fails, in lanes 48-63 only (reported, not asserted: a later
    hardware stepping may fix it)."""
    exe = os.path.join(ROOT, "tools", "ubench", "pk_opsel_lanes")
    assert os.path.exists(exe), "tools/ubench/pk_opsel_lanes missing: __graft_entry__.build() compiles it"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1000:]
    rows, name = {}, None
    for line in r.stdout.splitlines():
        if line.startswith("v_pk_"):
            name = line.strip()
        m = re.match(r"\s+f16 MFMA \+ v_pk_fma \+ ds_read\s+(\d+)/(\d+)/(\d+)/(\d+) \| (\d+)/(\d+)/(\d+)/(\d+)", line)
        if m and name and name not in rows:
            rows[name] = [int(x) for x in m.groups()]
    assert len(rows) >= 20, r.stdout[-2000:]
    unsafe = {k: v for k, v in rows.items() if re.search(r"op_sel:\[[01],1", k)}
    safe = {k: v for k, v in rows.items() if k not in unsafe}
    for k, v in safe.items():
        assert sum(v) == 0, f"{k}: a form the lint allows returned wrong values {v}"
    for k, v in unsafe.items():
        assert v[0] == v[1] == v[2] == 0 and sum(v[4:]) == 0, f"{k}: errors outside the low result of lanes 48-63: {v}"
    print("unsafe forms, wrong low results in lanes 48-63:", {k: v[3] for k, v in unsafe.items()})
