"""Builds libbts_render.so (hand-written HIP for gfx950) in-tree with hipcc.  No GPU is needed to build.

    python -m behindthescenes_amd.build [--force]

The shared object is git-ignored but travels to the GPU box with the source snapshot.
"""
import hashlib
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, "libbts_render.so")
VARIANTS = os.path.join(PKG, "variants")
# objects, saved assembly and the digest stamp live outside the repo (they are large and must not travel to the GPU box)
OBJ = os.path.join(os.environ.get("BTS_OBJ_DIR", "/tmp"), "bts_render_obj")
SOURCES = ["bts_fwd.hip", "bts_fwd_proj.hip", "bts_fwd_epi.hip", "bts_query.hip", "bts_bwd.hip", "bts_bwd_rows.hip", "bts_bwd_blocks.hip", "bts_prep.hip", "bts_aux.hip", "bts_loss.hip", "bts_train.hip", "bts_conv.hip", "bts_api.hip"]
# -fno-slp-vectorize: hipcc's SLP vectoriser builds v_pk_*_f32 with op_sel:[x,1], which MI355X evaluates wrongly in lanes 48-63 next
# to a wide MFMA (tools/check_pk_opsel.py, tools/ubench/pk_opsel_lanes.hip); explicit float2 code keeps the packed FMAs that matter
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-fno-gpu-rdc",
         "-munsafe-fp-atomics"]


def _digest():
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)) + [os.path.join("..", "..", "include", "bts_render.h")]:
        path = os.path.join(CSRC, name)
        if os.path.isfile(path) and name.split(".")[-1] in ("hip", "h"):
            h.update(name.encode()), h.update(open(path, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def build_library(force: bool = False, verbose: bool = False, probe: bool = False, tag: str = "", extra_flags=()) -> str:
    """probe=True builds variants/libbts_probe.so: the same sources with -DBTS_PROBE (section-ablation hooks for tools/ablate_probe.py;
    never loaded by the product path).  tag="x" builds behindthescenes_amd/variants/libbts_x.so with `extra_flags` appended (or
    replacing -O3 when an -O level is given): differently scheduled builds of the same sources for the second-schedule parity
    tests and for hazard bisection; loaded only through BTS_RENDER_LIB."""
    lib = os.path.join(VARIANTS, "libbts_probe.so") if probe else LIB      # (never in the package directory: the product has ONE library)
    if probe:
        os.makedirs(VARIANTS, exist_ok=True)
    obj_dir = OBJ + ("_probe" if probe else "")
    extra = list(extra_flags) + [f for f in os.environ.get("BTS_EXTRA_FLAGS", "").split() if f]
    flags = FLAGS + (["-DBTS_PROBE"] if probe else [])
    if any(re.fullmatch(r"-O[0-3sz]", f) for f in extra):
        flags = [f for f in flags if not re.fullmatch(r"-O[0-3sz]", f)]
    flags = flags + extra
    if tag:
        os.makedirs(VARIANTS, exist_ok=True)
        lib = os.path.join(VARIANTS, f"libbts_{tag}.so")
        obj_dir = OBJ + "_" + tag
    stamp = os.path.join(obj_dir, "digest.txt")
    dig = _digest() + hashlib.sha256(" ".join(flags).encode()).hexdigest()
    if not force and os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return lib
    os.makedirs(obj_dir, exist_ok=True)
    cc = hipcc()

    def compile_one(src):
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        cmd = [cc] + flags + ["-save-temps=obj", "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
        # lint the device assembly for the packed-FP32 operand-select form MI355X gets wrong (tools/check_pk_opsel.py)
        asm = obj[:-2] + "-hip-amdgcn-amd-amdhsa-gfx950.s"
        if os.path.exists(asm) and "-DBTS_ALLOW_PK_OPSEL" not in flags:
            chk = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_pk_opsel.py"), asm], capture_output=True, text=True)
            if chk.returncode != 0:
                raise RuntimeError(f"{src}: packed-FP32 instruction with op_sel[src1] = 1 (gfx950 erratum, tools/check_pk_opsel.py):\n{chk.stdout[-3000:]}")
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    r = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    open(stamp, "w").write(dig)
    if verbose:
        print(f"built {lib}")
    return lib


# differently scheduled builds of the product sources (tests/test_gpu_determinism.py re-runs parity against them)
SCHEDULE_VARIANTS = {"o2": ["-O2"], "regionbarrier": ["-DBTS_REGION_BARRIER"], "gatherregs": ["-DBTS_GATHER_REGS"],
                     "fetchearly": ["-DBTS_GL_FETCH_EARLY"]}
# the erratum on purpose: the round-1 flags (SLP vectoriser on) -- tests/test_gpu_determinism.py shows this build is NOT deterministic
ERRATUM_VARIANT = ("slp", ["-fslp-vectorize", "-DBTS_ALLOW_PK_OPSEL"])


def build_variants(names=None, force=False, verbose=False):
    return [build_library(force=force, verbose=verbose, tag=n, extra_flags=SCHEDULE_VARIANTS[n]) for n in (names or SCHEDULE_VARIANTS)]


if __name__ == "__main__":
    if "--tag" in sys.argv:
        i = sys.argv.index("--tag")
        print(build_library(force="--force" in sys.argv, verbose=True, tag=sys.argv[i + 1], extra_flags=sys.argv[i + 2:]))
    elif "--variants" in sys.argv:
        print(build_variants(force="--force" in sys.argv, verbose=True))
    else:
        print(build_library(force="--force" in sys.argv, verbose=True, probe="--probe" in sys.argv))
