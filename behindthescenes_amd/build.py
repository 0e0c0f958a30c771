"""Builds libbts_render.so (hand-written HIP for gfx950) in-tree with hipcc.  No GPU is needed to build.

    python -m behindthescenes_amd.build [--force]

The shared object is git-ignored but travels to the GPU box with the source snapshot.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, "libbts_render.so")
# objects, saved assembly and the digest stamp live outside the repo (they are large and must not travel to the GPU box)
OBJ = os.path.join(os.environ.get("BTS_OBJ_DIR", "/tmp"), "bts_render_obj")
SOURCES = ["bts_fwd.hip", "bts_fwd_proj.hip", "bts_bwd.hip", "bts_prep.hip", "bts_aux.hip", "bts_loss.hip", "bts_api.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fno-gpu-rdc", "-munsafe-fp-atomics"]


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


def build_library(force: bool = False, verbose: bool = False, probe: bool = False) -> str:
    """probe=True builds libbts_probe.so: the same sources with -DBTS_PROBE (section-ablation hooks for tools/ablate_probe.py;
    never loaded by the product path)."""
    lib = LIB.replace("libbts_render", "libbts_probe") if probe else LIB
    obj_dir = OBJ + ("_probe" if probe else "")
    flags = FLAGS + (["-DBTS_PROBE"] if probe else []) + [f for f in os.environ.get("BTS_EXTRA_FLAGS", "").split() if f]
    stamp = os.path.join(obj_dir, "digest.txt")
    dig = _digest()
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
        # lint the device assembly: an MFMA whose destination overlaps its own A/B sources computes garbage on MI355X
        asm = obj[:-2] + "-hip-amdgcn-amd-amdhsa-gfx950.s"
        if os.path.exists(asm):
            chk = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_mfma_overlap.py"), asm], capture_output=True, text=True)
            if chk.returncode != 0:
                raise RuntimeError(f"{src}: MFMA destination overlaps a source operand (see bts_common.h zero_acc):\n{chk.stdout[-3000:]}")
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


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True, probe="--probe" in sys.argv))
