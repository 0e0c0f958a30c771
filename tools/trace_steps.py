"""Steady-state per-kernel table of a training / evaluation loop from a rocprofv3 `--kernel-trace` CSV (one row per dispatch).

    python tools/trace_steps.py <kernel_trace.csv> <marker> [last_steps]

`marker` = substring of a kernel that runs exactly ONCE per step (e.g. patch_rays_kernel for the training workloads, gen_rays_kernel for
the eval frame): its dispatches delimit the steps.  Only the LAST `last_steps` complete steps (default 3) are aggregated, so warm-up
work -- MIOpen's solver search runs every candidate, naive ones included, and would otherwise dominate a --stats summary -- stays out.
Prints ms per step per kernel (names shortened), the sums per group (bts:: / convolution + GEMM / RCCL / other torch) and the GPU-busy
fraction of the step (kernel time / wall time between the step delimiters)."""
import csv
import re
import sys
from collections import defaultdict

path, marker = sys.argv[1], sys.argv[2]
last = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
if len(marks) < last + 1:
    sys.exit(f"only {len(marks)} dispatches of a kernel matching {marker!r}: need {last + 1}")
lo, hi = marks[-(last + 1)], marks[-1]
sel = rows[lo:hi]
wall = (rows[hi][0] - rows[lo][0]) / 1e6 / last


def short(n):
    n = re.sub(r"\(.*", "", n)
    n = re.sub(r"void |at::native::|\(anonymous namespace\)::", "", n)
    return n[:100]


def group(n):
    if "bts::" in n:
        return "bts::"
    if re.search(r"nccl|rccl", n, re.I):
        return "rccl"
    if re.search(r"miopen|conv|igemm|gemm|Cijk|naive_|im2col|col2im|xdlops|winograd|batchnorm|BatchNorm|SubTensor|Op1dTensor|Op2dTensor|pool|transpose", n, re.I):
        return "convolution / GEMM / MIOpen"
    return "other torch"


per, calls, grp = defaultdict(float), defaultdict(int), defaultdict(float)
for s, e, n in sel:
    per[n] += (e - s) / 1e6
    calls[n] += 1
    grp[group(n)] += (e - s) / 1e6
busy = sum(per.values()) / last
print(f"# {path}: last {last} steps delimited by {marker!r}: {wall:.3f} ms wall per step, {busy:.3f} ms of kernels per step "
      f"(GPU busy {busy / wall:.2f}), {len(sel) / last:.0f} dispatches per step")
print(f"{'ms/step':>9s} {'calls/step':>10s} {'avg ms':>9s} {'share':>6s}  kernel")
for n, t in sorted(per.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{t / last:9.4f} {calls[n] / last:10.1f} {t / calls[n]:9.4f} {100 * t / last / busy:5.1f}%  {short(n)}")
print("groups (ms per step):", {k: round(v / last, 3) for k, v in sorted(grp.items(), key=lambda kv: -kv[1])})
