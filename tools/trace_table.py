"""Per-kernel table of a rocprofv3 `--kernel-trace --stats` CSV: total ms, calls, average, share -- every row, names shortened.
    python tools/trace_table.py <kernel_stats.csv> [steps] [top]
`steps` (the traced command's warm-up + timed steps) turns totals into ms per step."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
tot = sum(float(r["TotalDurationNs"]) for r in rows)


def short(n):
    n = re.sub(r"\(.*", "", n)
    n = re.sub(r"void |at::native::|\(anonymous namespace\)::", "", n)
    return n[:96]


print(f"{'ms/step':>9s} {'calls/step':>10s} {'avg ms':>9s} {'share':>6s}  kernel     (total {tot / 1e6 / steps:.3f} ms per step over {steps:g} steps)")
groups = {"bts::": 0.0, "miopen / conv": 0.0, "rccl": 0.0, "other": 0.0}
for r in rows:
    t = float(r["TotalDurationNs"])
    n = r["Name"]
    k = "bts::" if "bts::" in n else ("miopen / conv" if re.search(r"miopen|MIOpen|conv|Conv|igemm|gemm|Cijk|naive_|im2col|xdlops|winograd|batchnorm|BatchNorm|SubTensor|Op1dTensor|Op2dTensor|pool", n, re.I) else
                                       ("rccl" if re.search(r"nccl|rccl", n, re.I) else "other"))
    groups[k] += t
for r in rows[:top]:
    t = float(r["TotalDurationNs"])
    print(f"{t / 1e6 / steps:9.4f} {int(r['Calls']) / steps:10.1f} {float(r['AverageNs']) / 1e6:9.4f} {100 * t / tot:5.1f}%  {short(r['Name'])}")
print("groups (ms per step):", {k: round(v / 1e6 / steps, 3) for k, v in groups.items()})
