"""One-screen summary of a bench.py JSON line (headline, roofline, others).   python tools/bench_summary.py <file>"""
import json, sys
lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not lines:
    sys.exit("no JSON line in " + sys.argv[1])
j = json.loads(lines[-1])


def r4(d):
    return {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items() if isinstance(v, (int, float)) and not isinstance(v, bool)}


print(j["metric"][:60], "| value", round(j["value"], 1), j["unit"], "| ms/step", round(j["ms_per_step"], 4))
print("  roofline", r4(j.get("roofline", {})))
for k, v in (j.get("others") or {}).items():
    if "error" in v:
        print(" ", k, "ERROR", v["error"][-300:])
        continue
    print(" ", k, "ms/step", round(v["ms_per_step"], 4), "value", round(v["value"], 1), "|", r4(v.get("roofline", {})))
for k in ("cpu_baseline", "ref_gpu_baseline", "ddp_train", "allreduce", "shard_rays"):
    if k in j:
        print(" ", k, {a: b for a, b in j[k].items() if a in ("value", "cores", "ms_per_step", "allreduce_ms", "ours_over_ref", "unit")})
