#!/bin/bash
# rocprofv3 passes over ONE WHOLE fused training step as bench.py runs it (bts_train_step_fwd / _bwd: every bts:: kernel of the step):
#   gpurun -- bash tools/profile_step.sh <tag> <train|kitti_raw|re10k> [K | ""] [nhwc|nchw]
# -> gpurun_out/prof_<tag>/traffic_step_<workload>[_k<K>].json: per kernel launches per step, average duration, HBM fetch / write bytes per
# launch and per step (FETCH_SIZE doubled on gfx950, MI355X_MICROARCH.md HBM section), L2 hit rate; `step` = the sums over the step's bts::
# kernels.  Counters in their own --pmc passes with --kernel-trace only; the trace pass gives the durations.  bench.py reads the newest
# profiles/<tag>/traffic_step_*.json and prints `traffic`, `algorithmic_bytes`, `traffic_ratio` from it.
set -u
TAG=${1:-r06}
WL=${2:-train}
KK=${3:-}
LAYOUT=${4:-nhwc}      # the stand-in maps' memory format (bench.py --feat-layout)
REPO=$(pwd)
TOP=$REPO/gpurun_out/prof_$TAG
OUT=$TOP/raw_step_${WL}${KK:+_k$KK}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
STEPS=4; WARM=2
CMD="python $REPO/bench.py --workload $WL ${KK:+--samples $KK} --steps $STEPS --warmup $WARM --no-cpu-baseline --no-others --no-other-layout --feat-layout $LAYOUT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1 || tail -5 $OUT/trace.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $OUT/pmc2 -o pmc2 -- $CMD > $OUT/pmc2.log 2>&1 || tail -5 $OUT/pmc2.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $OUT/pmc3 -o pmc3 -- $CMD > $OUT/pmc3.log 2>&1 || tail -5 $OUT/pmc3.log
cd $REPO
find $OUT -type f ! -name "*.csv" ! -name "*.log" -delete
python - <<PY
import csv, glob, collections, json, re
steps = $STEPS + $WARM
short = lambda n: re.sub(r"\(.*", "", n.replace("void ", "")).strip()
dur = {}
for f in sorted(glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        dur[short(row["Name"])] = (float(row["AverageNs"]), int(row["Calls"]))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        agg[short(row.get("Kernel_Name", ""))][row["Counter_Name"]].append(float(row["Counter_Value"]))
per, tot = {}, dict(fetch_bytes=0.0, write_bytes=0.0, kernel_ms=0.0, launches=0.0)
for name, c in sorted(agg.items()):
    if not name.startswith("bts::") or name not in dur:
        continue
    ns, calls = dur[name]
    m = {k: sum(v) / len(v) for k, v in c.items()}
    lps = calls / steps                       # launches per step
    e = dict(launches_per_step=lps, kernel_ms=ns / 1e6, fetch_bytes=2 * m.get("FETCH_SIZE", 0) * 1024, write_bytes=m.get("WRITE_SIZE", 0) * 1024)
    if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
        e["l2_hit"] = m["TCC_HIT_sum"] / max(m["TCC_HIT_sum"] + m["TCC_MISS_sum"], 1)
    e["bytes_per_step"] = lps * (e["fetch_bytes"] + e["write_bytes"])
    e["tb_per_s"] = (e["fetch_bytes"] + e["write_bytes"]) / (ns * 1e-9) / 1e12
    per[name] = e
    tot["fetch_bytes"] += lps * e["fetch_bytes"]; tot["write_bytes"] += lps * e["write_bytes"]; tot["kernel_ms"] += lps * ns / 1e6; tot["launches"] += lps
out = dict(workload="$WL", K="$KK" or None, feat_layout="$LAYOUT", steps_profiled=steps, step=tot, kernels=per,
           note="rocprofv3 --pmc passes of bench.py --workload $WL (FusedTrainStep: bts_train_step_fwd / _bwd); per-launch averages over every dispatch "
                "of the run; FETCH_SIZE (KB) x 2 on gfx950 per MI355X_MICROARCH.md; launches_per_step = calls / (steps + warmup)")
name = "$TOP/traffic_step_$WL" + ("_k$KK" if "$KK" else "") + ("" if "$LAYOUT" == "nhwc" else "_$LAYOUT") + ".json"
json.dump(out, open(name, "w"), indent=1)
print(f"step: {tot['kernel_ms']:.3f} ms of bts:: kernels, {tot['launches']:.0f} launches, fetch {tot['fetch_bytes'] / 1e6:.0f} MB + write {tot['write_bytes'] / 1e6:.0f} MB")
for k, e in sorted(per.items(), key=lambda kv: -kv[1]["launches_per_step"] * kv[1]["kernel_ms"])[:10]:
    print(f"  {e['launches_per_step']:4.1f} x {e['kernel_ms']:.4f} ms  {e['fetch_bytes'] / 1e6:8.1f} + {e['write_bytes'] / 1e6:8.1f} MB  {e['tb_per_s']:.2f} TB/s  L2 hit {e.get('l2_hit', float('nan')):.2f}  {k[:70]}")
PY
