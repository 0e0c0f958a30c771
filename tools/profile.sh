#!/bin/bash
# rocprofv3 passes over the render kernels.  Run on the GPU box via gpurun:
#   gpurun -- bash tools/profile.sh <tag> [fwd|train|bwd|bwd_re10k [K]|bwd_kitti_raw|profile|conv]
# fwd (default): tools/kernel_probe.py = the bench.py kernel (BASELINE configs[1]); train: tools/train_probe.py = configs[2] shapes;
# bwd: tools/bwd_probe.py = bts_render_bwd alone on the configs[2] shape; bwd_re10k [K] / bwd_kitti_raw: the same on configs[4] / [3];
# profile: bench.py --workload profile (the occupancy-grid query kernel).
# Writes raw output under gpurun_out/prof_<tag>/ and a summary (traffic.json with the derived fractions bench.py reports);
# copy what you want judged into profiles/<tag>/.  Counters are collected in separate --pmc passes with --kernel-trace only.
set -u
TAG=${1:-r02}
MODE=${2:-fwd}
REPO=$(pwd)
TOP=$REPO/gpurun_out/prof_$TAG                 # the summaries (traffic*.json) of every mode of a tag collect here ...
OUT=$TOP/raw_${MODE}${3:+_k$3}                 # ... the raw CSVs of a mode in its own directory
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
if [ "$MODE" = train ]; then CMD="python $REPO/tools/train_probe.py 16 3"; elif [ "$MODE" = bwd ]; then CMD="python $REPO/tools/bwd_probe.py 3";
elif [ "$MODE" = bwd_re10k ]; then CMD="python $REPO/tools/bwd_probe.py 3 re10k ${3:-48}"; elif [ "$MODE" = bwd_kitti_raw ]; then CMD="python $REPO/tools/bwd_probe.py 3 kitti_raw";
elif [ "$MODE" = profile ]; then CMD="python $REPO/bench.py --workload profile --steps 3 --warmup 1 --no-cpu-baseline";
elif [ "$MODE" = conv ]; then CMD="python $REPO/tools/conv_probe.py 3";
else CMD="python $REPO/tools/kernel_probe.py 5"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1 || tail -5 $OUT/trace.log
for i in 0 1 2 3 4; do
  case $i in
    0) PMC="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" ;;
    1) PMC="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" ;;
    2) PMC="FETCH_SIZE TCC_HIT_sum" ;;
    3) PMC="WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" ;;
    4) PMC="GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS TCP_TCC_READ_REQ_sum TA_BUSY_avr" ;;
  esac
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -o pmc$i -- $CMD > $OUT/pmc$i.log 2>&1 || tail -5 $OUT/pmc$i.log
done
cd $REPO
find $OUT -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" -delete
du -sh $OUT; tail -2 $OUT/trace.log
python - <<PY
import csv, glob, collections, json, re
mode = "$MODE"
# kernels of interest and their average duration (ns) from the stats pass
dur = {}
for f in sorted(glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        dur[row["Name"]] = (float(row["AverageNs"]), int(row["Calls"]))
        if len(dur) <= 10:
            print(f"{float(row['AverageNs']) / 1e6:9.4f} ms x {row['Calls']:>4s}  {row['Name'][:110]}")
train = ["rows_kernel", "scatter_kernel", "dwpe_kernel", "render_bwd_kernel", "scatter_dg_kernel", "render_kernel_p", "project_kernel",
         "project_bwd_kernel", "photometric_loss_kernel"]
rows = ["rowsb_kernel", "scatter_kernel", "dwpe_rows_kernel", "render_kernel_p"]
pats = {"conv": ["conv_fwd", "conv_dgrad", "conv_wgrad_", "elu_bwd_kernel"], "fwd": ["render_kernel_p"], "train": train, "bwd": train[:3], "bwd_re10k": rows, "bwd_kitti_raw": train[:3] + ["render_kernel_p"],
        "profile": ["query_kernel_p"]}[mode]
per = {p: collections.defaultdict(list) for p in pats}
for f in sorted(glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        for p in pats:
            if p in row.get("Kernel_Name", ""):
                per[p][row["Counter_Name"]].append(float(row["Counter_Value"]))
summary = {}
for p, agg in per.items():
    c = {k: sum(v) / len(v) for k, v in agg.items()}
    if not c:
        continue
    ns = next((d[0] for n, d in dur.items() if p in n), None)
    s = {"counters": c, "kernel_ms_rocprof": None if ns is None else ns / 1e6}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # rocprofv3 reports KB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read -> doubled (MI355X_MICROARCH.md, HBM)
        s.update(fetch_bytes=2 * c["FETCH_SIZE"] * 1024, write_bytes=c["WRITE_SIZE"] * 1024, fetch_size_kb_raw=c["FETCH_SIZE"], write_size_kb_raw=c["WRITE_SIZE"])
    if "SQ_WAVE_CYCLES" in c:
        s.update(valu_busy=c.get("SQ_ACTIVE_INST_VALU", 0) / c["SQ_WAVE_CYCLES"], wait_frac=c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"])
    if ns and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        s["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * ns * 2.4)      # 1024 SIMDs x kernel cycles at 2.4 GHz
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        s["l2_hit"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1)
    summary[p] = s
    print(p, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in s.items() if k != "counters"})
if mode == "fwd" and "render_kernel_p" in summary:
    s = summary["render_kernel_p"]
    rays = 245760
    s["valu_insts_per_ray"] = s["counters"].get("SQ_INSTS_VALU", 0) / rays
    s["mfma_insts_per_ray"] = s["counters"].get("SQ_INSTS_MFMA", 0) / rays
    json.dump(s, open("$TOP/traffic.json", "w"), indent=1)
else:
    k = "${3:-}"
    ksfx = "_k" + k if mode == "bwd_re10k" and k and k != "48" else ""      # a pass at another K than the yaml's names it (bench.py matches on that)
    json.dump(summary, open("$TOP/traffic_" + mode + ksfx + ".json", "w"), indent=1)
PY
