#!/bin/bash
# rocprofv3 passes over the forward kernel on the BASELINE configs[1] workload.  Run on the GPU box via gpurun:
#   gpurun -- bash tools/profile.sh <tag>
# Writes raw output under gpurun_out/prof_<tag>/ ; copy the summaries you want judged into profiles/.
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $REPO/tools/kernel_probe.py 3 proj__all"
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
rocprofv3 -L > $OUT/counters.txt 2>&1
cd $REPO
find $OUT -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" -delete
du -sh $OUT; tail -3 $OUT/trace.log; find $OUT -name "*.csv" | head -30
python - <<PY
import csv, glob, collections
allc = {}
for f in sorted(glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)):
    print("==", f)
    for i, row in enumerate(csv.reader(open(f))):
        if i < 8: print(row)
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "render_kernel" in row.get("Kernel_Name", ""):
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("==", f.split("/")[-3:], {k: sum(v) / len(v) for k, v in agg.items()}, "n=", {k: len(v) for k, v in agg.items()})
    allc.update({k: sum(v) / len(v) for k, v in agg.items()})
import json
if "FETCH_SIZE" in allc and "WRITE_SIZE" in allc:
    # rocprofv3 reports KB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read -> doubled (MI355X_MICROARCH.md, HBM)
    json.dump({"fetch_bytes": 2 * allc["FETCH_SIZE"] * 1024, "write_bytes": allc["WRITE_SIZE"] * 1024, "fetch_size_kb_raw": allc["FETCH_SIZE"],
               "write_size_kb_raw": allc["WRITE_SIZE"], "counters": allc}, open("$OUT/traffic.json", "w"), indent=1)
PY
