cd $GRAFT_REPO_ROOT
O=gpurun_out/r04n
mkdir -p $O
for w in train re10k kitti_raw; do python bench.py --workload $w --tile-stats --no-cpu-baseline 2>&1 | grep "^map" | sed "s/^/$w /" | tee -a $O/tile_stats.txt; done
for rep in 1 2; do for w in train kitti_raw re10k; do for mode in "" "--dense-proj-grad"; do timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline $mode 2>/dev/null | python -c "
import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$w', '$mode', round(j['ms_per_step'],3), round(j['roofline']['fwd_ms'],3), round(j['roofline']['bwd_ms'],3))" | tee -a $O/ab.txt; done; done; done
export TMPDIR=/tmp
for mode in sparse dense; do
F=""; if [ $mode = dense ]; then F="--dense-proj-grad"; fi
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_re10k_$mode -o trace -- python $GRAFT_REPO_ROOT/bench.py --workload re10k --steps 5 --warmup 2 --no-cpu-baseline $F > $GRAFT_REPO_ROOT/$O/trace_re10k_$mode.log 2>&1)
python - <<PY
import csv,glob
for f in glob.glob("$O/trace_re10k_$mode/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for row in rows[:9]: print("re10k $mode", f"{float(row['AverageNs'])/1e6:9.4f} ms x {row['Calls']:>4s} {row['Percentage']:>6s}%  {row['Name'][:90]}")
PY
done
python bench.py --workload kitti_raw --ops-profile --no-cpu-baseline 2> $O/ops_kitti_raw.txt >/dev/null
find $O -type f ! -name "*stats.csv" ! -name "*.log" ! -name "*.txt" ! -name "*.json" ! -name "*.err" -delete
