cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i
mkdir -p $O
export TMPDIR=/tmp
for v in default ablE1 ablE2 ablE3; do
  if [ $v = default ]; then L=$PWD/behindthescenes_amd/libbts_render.so; else L=$PWD/behindthescenes_amd/variants/libbts_$v.so; fi
  (cd /tmp && BTS_RENDER_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_$v -o trace -- python $GRAFT_REPO_ROOT/tools/bwd_probe.py 3 re10k 48 > $GRAFT_REPO_ROOT/$O/trace_$v.log 2>&1)
  python - <<PY
import csv,glob
for f in glob.glob("$O/trace_$v/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if any(k in row['Name'] for k in ('dwpe_rows','rowsb','scatter_kernel')): print("$v", f"{float(row['AverageNs'])/1e6:9.4f} ms x {row['Calls']:>4s}  {row['Name'][:70]}")
PY
done
find $O -type f ! -name "*stats.csv" ! -name "*.log" -delete
