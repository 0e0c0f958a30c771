cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j
mkdir -p $O
export TMPDIR=/tmp
for shape in re10k kitti360; do
for v in default noflush; do
  if [ $v = default ]; then L=$PWD/behindthescenes_amd/libbts_render.so; else L=$PWD/behindthescenes_amd/variants/libbts_$v.so; fi
  K=""; if [ $shape = re10k ]; then K=48; fi
  (cd /tmp && BTS_RENDER_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_${shape}_$v -o trace -- python $GRAFT_REPO_ROOT/tools/bwd_probe.py 3 $shape $K > $GRAFT_REPO_ROOT/$O/trace_${shape}_$v.log 2>&1)
  python - <<PY
import csv,glob
for f in glob.glob("$O/trace_${shape}_$v/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if any(k in row['Name'] for k in ('dwpe','rowsb','rows_kernel','scatter_kernel','project_bwd')): print("$shape $v", f"{float(row['AverageNs'])/1e6:9.4f} ms x {row['Calls']:>4s}  {row['Name'][:70]}")
PY
done; done
BTS_RENDER_LIB=$PWD/behindthescenes_amd/variants/libbts_noflush.so timeout 300 python bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('train noflush', round(j['ms_per_step'],3), round(j['roofline']['fwd_ms'],3), round(j['roofline']['bwd_ms'],3))"
find $O -type f ! -name "*stats.csv" ! -name "*.log" -delete
