cd $GRAFT_REPO_ROOT
O=gpurun_out/r04k
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_grad.py tests/test_gpu_scales.py tests/test_gpu_train_step.py -q --timeout 600 --tb=short -rf -x 2>&1 | tail -6
export TMPDIR=/tmp
for shape in re10k kitti360; do
  K=""; if [ $shape = re10k ]; then K=48; fi
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_${shape} -o trace -- python $GRAFT_REPO_ROOT/tools/bwd_probe.py 3 $shape $K > $GRAFT_REPO_ROOT/$O/trace_${shape}.log 2>&1)
  python - <<PY
import csv,glob
for f in glob.glob("$O/trace_${shape}/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if any(k in row['Name'] for k in ('dwpe','rowsb','rows_kernel','scatter_kernel','project_bwd','Memset','fill')): print("$shape", f"{float(row['AverageNs'])/1e6:9.4f} ms x {row['Calls']:>4s}  {row['Name'][:70]}")
PY
done
find $O -type f ! -name "*stats.csv" ! -name "*.log" -delete
for w in train kitti_raw re10k; do timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$w.json 2>$O/bench_$w.err; python -c "
import json; j=json.loads([l for l in open('$O/bench_$w.json') if l.startswith('{')][0]); print('$w', round(j['ms_per_step'],3), round(j['roofline']['fwd_ms'],3), round(j['roofline']['bwd_ms'],3))"; done
