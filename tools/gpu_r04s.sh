cd $GRAFT_REPO_ROOT
O=gpurun_out/r04s
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_grad.py tests/test_gpu_scales.py tests/test_gpu_sparse_grad.py -q --timeout 600 --tb=short -rf 2>&1 | tail -8 > $O/pytest.txt; tail -8 $O/pytest.txt
for rep in 1 2; do for w in train re10k kitti_raw; do timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$w', round(j['ms_per_step'],3), round(j['roofline']['fwd_ms'],3), round(j['roofline']['bwd_ms'],3))" | tee -a $O/bench.txt; done; done
timeout 300 python bench.py --workload re10k --samples 128 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('re10k_k128', round(j['ms_per_step'],3), round(j['roofline']['fwd_ms'],3), round(j['roofline']['bwd_ms'],3))" | tee -a $O/bench.txt
export TMPDIR=/tmp
for K in 48 128; do
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_k$K -o trace -- python $GRAFT_REPO_ROOT/bench.py --workload re10k --samples $K --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/trace_k$K.log 2>&1)
python - <<PY
import csv,glob
for f in glob.glob("$O/trace_k$K/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for row in rows[:6]: print("k$K", f"{float(row['AverageNs'])/1e6:9.4f} ms x {row['Calls']:>4s} {row['Percentage']:>6s}%  {row['Name'][:90]}")
PY
done
find $O -type f ! -name "*stats.csv" ! -name "*.log" ! -name "*.txt" ! -name "*.json" ! -name "*.err" -delete
