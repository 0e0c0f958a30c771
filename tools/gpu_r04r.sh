cd $GRAFT_REPO_ROOT
O=gpurun_out/r04r
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_grad.py tests/test_gpu_train_step.py tests/test_gpu_scales.py tests/test_gpu_sparse_grad.py -q --timeout 600 --tb=short -rf 2>&1 | tail -8 > $O/pytest.txt; tail -8 $O/pytest.txt
for rep in 1 2; do for w in train re10k kitti_raw; do for lib in "" behindthescenes_amd/variants/libbts_r04q.so; do BTS_RENDER_LIB=$lib timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$w', '${lib:-default}', round(j['ms_per_step'],3), round(j['roofline']['fwd_ms'],3), round(j['roofline']['bwd_ms'],3))" | tee -a $O/ab.txt; done; done; done
BTS_RENDER_LIB= timeout 300 python bench.py --workload re10k --samples 128 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('re10k_k128', round(j['ms_per_step'],3), round(j['roofline']['fwd_ms'],3), round(j['roofline']['bwd_ms'],3))" | tee -a $O/ab.txt
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_k128 -o trace -- python $GRAFT_REPO_ROOT/bench.py --workload re10k --samples 128 --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/trace_k128.log 2>&1)
python - <<PY
import csv,glob
for f in glob.glob("$O/trace_k128/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for row in rows[:9]: print("k128", f"{float(row['AverageNs'])/1e6:9.4f} ms x {row['Calls']:>4s} {row['Percentage']:>6s}%  {row['Name'][:90]}")
PY
find $O -type f ! -name "*stats.csv" ! -name "*.log" ! -name "*.txt" ! -name "*.json" ! -name "*.err" -delete
