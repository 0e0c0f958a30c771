cd $GRAFT_REPO_ROOT
O=gpurun_out/r04m
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sparse_grad.py tests/test_gpu_parity.py -q --timeout 600 --tb=short -rf -k "sparse or tile or kept or flags or 48_lane" 2>&1 | tail -25 > $O/pytest_new.txt; tail -25 $O/pytest_new.txt
timeout 900 python -m pytest tests/test_gpu_grad.py tests/test_gpu_train_step.py tests/test_gpu_scales.py tests/test_gpu_handover.py tests/test_gpu_abi5.py -q --timeout 600 --tb=short -rf 2>&1 | tail -8 > $O/pytest_grad.txt; tail -8 $O/pytest_grad.txt
for w in train kitti_raw re10k; do timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$w.json 2>$O/bench_$w.err; python -c "
import json; j=json.loads([l for l in open('$O/bench_$w.json') if l.startswith('{')][0]); print('$w', round(j['ms_per_step'],3), round(j['roofline']['fwd_ms'],3), round(j['roofline']['bwd_ms'],3))"; done
export TMPDIR=/tmp
for w in train re10k; do
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_$w -o trace -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/trace_$w.log 2>&1)
python - <<PY
import csv,glob
for f in glob.glob("$O/trace_$w/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for row in rows[:16]: print("$w", f"{float(row['AverageNs'])/1e6:9.4f} ms x {row['Calls']:>4s} {row['Percentage']:>6s}%  {row['Name'][:90]}")
PY
done
find $O -type f ! -name "*stats.csv" ! -name "*.log" ! -name "*.txt" ! -name "*.json" ! -name "*.err" -delete
