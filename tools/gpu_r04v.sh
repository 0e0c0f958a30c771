cd $GRAFT_REPO_ROOT
O=gpurun_out/r04v
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sparse_grad.py tests/test_gpu_train_step.py tests/test_gpu_scales.py tests/test_gpu_abi5.py -q --timeout 600 --tb=short -rf 2>&1 | tail -25 > $O/pytest.txt; tail -25 $O/pytest.txt
for rep in 1 2; do for w in train re10k kitti_raw; do for mode in "" "--dense-projection"; do timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline $mode 2>/dev/null | python -c "
import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$w', '${mode:-sparse}', round(j['ms_per_step'],3), round(j['roofline']['fwd_ms'],3), round(j['roofline']['bwd_ms'],3))" | tee -a $O/ab.txt; done; done; done
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_train -o trace -- python $GRAFT_REPO_ROOT/bench.py --workload train --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/trace_train.log 2>&1)
python - <<PY
import csv,glob
for f in glob.glob("$O/trace_train/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for row in rows[:10]: print("train", f"{float(row['AverageNs'])/1e6:9.4f} ms x {row['Calls']:>4s} {row['Percentage']:>6s}%  {row['Name'][:90]}")
PY
find $O -type f ! -name "*stats.csv" ! -name "*.log" ! -name "*.txt" ! -name "*.json" ! -name "*.err" -delete
