cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_grad.py tests/test_gpu_scales.py -q --timeout 600 --tb=short -rf -x -k "re10k or multiscale or rows or block or k128 or golden" 2>&1 | tail -6
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --workload re10k --steps 6 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/trace.log 2>&1)
python - <<PY
import csv,glob
for f in glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True):
    for i,row in enumerate(csv.DictReader(open(f))):
        if i<12: print(f"{float(row['AverageNs'])/1e6:9.4f} ms x {row['Calls']:>4s}  {row['Name'][:100]}")
PY
find $O/trace -type f ! -name "*stats.csv" -delete
for w in re10k; do timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$w.json 2>$O/bench_$w.err; python -c "
import json; j=json.loads([l for l in open('$O/bench_$w.json') if l.startswith('{')][0]); print('$w', round(j['ms_per_step'],3), round(j['roofline']['fwd_ms'],3), round(j['roofline']['bwd_ms'],3))"; done
timeout 300 python bench.py --workload re10k --samples 128 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_re10k_k128.json 2>/dev/null; python -c "
import json; j=json.loads([l for l in open('$O/bench_re10k_k128.json') if l.startswith('{')][0]); print('re10k_k128', round(j['ms_per_step'],3), round(j['roofline']['fwd_ms'],3), round(j['roofline']['bwd_ms'],3))"
