cd $GRAFT_REPO_ROOT
O=gpurun_out/r04zz
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --durations=5 -rf 2>&1 | tail -25 > $O/pytest_gpu.txt; tail -12 $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<PY
import json
j=json.loads([l for l in open("$O/bench_default.json") if l.startswith("{")][0])
print("eval", j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], j["roofline"]["frac"], j["roofline"].get("frac_executed"))
for k,v in (j.get("others") or {}).items():
    print(k, {a: (round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ("value","ms_per_step")}, {a: round(b,4) for a,b in (v.get("roofline") or {}).items() if isinstance(b,float)})
print("cpu_baseline", (j.get("cpu_baseline") or {}).get("value"))
PY
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_train -o trace -- python $GRAFT_REPO_ROOT/bench.py --workload train --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/trace_train.log 2>&1)
python - <<PY
import csv,glob
for f in glob.glob("$O/trace_train/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "bts::" in row["Name"]: print("train", f"{float(row['AverageNs'])/1e6:9.4f} ms x {row['Calls']:>4s}  {row['Name'][:80]}")
PY
find $O -type f ! -name "*stats.csv" ! -name "*.log" ! -name "*.txt" ! -name "*.json" ! -name "*.err" -delete
