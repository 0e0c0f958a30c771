cd $GRAFT_REPO_ROOT
O=gpurun_out/r03m
mkdir -p $O
for w in re10k train; do
  timeout 300 python bench.py --workload $w --warmup 3 --ops-profile 2> $O/ops_$w.txt
  timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err
done
python - <<PY
import json
for w in ("re10k","train"):
    j=json.loads([l for l in open("$O/bench_%s.json"%w) if l.startswith("{")][0]); r=j["roofline"]
    print("%-11s value %.4g ms/step %.3f kernel_ms %.3f"%(w,j["value"],j["ms_per_step"],r.get("kernel_ms") or 0))
PY
for i in 1 2 3; do timeout 200 python -m pytest tests/test_gpu_handover.py -m gpu -q -s --tb=line -k matches 2>&1 | grep -i "grad err\|passed\|failed\|Assertion" | tail -12; done > $O/handover_repeat.txt
cat $O/handover_repeat.txt
