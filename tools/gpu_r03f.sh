set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --durations=8 -rf --tb=short 2>&1 | grep -v "^$" | tail -80 > $O/pytest_gpu.txt
grep -n "^FAILED\|passed\|failed" $O/pytest_gpu.txt | tail
timeout 400 python bench.py --steps 20 --warmup 3 > $O/bench_eval.json 2> $O/bench_eval.err
python - <<PY
import json
j=json.loads([l for l in open("$O/bench_eval.json") if l.startswith("{")][0]); r=j["roofline"]
print("EVAL value %.4g ms/step %.4f kernel_ms %.4f frac %.3f frac_exec %.3f"%(j["value"],j["ms_per_step"],r["kernel_ms"],r["frac"],r["frac_executed"]))
PY
python tools/bwd_probe.py 5 re10k 48 2>&1 | tail -2
python tools/bwd_probe.py 5 re10k 128 2>&1 | tail -1
python tools/bwd_probe.py 5 kitti360 2>&1 | tail -1
python tools/bwd_probe.py 5 kitti_raw 2>&1 | tail -1
bash tools/profile.sh r03f fwd > $O/profile_fwd.log 2>&1
tail -4 $O/profile_fwd.log
