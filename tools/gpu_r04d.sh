cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d
mkdir -p $O
LIB_AB_PASSES=3 python tools/lib_ab.py --learn-empty r03base default default:jitter > $O/lib_ab.txt 2>&1; cat $O/lib_ab.txt
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 --tb=short -rf --durations=5 2>&1 | tail -60 > $O/pytest_gpu.txt
tail -30 $O/pytest_gpu.txt
bash tools/profile.sh r04d fwd > $O/profile_fwd.log 2>&1; tail -3 $O/profile_fwd.log
python tools/section_probe.py 2>&1 | head -3 > $O/section_probe.txt; cat $O/section_probe.txt
timeout 900 python bench.py > $O/bench_eval.json 2> $O/bench_eval.err; python - <<PY
import json
j=json.loads([l for l in open("$O/bench_eval.json") if l.startswith("{")][0]); r=j["roofline"]
print("eval value %.4g ms/step %.3f kernel_ms %.3f frac %.3f"%(j["value"],j["ms_per_step"],r["kernel_ms"],r["frac"]))
for k,v in j.get("others",{}).items():
    print(k, v.get("error") or ("%.4g %s ms/step %.3f fwd %.3f bwd %.3f"%(v["value"],v["unit"],v["ms_per_step"],v["roofline"].get("fwd_ms",0),v["roofline"].get("bwd_ms",0))))
PY
tail -3 $O/bench_eval.err
