cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_abi5.py -q --timeout 600 --tb=short -rf -x 2>&1 | tail -30 > $O/pytest_abi5.txt; tail -30 $O/pytest_abi5.txt
LIB_AB_PASSES=3 python tools/lib_ab.py --learn-empty r03base default:nohint default default:jitter > $O/lib_ab.txt 2>&1; cat $O/lib_ab.txt
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 --tb=short -rf --durations=8 2>&1 | tail -60 > $O/pytest_gpu.txt
tail -40 $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench_eval.json 2> $O/bench_eval.err; python - <<PY
import json
j=json.loads([l for l in open("$O/bench_eval.json") if l.startswith("{")][0]); r=j["roofline"]
print("eval value %.4g ms/step %.3f kernel_ms %.3f frac %.3f"%(j["value"],j["ms_per_step"],r["kernel_ms"],r["frac"]))
for k,v in j.get("others",{}).items():
    print(k, v.get("error") or ("%.4g %s ms/step %.3f fwd %.3f bwd %.3f"%(v["value"],v["unit"],v["ms_per_step"],v["roofline"].get("fwd_ms",0),v["roofline"].get("bwd_ms",0))))
PY
tail -3 $O/bench_eval.err
