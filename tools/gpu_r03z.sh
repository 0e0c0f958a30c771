cd $GRAFT_REPO_ROOT
O=gpurun_out/r03z
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --timeout 600 --tb=short -rf --durations=4 2>&1 | tail -20 > $O/pytest_gpu.txt
tail -8 $O/pytest_gpu.txt
python bench.py > $O/bench_eval.json 2> $O/bench_eval.err
for w in train kitti_raw re10k profile; do
  timeout 400 python bench.py --workload $w --steps 10 --warmup 3 > $O/bench_$w.json 2> $O/bench_$w.err
done
timeout 400 python bench.py --workload re10k --samples 128 --steps 10 --warmup 3 > $O/bench_re10k_k128.json 2> $O/bench_re10k_k128.err
python - <<PY
import json
for w in ("eval","train","kitti_raw","re10k","re10k_k128","profile"):
    try:
        j=json.loads([l for l in open("$O/bench_%s.json"%w) if l.startswith("{")][0]); r=j["roofline"]
        print("%-11s value %.4g %s ms/step %.3f kernel_ms %.3f frac %.3f"%(w,j["value"],j["unit"],j["ms_per_step"],r.get("kernel_ms") or 0,r["frac"]))
    except Exception as e:
        print(w,"ERR",e)
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/profile.sh r03z fwd > $O/profile_fwd.log 2>&1; tail -2 $O/profile_fwd.log
