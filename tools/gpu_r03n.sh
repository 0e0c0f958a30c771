cd $GRAFT_REPO_ROOT
O=gpurun_out/r03n
mkdir -p $O
for w in re10k train kitti_raw; do
  timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err
done
timeout 300 python bench.py --workload re10k --samples 128 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_re10k_k128.json 2> $O/bench_re10k_k128.err
python - <<PY
import json
for w in ("re10k","re10k_k128","train","kitti_raw"):
    try:
        j=json.loads([l for l in open("$O/bench_%s.json"%w) if l.startswith("{")][0]); r=j["roofline"]
        print("%-11s value %.4g ms/step %.3f kernel_ms %.3f frac %.3f"%(w,j["value"],j["ms_per_step"],r.get("kernel_ms") or 0,r["frac"]))
    except Exception as e:
        print(w,"ERR",e)
PY
timeout 300 python -m pytest tests/test_gpu_loss.py tests/test_gpu_train_step.py tests/test_gpu_protocol.py tests/test_gpu_ddp.py -m gpu -q --timeout 600 --tb=short -rf 2>&1 | tail -8
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_train -o train -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload train --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/$O/prof_train.log 2>&1
cd $GRAFT_REPO_ROOT; find $O/prof_train -type f ! -name "*stats.csv" -delete
