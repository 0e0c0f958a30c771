cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_grad.py tests/test_gpu_scales.py tests/test_gpu_train_step.py tests/test_gpu_handover.py -q --timeout 600 --tb=short -rf -x 2>&1 | tail -15 > $O/pytest_grad.txt; tail -15 $O/pytest_grad.txt
for w in train kitti_raw re10k; do timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$w.json 2>$O/bench_$w.err; python -c "
import json; j=json.loads([l for l in open('$O/bench_$w.json') if l.startswith('{')][0]); print('$w', round(j['ms_per_step'],3), round(j['roofline']['fwd_ms'],3), round(j['roofline']['bwd_ms'],3))"; done
bash tools/profile.sh r04e train > $O/profile_train.log 2>&1; tail -14 $O/profile_train.log
