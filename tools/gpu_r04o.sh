cd $GRAFT_REPO_ROOT
O=gpurun_out/r04o
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_loss.py tests/test_gpu_train_step.py tests/test_gpu_ddp.py -q --timeout 600 --tb=short -rf -x 2>&1 | tail -8 > $O/pytest.txt; tail -8 $O/pytest.txt
for w in train kitti_raw re10k; do timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$w', round(j['ms_per_step'],3), round(j['roofline']['fwd_ms'],3), round(j['roofline']['bwd_ms'],3))" | tee -a $O/bench.txt; done
for w in kitti_raw re10k; do python bench.py --workload $w --ops-profile --no-cpu-baseline 2> $O/ops_$w.txt >/dev/null; done
