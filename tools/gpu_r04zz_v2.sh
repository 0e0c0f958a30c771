cd $GRAFT_REPO_ROOT
O=gpurun_out/r04zz
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_grad.py tests/test_gpu_sparse_grad.py tests/test_gpu_train_step.py tests/test_gpu_scales.py -q --timeout 600 --tb=short -rf 2>&1 | tail -5 | tee $O/pytest_after_scatter_segments.txt
for w in train kitti_raw re10k; do timeout 200 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$w', round(j['ms_per_step'],3), round(j['roofline']['fwd_ms'],3), round(j['roofline']['bwd_ms'],3))" | tee -a $O/bench_after_scatter_segments.txt; done
