cd $GRAFT_REPO_ROOT
O=gpurun_out/r04q
mkdir -p $O
timeout 600 python tools/lib_ab.py r04p default r04p:jitter default:jitter 2>&1 | tail -6 | tee $O/lib_ab.txt
timeout 900 python -m pytest tests/test_gpu_grad.py tests/test_gpu_train_step.py tests/test_gpu_determinism.py tests/test_gpu_abi5.py -q --timeout 600 --tb=short -rf 2>&1 | tail -8 > $O/pytest.txt; tail -8 $O/pytest.txt
for rep in 1 2; do for w in train re10k kitti_raw; do for lib in "" behindthescenes_amd/variants/libbts_r04p.so; do BTS_RENDER_LIB=$lib timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$w', '${lib:-default}', round(j['ms_per_step'],3), round(j['roofline']['fwd_ms'],3), round(j['roofline']['bwd_ms'],3))" | tee -a $O/ab.txt; done; done; done
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-others 2>/dev/null | python -c "
import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('eval', j['value'], round(j['ms_per_step'],4), j['roofline'].get('kernel_ms'))" | tee -a $O/ab.txt
