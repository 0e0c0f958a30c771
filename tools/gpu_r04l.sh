cd $GRAFT_REPO_ROOT
O=gpurun_out/r04l
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q --timeout 600 --tb=short -rf -k "48_lane or re10k or ragged" 2>&1 | tail -25 > $O/pytest.txt; tail -25 $O/pytest.txt
timeout 600 python -m pytest tests/test_gpu_grad.py tests/test_gpu_train_step.py tests/test_gpu_abi5.py -q --timeout 600 --tb=short -rf -x 2>&1 | tail -5
python tools/config_probe.py 2>&1 | tail -8
timeout 300 python bench.py --workload re10k --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('re10k', round(j['ms_per_step'],3), round(j['roofline']['fwd_ms'],3), round(j['roofline']['bwd_ms'],3))"
