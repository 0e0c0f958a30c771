cd $GRAFT_REPO_ROOT
O=gpurun_out/r04y2
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_loss.py tests/test_gpu_scales.py tests/test_gpu_sparse_grad.py tests/test_gpu_parity.py -q --timeout 600 --tb=short -rf 2>&1 | tail -6 | tee $O/pytest.txt
for rep in 1 2; do for w in train re10k; do for lib in "" behindthescenes_amd/variants/libbts_r04w.so; do BTS_RENDER_LIB=$lib timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$w', '${lib:-default}', round(j['ms_per_step'],3), round(j['roofline']['fwd_ms'],3), round(j['roofline']['bwd_ms'],3))" | tee -a $O/ab.txt; done; done; done
for s in kitti360; do timeout 300 python tools/section_probe_train.py $s 2>&1 | grep -v amdgpu.ids | tee -a $O/section_probe.txt; done
