cd $GRAFT_REPO_ROOT
O=gpurun_out/r04u
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train_step.py tests/test_gpu_determinism.py -q --timeout 600 --tb=short -rf 2>&1 | tail -8 > $O/pytest.txt; tail -8 $O/pytest.txt
for s in kitti360 re10k; do timeout 300 python tools/section_probe_train.py $s 2>&1 | grep -v amdgpu.ids | tee -a $O/section_probe_train.txt; done
for rep in 1 2; do for w in train re10k kitti_raw; do for lib in "" behindthescenes_amd/variants/libbts_r04z.so; do BTS_RENDER_LIB=$lib timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$w', '${lib:-default}', round(j['ms_per_step'],3), round(j['roofline']['fwd_ms'],3), round(j['roofline']['bwd_ms'],3))" | tee -a $O/ab.txt; done; done; done
timeout 300 python tools/lib_ab.py r04z default 2>&1 | tail -3 | tee $O/lib_ab.txt
