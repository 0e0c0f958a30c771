cd $GRAFT_REPO_ROOT
O=gpurun_out/r03u
mkdir -p $O
python tools/bwd_probe.py 7 re10k 48 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/bwd_probe.txt
python tools/bwd_probe.py 7 re10k 128 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/bwd_probe.txt
python tools/bwd_probe.py 7 kitti360 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/bwd_probe.txt
python tools/bwd_probe.py 7 kitti_raw 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/bwd_probe.txt
BTS_RENDER_LIB=$PWD/behindthescenes_amd/variants/libbts_ticks.so python tools/rowsb_ticks.py 48 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/rowsb_ticks_k48.txt
timeout 600 python -m pytest tests/test_gpu_grad.py tests/test_gpu_scales.py tests/test_gpu_train_step.py tests/test_gpu_protocol.py -m gpu -q --timeout 600 --tb=short -rf 2>&1 | tail -8 | tee $O/pytest_sel.txt
