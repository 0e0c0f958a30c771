cd $GRAFT_REPO_ROOT
O=gpurun_out/r03za
mkdir -p $O
python tools/lib_ab.py default presplit 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/lib_ab_split.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_grad.py tests/test_gpu_determinism.py tests/test_gpu_scales.py -m gpu -q --timeout 600 --tb=short -rf 2>&1 | tail -12 | tee $O/pytest_sel.txt
