cd $GRAFT_REPO_ROOT
O=gpurun_out/r03q
mkdir -p $O
python tools/lib_ab.py default noiterhead 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/lib_ab_iterhead.txt
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_determinism.py -m gpu -q --timeout 600 --tb=short -rf -x -k "golden or full_size_frame_vs or bit_deterministic" 2>&1 | tail -5 | tee $O/pytest_sel.txt
